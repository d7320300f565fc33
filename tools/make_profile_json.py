"""gpurun_out/prof (tools/collect_profiles.sh) -> profiles/<tag>_*: kernel stats CSV, bench lines, PMC summary JSON."""
import collections, csv, glob, json, os, re, shutil, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "prof"); dst = os.path.join(ROOT, "profiles")

def line(path):
    return [l for l in open(path) if l.startswith('{"metric"')][-1]

open(os.path.join(dst, f"{tag}_bench.json"), "w").write(line(os.path.join(src, "bench.json")))
open(os.path.join(dst, f"{tag}_bench_under_kernel_trace.json"), "w").write(line(os.path.join(src, "bench_under_trace.json")))
shutil.copy(glob.glob(os.path.join(src, "trace", "*kernel_stats.csv"))[0], os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))

passes = {}
for name in ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_flops"):
    agg = collections.defaultdict(float); launches = set()
    for f in glob.glob(os.path.join(src, name, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "k_path" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 200000:   # the bench frame (item form: a persistent grid of 8 blocks per CU), not the 16x16 setup render
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); launches.add(r["Dispatch_Id"])
    passes[name] = dict(agg); passes[name + "_launches"] = len(launches)
b = json.loads(line(os.path.join(src, "pmc_sq.json")))
rays = b["roofline"]["rays_per_launch"]
sq, fe, wr = passes["pmc_sq"], passes["pmc_fetch"], passes["pmc_write"]
n = max(1, passes["pmc_sq_launches"])
valu = sq["SQ_INSTS_VALU"] / n
d = {
    "command": "rocprofv3 --pmc <counters> -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline (one run per counter group, tools/collect_profiles.sh)",
    "kernel": b["roofline"]["kernel"], "workload": "x".join(re.search(r"(\d+)x(\d+), (\d+) spp", b["config"]["workload"]).groups()), "launches_per_pass": n, "passes": {k: v for k, v in passes.items() if not k.endswith("_launches")},
    "derived": {
        "valu_wave_instructions": valu, "salu_wave_instructions": sq["SQ_INSTS_SALU"] / n,
        "valu_lane_utilisation": sq["SQ_THREAD_CYCLES_VALU"] / (sq["SQ_ACTIVE_INST_VALU"] * 64) if sq.get("SQ_ACTIVE_INST_VALU") else None,
        "valu_wave_instructions_per_64_rays": valu / (rays / 64.0),
        "waves": sq["SQ_WAVES"] / n,
        "hbm_read_bytes_raw": fe["FETCH_SIZE"] * 1024 / max(1, passes["pmc_fetch_launches"]),
        "hbm_write_bytes": wr["WRITE_SIZE"] * 1024 / max(1, passes["pmc_write_launches"]),
    },
}
fl = passes.get("pmc_flops") or {}
if fl:   # rocprof-counted FP32 work beside the algorithmic figure: wave-instructions x 64 lanes x lane utilisation
    nf = max(1, passes["pmc_flops_launches"]); util = d["derived"]["valu_lane_utilisation"] or 1.0
    add, mul, fma, trans = (fl.get("SQ_INSTS_VALU_" + k, 0.0) / nf for k in ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32"))
    d["derived"]["valu_f32_wave_instructions"] = {"add": add, "mul": mul, "fma": fma, "trans": trans, "int32": fl.get("SQ_INSTS_VALU_INT32", 0.0) / nf, "cvt": fl.get("SQ_INSTS_VALU_CVT", 0.0) / nf}
    d["derived"]["counted_fp32_flops_per_launch"] = (add + mul + 2 * fma + trans) * 64 * util
    d["derived"]["counted_fp32_flops_per_ray"] = d["derived"]["counted_fp32_flops_per_launch"] / rays
    d["derived"]["algorithmic_flops_per_ray"] = b["roofline"]["flops_per_ray"]
d["derived"]["hbm_read_bytes_x2_gfx950_correction"] = 2 * d["derived"]["hbm_read_bytes_raw"]
d["hbm_bytes_per_launch"] = d["derived"]["hbm_read_bytes_x2_gfx950_correction"] + d["derived"]["hbm_write_bytes"]
d["note"] = ("FETCH_SIZE/WRITE_SIZE are in KiB (rocprofv3). FETCH_SIZE doubled as MI355X_MICROARCH.md 'HBM' prescribes for gfx950. "
             "The kernel's algorithmic HBM bytes are its output, 16 B per pixel; the path is FP32-VALU-bound.")
json.dump(d, open(os.path.join(dst, f"{tag}_pmc_hbm_traffic.json"), "w"), indent=1)
print(json.dumps(d["derived"], indent=1)); print("hbm_bytes_per_launch", d["hbm_bytes_per_launch"])
