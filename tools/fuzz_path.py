"""Random path-traced scenes (03 sky estimator / 04 lights estimator) on the GPU against the oracle at equal seeds.
python3 tools/fuzz_path.py [n_scenes] [first_seed]
Per scene: scene queries per sample (HIP, oracle, relative difference), share of pixels with the oracle's own query count, relative
difference of the channel means.  Exit status 1 when queries/sample differ by more than 2e-3 or a channel mean by more than 1e-2."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
from rtamd import abi, lib, scenes
import orc as oracle

sys.path.insert(0, os.path.join(ROOT, "tools"))
from fuzz_scenes import FAMILIES, random_scene, random_scene_b, random_scene_c, random_scene_d      # noqa: E402

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    ctx = lib.Context()
    if os.environ.get("FUZZ_ARITH") == "ref": ctx.set_option(abi.RT_OPT_ARITHMETIC, abi.RT_ARITH_REFERENCE)      # the reference's arithmetic for rays and spheres
    W, H, SPP = 96, 64, (16 if os.environ.get("FUZZ_FAMILY") == "d" else 64)
    bad = 0; worst_q = 0.0; worst_m = 0.0
    for seed in range(first, first + n):
        s, cam, variant = FAMILIES[os.environ.get("FUZZ_FAMILY", "")](seed)
        prm = scenes.path_params(variant, SPP, seed=seed, **({"t_max": float(os.environ["FUZZ_TMAX"])} if os.environ.get("FUZZ_TMAX") else {}), **({"t_min": float(os.environ["FUZZ_TMIN"])} if os.environ.get("FUZZ_TMIN") else {}), **({"max_depth": int(os.environ["FUZZ_DEPTH"])} if os.environ.get("FUZZ_DEPTH") else {}))      # FUZZ_TMAX: a short tMax brings the sphere rule 'a hit AT tMax' into play
        o = oracle.render_path_px(s, cam, prm, W, H, threads=16)
        line = f"seed {seed:3d} {'03' if variant == abi.RT_PATH_SKY else '04'} prims {len(s.prims):2d} oracle q/s {o['rays'] / (W * H * SPP):8.4f}:"
        for name, flags in (("generic", 8), ("generated", 16)):
            acc = torch.zeros(H, W, 4, device="cuda")
            ctx.set_debug_flags(4 | flags)
            try:
                ctx.path_accumulate_device(s, cam, prm, W, H, acc.data_ptr()); st = ctx.sync()
            finally: ctx.set_debug_flags(0)
            a = acc.cpu().numpy()
            relq = (st["rays"] - o["rays"]) / o["rays"]
            same = float((a[:, :, 3].astype(np.int64) == o["px_rays"]).mean())
            mo = o["sum"].reshape(-1, 3).mean(axis=0).astype(np.float64); mg = a[:, :, :3].reshape(-1, 3).mean(axis=0).astype(np.float64)
            lit = mo >= 0.01                       # darker channels (a camera shut into a box: means of 1e-9 ... 1e-3) are a few rare paths: noise, not arithmetic
            relm = float(np.abs((mg - mo)[lit] / mo[lit]).max()) if lit.any() else 0.0
            line += f"  {name} k{ctx.last_path_kernel()[0]} rel q {relq:+.1e} px= {same:.4f} mean {relm:.1e}"
            worst_q = max(worst_q, abs(relq)); worst_m = max(worst_m, relm)
            if abs(relq) > 2e-3 or relm > 1e-2: bad += 1; line += " <<<"
        # the item form (the default on real frames; forced here by debug flag 1024), generic and specialised: the same paths, so the same count
        for name, flags in (("items", 8 | 1024), ("items-gen", 16 | 1024)):
            ctx.set_debug_flags(flags)
            try: it = ctx.render_path(s, cam, prm, W, H)
            finally: ctx.set_debug_flags(0)
            same = it["rays"] == st["rays"]
            line += f"  {name} form {ctx.last_path_form()[0]} {'rays=' if same else 'RAYS DIFFER ' + str(it['rays'] - st['rays'])}"
            if not same: bad += 1; line += " <<<"
        # the BVH walk against the per-lane brute-force kernel: the same image bit for bit (ties, the tMax rule, strict slots included)
        if os.environ.get("FUZZ_ARITH") != "ref":
            ctx.set_debug_flags(8 | 512)
            try: bf = ctx.render_path(s, cam, prm, W, H)
            finally: ctx.set_debug_flags(0)
            ctx.set_option(abi.RT_OPT_ACCEL, abi.RT_ACCEL_BVH)
            try: bv = ctx.render_path(s, cam, prm, W, H)
            finally: ctx.set_option(abi.RT_OPT_ACCEL, abi.RT_ACCEL_NONE)
            same = bv["rays"] == bf["rays"] and np.array_equal(bv["rgb"], bf["rgb"])
            line += f"  bvh {'bits=' if same else 'DIFFERS rays ' + str(bv['rays'] - bf['rays'])}"
            if not same: bad += 1; line += " <<<"
        print(line, flush=True)
    print(f"largest |rel queries/sample| {worst_q:.2e}, largest rel channel-mean difference {worst_m:.2e}; FAILURES {bad}")
    return 1 if bad else 0
sys.exit(main())
