"""SURVEY.md 8(d) configs 1-5 on one GPU -> one JSON line each (profiles/rNN_configs.jsonl; "--no5" leaves config 5 out: its full-size run is tools/run_config5.py).  Config 4 is what bench.py
reports; it is repeated here at the same size for one table.  Config 5 runs at a reduced frame/spp (the full 4K/4096-spp
frame is ~1.6e15 primitive tests) with the full 18,194-primitive scene; its rate is per ray, so it carries over."""
import json, os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
from rtamd import abi, lib, scenes
ctx = lib.Context()
PEAK = 157.3e12
def best(fn, n=3): return min((fn() for _ in range(n)), key=lambda g: g["kernel_ms"])
def emit(cfg, name, g, flops, extra=None):
    rays = g["rays"] + g["shadow_rays"]
    d = {"config": cfg, "workload": name, "kernel_ms": round(g["kernel_ms"], 3), "call_ms": round(g["total_ms"], 3), "rays": g["rays"], "shadow_rays": g["shadow_rays"],
         "rays_per_s": rays / (g["kernel_ms"] * 1e-3), "algorithmic_flops": flops, "frac_fp32_peak": flops / (g["kernel_ms"] * 1e-3) / PEAK}
    d.update(extra or {})
    print(json.dumps(d), flush=True)
s, cam, prm = scenes.scene01()
g = best(lambda: ctx.render_whitted(s, cam, prm, 784, 561))
emit(1, "01_Basic 784x561 depth 4", g, g["rays"] * 223 + g["shadow_rays"] * 123)
for w, h in ((784, 561), (3840, 2160)):
    for ss in (1, 2, 4, 8, 16, 32, 64, 128, 256):
        s, cam, prm = scenes.scene02(ss, seed=1)
        g = best(lambda: ctx.render_whitted(s, cam, prm, w, h), 3)
        emit(2, f"02_SoftShadows {w}x{h} shadow_samples {ss}", g, g["rays"] * 189 + g["shadow_rays"] * 134, {"kernel": {2: "k_whitted_jit (scene-specialised)", 1: "k_whitted<true>"}[ctx.last_whitted_kernel()]})
s, cam, prm = scenes.scene03(256, seed=1)
g = best(lambda: ctx.render_path(s, cam, prm, 1920, 1080))
emit(3, "03_PathTracingBasics 1920x1080 256 spp depth 24", g, g["rays"] * scenes.flops_per_ray(s, 100), {"rays_per_sample": g["rays"] / g["samples"]})
s, cam, prm = scenes.scene04(1024, seed=1)
g = best(lambda: ctx.render_path(s, cam, prm, 3840, 2160), 3)
emit(4, "04_PathTracingLights 3840x2160 1024 spp depth 6", g, g["rays"] * scenes.flops_per_ray(s, 100), {"rays_per_sample": g["rays"] / g["samples"]})
if "--no5" in sys.argv: sys.exit(0)
s, cam, prm = scenes.scene_stress(spp=16, seed=1)
g = best(lambda: ctx.render_path(s, cam, prm, 1920, 1080), 1)
emit(5, f"stress scene ({len(s.prims)} primitives, brute force) 1920x1080 16 spp depth 8 (reduced from 3840x2160, 4096 spp)", g, g["rays"] * scenes.flops_per_ray(s, 100),
     {"rays_per_sample": g["rays"] / g["samples"], "primitive_tests_per_s": g["rays"] * len(s.prims) / (g["kernel_ms"] * 1e-3)})
ctx.set_option(abi.RT_OPT_ACCEL, abi.RT_ACCEL_BVH)      # for information: the optional BVH walk on the same frame (same image; outside the roofline contract)
b = best(lambda: ctx.render_path(s, cam, prm, 1920, 1080), 2)
assert b["rays"] == g["rays"]
print(json.dumps({"config": 5, "workload": "same frame through the optional BVH (RT_OPT_ACCEL)", "kernel_ms": round(b["kernel_ms"], 3), "call_ms": round(b["total_ms"], 3), "rays": b["rays"],
                  "rays_per_s": b["rays"] / (b["kernel_ms"] * 1e-3), "speedup_vs_brute_force": g["kernel_ms"] / b["kernel_ms"]}), flush=True)
