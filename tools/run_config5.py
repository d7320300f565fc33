"""BASELINE.json config 5 at its stated size on one GPU, brute force (the reference's algorithm, the roofline contract):
the 18,194-primitive stress scene, 3840x2160, 4096 spp, depth 8 = 3.4e10 samples, ~1.6e11 rays, ~2.8e15 primitive tests
(about 16 minutes on one MI355X).  Rendered in sample ranges so that the run reports progress and can be split over
several invocations:  python3 tools/run_config5.py --begin 0 --end 2048 [--chunk 128] [--bvh]
Appends one JSON line per invocation to gpurun_out/config5.jsonl and stores the 8x8 block means of the accumulator."""
import argparse, json, os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
import numpy as np
import torch
from rtamd import abi, lib, scenes
ap = argparse.ArgumentParser()
ap.add_argument("--begin", type=int, default=0); ap.add_argument("--end", type=int, default=4096); ap.add_argument("--chunk", type=int, default=128)
ap.add_argument("--width", type=int, default=3840); ap.add_argument("--height", type=int, default=2160); ap.add_argument("--spp", type=int, default=4096)
ap.add_argument("--bvh", action="store_true")
a = ap.parse_args()
W, H, SPP = a.width, a.height, a.spp
ctx = lib.Context()
if a.bvh:
    ctx.set_option(abi.RT_OPT_ACCEL, abi.RT_ACCEL_BVH)
s, cam, _ = scenes.scene_stress(spp=SPP, seed=1)
acc = torch.zeros(H * W * 4, dtype=torch.float32, device="cuda"); part = torch.empty_like(acc)
rays, kms, t0 = 0, 0.0, time.perf_counter()
for b in range(a.begin, a.end, a.chunk):
    e = min(b + a.chunk, a.end)
    ctx.path_accumulate_device(s, cam, scenes.path_params(abi.RT_PATH_LIGHTS, SPP, max_depth=8, seed=1, sample_begin=b, sample_end=e), W, H, part.data_ptr())
    st = ctx.sync()
    acc += part; rays += st["rays"]; kms += st["kernel_ms"]
    print(f"samples [{b},{e}) of {SPP}: {st['kernel_ms'] / 1e3:.1f} s, {st['rays'] / st['kernel_ms'] / 1e3:.1f} Mrays/s; total {rays} rays in {kms / 1e3:.1f} s kernel time ({time.perf_counter() - t0:.0f} s wall)", flush=True)
fpr = scenes.flops_per_ray(s, 100)
img = acc.view(H, W, 4)
assert bool((img[:, :, 3] == a.end - a.begin).all()) and bool(torch.isfinite(img).all())
blocks = img[: H // 8 * 8, : W // 8 * 8, :3].reshape(H // 8, 8, W // 8, 8, 3).mean(dim=(1, 3)).cpu().numpy() / SPP
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
tag = f"{'bvh' if a.bvh else 'brute'}_{a.begin}_{a.end}"
np.save(os.path.join(ROOT, "gpurun_out", f"config5_{tag}_blockmeans.npy"), blocks.astype(np.float32))
line = {"config": 5, "workload": f"stress scene ({len(s.prims)} primitives, {'optional BVH mode' if a.bvh else 'brute force'}) {W}x{H}, samples [{a.begin},{a.end}) of {SPP} spp, depth 8",
        "kernel": ctx.last_path_kernel()[0], "kernel_s": kms / 1e3, "rays": rays, "samples": W * H * (a.end - a.begin), "rays_per_sample": rays / (W * H * (a.end - a.begin)),
        "rays_per_s": rays / (kms * 1e-3), "primitive_tests_per_s": rays * len(s.prims) / (kms * 1e-3), "algorithmic_flops_per_ray": fpr,
        "frac_fp32_peak": (None if a.bvh else rays * fpr / (kms * 1e-3) / 157.3e12), "image_mean_rgb": [float(x) for x in blocks.mean(axis=(0, 1))]}
with open(os.path.join(ROOT, "gpurun_out", "config5.jsonl"), "a") as f:
    f.write(json.dumps(line) + "\n")
print(json.dumps(line), flush=True)
