"""What one rank of an N-GPU bench step does, measured on ONE GPU: rank r's sample share [r*spp/N, (r+1)*spp/N) of the 4K / 1024-spp frame of the 04 scene
through the same calls as bench.py's step (path_accumulate_device -> pack -> unpack -> resolve_device on a side stream; no collective: there is one GPU).
Gives the per-rank time that bounds the N-GPU step from below, i.e. the scaling the tracing itself allows: 242 ms / (N x share time).  Not a scaling measurement.
python3 tools/shares_probe.py [spp]"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
os.environ.setdefault("RT_MI355_JIT", "2")
from rtamd import lib as _early
_early.prefer_system_compiler()
import numpy as np
import torch
from rtamd import abi, lib, scenes, shard

W, H = 3840, 2160
SPP = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = lib.Context(0)
s, cam, _ = scenes.scene04(SPP, seed=1)
stream = torch.cuda.Stream(); sp = stream.cuda_stream
accum = torch.zeros(H * W * 4, dtype=torch.float32, device="cuda")
bgra = torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda")
rgb = torch.zeros(H * W, 3, dtype=torch.float32, device="cuda")


def step(prm):
    with torch.cuda.stream(stream):
        ctx.path_accumulate_device(s, cam, prm, W, H, accum.data_ptr(), sp)
        a4 = accum.view(H * W, 4)
        rgb.copy_(a4[:, :3]); a4[:, :3].copy_(rgb)
        ctx.resolve_device(accum.data_ptr(), W, H, float(SPP), 1, bgra.data_ptr(), 0, None, sp)


full = None
for n in (1, 2, 4, 8):
    for r in sorted({0, n - 1}):
        b, e = shard.sample_range(r, n, SPP)
        prm = scenes.path_params(abi.RT_PATH_LIGHTS, SPP, seed=1, sample_begin=b, sample_end=e)
        step(prm); ctx.sync(sp)                      # upload, specialised build, warm clocks
        kms, rays = [], 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            step(prm); st = ctx.sync(sp); kms.append(st["kernel_ms"]); rays = st["rays"]
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 5 * 1e3
        k = float(np.mean(kms))
        if full is None: full = wall
        print(f"N={n} rank {r}: samples [{b}, {e})  rays {rays}  kernel {k:8.3f} ms  step wall {wall:8.3f} ms  ->  tracing allows {full / wall:5.2f}x at N={n} "
              f"(ideal {n}); ns per 1000 rays {k * 1e6 / rays * 1e3:.3f}", flush=True)
