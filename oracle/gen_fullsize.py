#!/usr/bin/env python3
"""BASELINE.json's configurations at their FULL size, rendered by the REFERENCE's own code (oracle/_ref/libref_e{2,3,4}.so, the
harness loops of oracle/ref_harness_e*.cpp that also made tests/golden/converged_*.npz), reduced to block means so that the result
is a small fixture: tests/golden/fullsize_0{2,3,4}.npz = per block of BxB pixels the mean linear colour, the mean per-pixel
sample variance and the pixel count, plus sample and ray counts and the measured rate.

  04: 3840x2160, 1024 spp (config 4, the headline frame)      ~5.8e10 rays: ~9 minutes on 16 cores of the GPU box's host
  03: 1920x1080,  256 spp (config 3)                          ~1.2e9 rays
  02: 3840x2160, 16 shadow samples, 64 evaluations (config 2) ~9e9 rays

CPU only; it runs on the GPU box's host cores (the prebuilt oracle/_ref/*.so travels there, /root/reference does not and is not
needed) because the build container has 8 slower cores.  Progress goes to stdout and gpurun_out/ once per band.

usage: gen_fullsize.py 04|03|02 [--procs 16] [--out gpurun_out]"""
import argparse
import ctypes as C
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DP = C.POINTER(C.c_double)
CONFIG = {"04": dict(w=3840, h=2160, spp=1024, block=32, rows=8), "03": dict(w=1920, h=1080, spp=256, block=16, rows=8),
          "02": dict(w=3840, h=2160, spp=64, block=32, rows=32)}        # 02: spp = evaluations of the 16-sample estimator


def _worker(which, jobs, results, rank):
    time.sleep(0.011 * rank)                       # the reference seeds its engines from the clock in milliseconds: distinct seeds
    n = {"02": 2, "03": 3, "04": 4}[which]
    L = C.CDLL(os.path.join(HERE, "_ref", f"libref_e{n}.so"))
    cfg = CONFIG[which]
    W, H, spp = cfg["w"], cfg["h"], cfg["spp"]
    while True:
        job = jobs.get()
        if job is None:
            return
        y0, y1 = job
        s = np.zeros((y1 - y0, W, 3), np.float64); q = np.zeros_like(s)
        rays = C.c_ulonglong(0)
        t0 = time.perf_counter()
        if which == "04":
            L.ref_e4_golden_rows(W, H, spp, y0, y1, s.ctypes.data_as(DP), q.ctypes.data_as(DP), C.byref(rays))
        elif which == "03":
            L.ref_e3_golden_rows(W, H, spp, y0, y1, 0, s.ctypes.data_as(DP), q.ctypes.data_as(DP), C.byref(rays))
        else:
            seed = (int(time.time() * 1000) + 7919 * rank + 104729 * y0) & 0xFFFFFFFF
            L.ref_e2_golden_rows(W, H, spp, 16, C.c_uint(seed), y0, y1, s.ctypes.data_as(DP), q.ctypes.data_as(DP), C.byref(rays))
        mean = s / spp
        var = np.maximum(q / spp - mean ** 2, 0.0) * spp / max(spp - 1, 1)       # unbiased per-pixel sample variance
        B = cfg["block"]
        nbx = (W + B - 1) // B
        pad = nbx * B - W
        if pad:
            mean = np.pad(mean, ((0, 0), (0, pad), (0, 0))); var = np.pad(var, ((0, 0), (0, pad), (0, 0)))
        # the band lies inside one row of blocks (rows-per-band divides the block size): column-block sums of its pixels
        bm = mean.reshape(y1 - y0, nbx, B, 3).sum(axis=(0, 2)); bv = var.reshape(y1 - y0, nbx, B, 3).sum(axis=(0, 2))
        results.put((y0, y1, bm, bv, rays.value, time.perf_counter() - t0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", choices=sorted(CONFIG))
    ap.add_argument("--procs", type=int, default=16)
    ap.add_argument("--out", default=os.path.join(HERE, "..", "gpurun_out"))
    a = ap.parse_args()
    cfg = CONFIG[a.which]
    W, H, spp, B, R = cfg["w"], cfg["h"], cfg["spp"], cfg["block"], cfg["rows"]
    assert B % R == 0
    os.makedirs(a.out, exist_ok=True)
    ctx = mp.get_context("spawn")
    jobs, results = ctx.Queue(), ctx.Queue()
    bands = [(y, min(y + R, H)) for y in range(0, H, R)]
    for b in bands:
        jobs.put(b)
    for _ in range(a.procs):
        jobs.put(None)
    procs = [ctx.Process(target=_worker, args=(a.which, jobs, results, r)) for r in range(a.procs)]
    t0 = time.perf_counter()
    for p in procs:
        p.start()
    nby, nbx = (H + B - 1) // B, (W + B - 1) // B
    sm = np.zeros((nby, nbx, 3)); sv = np.zeros((nby, nbx, 3)); cnt = np.zeros((nby, nbx))
    rays, cpu, done = 0, 0.0, 0
    log = open(os.path.join(a.out, f"fullsize_{a.which}.log"), "w")
    for _ in bands:
        y0, y1, bm, bv, r, dt = results.get()
        by = y0 // B
        sm[by] += bm; sv[by] += bv
        cnt[by] += (y1 - y0) * np.minimum(B, W - B * np.arange(nbx))
        rays += r; cpu += dt; done += y1 - y0
        msg = f"{a.which}: rows {done}/{H}, {rays} rays, wall {time.perf_counter() - t0:.0f} s, {rays / max(cpu, 1e-9) / 1e6:.2f} Mrays/s per process"
        print(msg, flush=True); log.write(msg + "\n"); log.flush()
    for p in procs:
        p.join()
    wall = time.perf_counter() - t0
    path = os.path.join(a.out, f"fullsize_{a.which}.npz")
    np.savez_compressed(path, mean=(sm / cnt[..., None]).astype(np.float32), var=(sv / cnt[..., None]).astype(np.float32), count=cnt.astype(np.int32),
                        width=np.int64(W), height=np.int64(H), spp=np.int64(spp), block=np.int64(B), rays=np.int64(rays),
                        rays_per_sample=np.float64(rays / (W * H * spp)), wall_s=np.float64(wall), cpu_s=np.float64(cpu), procs=np.int64(a.procs),
                        rays_per_s=np.float64(rays / wall))
    msg = (f"{a.which}: {W}x{H} x {spp}: {rays} rays ({rays / (W * H * spp):.4f}/sample) in {wall:.1f} s on {a.procs} processes = {rays / wall / 1e6:.1f} Mrays/s "
           f"({rays / cpu / 1e6:.2f} per process); frame mean {(sm.sum(axis=(0, 1)) / cnt.sum())}; -> {path}")
    print(msg, flush=True); log.write(msg + "\n"); log.close()


if __name__ == "__main__":
    sys.exit(main())
