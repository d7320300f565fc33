#!/usr/bin/env python3
"""Converged-image goldens for 02/03/04 from the REFERENCE's own code (oracle/_ref/libref_e{2,3,4}.so, built by oracle/Makefile
from /root/reference): tests/golden/converged_0{2,3,4}.npz = per-pixel mean and sample variance of the linear (pre-gamma)
colour at 196x140, plus the sample count, the ray count and the measured single-core rate.

Run in the build container only (needs /root/reference to build oracle/_ref).  The reference's engines are function-static
and seeded from the wall clock in milliseconds, so the frame is cut into row bands rendered by separate PROCESSES (each with
its own engines), started a few milliseconds apart so that no two share a seed.

usage: gen_converged.py [02|03|03r|04 ...] [--spp N] [--procs P]"""
import argparse
import ctypes as C
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")
W, H = 196, 140
DP = C.POINTER(C.c_double)


def _band(args):
    which, spp, y0, y1, rank = args[:5]
    W, H = args[5:7] if len(args) > 5 else (globals()['W'], globals()['H'])      # spawned workers do not see a changed module global
    time.sleep(0.007 * rank)                       # distinct clock seeds (milliseconds) per process
    shadow = 16
    if which.startswith("02s"):                    # 02 at another shadow-sample count (BASELINE config 2's sweep): "02s64"
        shadow, which = int(which[3:]), "02"
    n = {"02": 2, "03": 3, "03r": 3, "04": 4}[which]
    L = C.CDLL(os.path.join(HERE, "_ref", f"libref_e{n}.so"))
    s = np.zeros((y1 - y0, W, 3), np.float64); q = np.zeros_like(s)
    rays = C.c_ulonglong(0)
    t0 = time.perf_counter()
    if which == "04":
        L.ref_e4_golden_rows(W, H, spp, y0, y1, s.ctypes.data_as(DP), q.ctypes.data_as(DP), C.byref(rays))
    elif which in ("03", "03r"):
        L.ref_e3_golden_rows(W, H, spp, y0, y1, 1 if which == "03r" else 0, s.ctypes.data_as(DP), q.ctypes.data_as(DP), C.byref(rays))
    else:
        seed = (int(time.time() * 1000) + 7919 * rank) & 0xFFFFFFFF      # the reference seeds its engine from the clock
        L.ref_e2_golden_rows(W, H, spp, shadow, C.c_uint(seed), y0, y1, s.ctypes.data_as(DP), q.ctypes.data_as(DP), C.byref(rays))
    return y0, y1, s, q, rays.value, time.perf_counter() - t0


def render(which, spp, procs):
    bands = np.linspace(0, H, procs * 4 + 1).astype(int)
    jobs = [(which, spp, int(bands[i]), int(bands[i + 1]), i, W, H) for i in range(len(bands) - 1)]
    s = np.zeros((H, W, 3), np.float64); q = np.zeros_like(s)
    rays, cpu = 0, 0.0
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(procs, maxtasksperchild=1) as pool:       # a fresh process (fresh engines) per band
        for y0, y1, bs, bq, r, dt in pool.imap_unordered(_band, jobs):
            s[y0:y1] = bs; q[y0:y1] = bq; rays += r; cpu += dt
    wall = time.perf_counter() - t0
    mean = s / spp
    var = np.maximum(q / spp - mean ** 2, 0.0) * spp / max(spp - 1, 1)          # unbiased per-pixel sample variance
    return mean, var, rays, wall, cpu


def sweep02(procs, evals=256, size=(98, 70)):
    """BASELINE config 2's shadow-sample sweep at a small frame: tests/golden/converged_02_sweep.npz holds, per sample count,
    the mean and the per-pixel variance of 'evals' evaluations of the reference's estimator."""
    global W, H
    W, H = size
    out = {"evals": np.int64(evals), "width": np.int64(W), "height": np.int64(H)}
    for shadow in (1, 2, 4, 8, 32, 64, 128, 256):
        mean, var, rays, wall, cpu = render(f"02s{shadow}", evals, procs)
        assert rays == W * H * evals * (1 + shadow), (rays, shadow)
        out[f"mean_s{shadow}"] = mean.astype(np.float32); out[f"var_s{shadow}"] = var.astype(np.float32)
        print(f"02 with {shadow} shadow samples: {rays} rays, wall {wall:.1f} s, mean {mean.mean(axis=(0, 1))}", flush=True)
    np.savez_compressed(os.path.join(OUT, "converged_02_sweep.npz"), **out)


def main():
    if "sweep02" in sys.argv:
        return sweep02(7)
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="*", default=["03", "03r", "04", "02"])
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--procs", type=int, default=7)
    a = ap.parse_args()
    default_spp = {"02": 512, "03": 16384, "03r": 8192, "04": 16384}
    for which in a.which:
        spp = a.spp or default_spp[which]
        mean, var, rays, wall, cpu = render(which, spp, a.procs)
        # single-core rate of the reference's code on this container: one process, a 12-row band in the middle of the frame
        y0, y1, _, _, r1, dt1 = _band((which, max(8, min(spp, 64)), H // 2 - 6, H // 2 + 6, 0))
        path = os.path.join(OUT, f"converged_{which}.npz")
        np.savez_compressed(path, mean=mean.astype(np.float32), var=var.astype(np.float32), spp=np.int64(spp), rays=np.int64(rays),
                            rays_per_sample=np.float64(rays / (W * H * spp)), single_core_rays_per_s=np.float64(r1 / dt1),
                            wall_s=np.float64(wall), cpu_s=np.float64(cpu), procs=np.int64(a.procs))
        print(f"{which}: {spp} spp, {rays} rays ({rays / (W * H * spp):.4f}/sample), wall {wall:.1f} s, cpu {cpu:.1f} s, "
              f"{rays / cpu / 1e6:.2f} Mrays/s/process under load, {r1 / dt1 / 1e6:.2f} Mrays/s one process alone; mean {mean.mean(axis=(0, 1))}; -> {path}", flush=True)


if __name__ == "__main__":
    main()
