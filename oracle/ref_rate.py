#!/usr/bin/env python3
"""Times the REFERENCE's own 04 code (oracle/_ref/libref_e4.so: scene::List::intersectsRay, materials, samplers, <random>
engines, driven by the TraceTay-following loop of ref_harness_e4.cpp) on host cores: `procs` processes (the reference's
engines are function-static and not thread-safe, so processes, not threads), each rendering `rows_per_proc` rows from the
middle of a width x height frame at `spp` samples per pixel.  Prints one JSON object.  TEST / BENCH INFRASTRUCTURE ONLY
(bench.py's cpu_baseline leg runs it as a child process; it never touches the GPU)."""
import argparse
import ctypes as C
import json
import multiprocessing as mp
import os
import time

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_ref", "libref_e4.so")


def _rows(args):
    w, h, spp, y0, y1, rank = args
    time.sleep(0.005 * rank)      # distinct clock seeds
    L = C.CDLL(LIB)
    n = (y1 - y0) * w * 3
    s = (C.c_double * n)(); q = (C.c_double * n)()
    rays = C.c_ulonglong(0)
    t0 = time.perf_counter()
    L.ref_e4_golden_rows(w, h, spp, y0, y1, s, q, C.byref(rays))
    return rays.value, time.perf_counter() - t0


def measure(width, height, spp, procs, rows_per_proc):
    y0 = height // 2 - (procs * rows_per_proc) // 2
    jobs = [(width, height, spp, y0 + k * rows_per_proc, y0 + (k + 1) * rows_per_proc, k) for k in range(procs)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(procs) as pool:
        res = pool.map(_rows, jobs)
    wall = time.perf_counter() - t0
    rays = sum(r for r, _ in res)
    busy = max(dt for _, dt in res)            # the slowest process: what a caller would wait for
    return {"rays": rays, "seconds": busy, "wall_incl_process_start": wall, "rays_per_s": rays / busy, "procs": procs,
            "rows": [y0, y0 + procs * rows_per_proc - 1], "per_process_rays_per_s": [r / dt for r, dt in res]}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=3840); ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--spp", type=int, default=1024); ap.add_argument("--procs", type=int, default=16)
    ap.add_argument("--rows-per-proc", type=int, default=1)
    a = ap.parse_args()
    print(json.dumps(measure(a.width, a.height, a.spp, a.procs, a.rows_per_proc)))
