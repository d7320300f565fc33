#!/usr/bin/env python3
"""bench.py - rays/s of the 04_PathTracingLights tracing path on MI355X.

A step = one full render of BASELINE.json config 4: the 04 scene at 3840x2160, 1024 spp, depth 6
(8.49e9 pixel-samples, ~5.8e10 rays), inputs (the flattened scene) resident in HBM, output = the
8-bit BGRA frame in HBM.  With N ranks (one process per GPU, torch.distributed/RCCL) the frame's
samples are split into N contiguous ranges - rank r renders samples [r*spp/N, (r+1)*spp/N) of every
pixel - and the per-GPU float accumulators (3840*2160*4 floats = 132.7 MB) are combined with ONE
all-reduce over xGMI before the resolve (the three data floats of every pixel, packed: 99.5 MB); total work is fixed, so scaling is "strong".
N > 1 runs add per-rank ray counts and the stream-side phases of a step (render / pack / all-reduce / unpack + resolve, HIP events) to `config`.

Prints ONE JSON line (rank 0).  roofline: FP32 VALU (no MFMA on this path), algorithmic FLOPs per ray
from SURVEY.md 8(d) (04 scene: 249) x rays the kernel counted / the kernel's HIP-event duration.
cpu_baseline: the reference's own code (oracle/_ref/libref_e4.so, kind "reference") on 16 host processes, one row of the same
frame each; if that library is absent, the CPU oracle (a restatement, kind "port") on a band of rows."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))     # orc.py: the checker's binding

PEAK_FP32_VALU_TFLOPS = 157.3      # MI355X_MICROARCH.md "Peak FP32 (vector)"


def cpu_baseline_reference(width, height, spp, procs):
    """The REFERENCE's own 04 code (oracle/_ref/libref_e4.so, built from /root/reference by oracle/Makefile and shipped as a
    built library) on `procs` host processes, one row of the same frame each; run as a child process that never touches
    the GPU.  None when the library is not there (then the port is timed instead)."""
    import subprocess
    lib_path = os.path.join(ROOT, "oracle", "_ref", "libref_e4.so")
    if not os.path.exists(lib_path):
        return None
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_rate.py"), "--width", str(width), "--height", str(height),
                              "--spp", str(spp), "--procs", str(procs), "--rows-per-proc", "1"], check=True, capture_output=True, text=True, timeout=600)
        m = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:      # the baseline is a reported figure, never a reason to lose the bench line
        print(f"cpu_baseline (reference) failed: {e}", file=sys.stderr)
        return None
    return {"value": m["rays_per_s"], "unit": "rays/s", "cores": procs, "kind": "reference",
            "sample": f"the reference's own 04 code (scene::List::intersectsRay, Materials/*.hpp, RndHemisphereVec3, its <random> engines; g++ -O2, "
                      f"oracle/_ref/libref_e4.so) driven by the TraceTay-following loop of oracle/ref_harness_e4.cpp: rows {m['rows'][0]}..{m['rows'][1]} of the same "
                      f"{width}x{height} frame at {spp} spp, one row per process: {m['rays']} rays in {m['seconds']:.1f} s on {procs} processes "
                      f"({m['rays_per_s'] / procs / 1e6:.2f} Mrays/s per core)"}


def cpu_baseline(scenes, width, height, spp, rows, threads):
    """Times the oracle on `rows` rows in the middle of the same frame (same scene, camera, spp)."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    import orc as oracle
    s, cam, prm = scenes.scene04(spp, seed=1)
    L = oracle.lib()
    sc = s.c()
    y0 = height // 2 - rows // 2
    counts = []

    def band(y):
        cnt = (C.c_uint64 * 2)(0, 0)
        rc = L.orc_render_path(C.byref(sc), C.byref(cam), C.byref(prm), width, height, y, y + 1, None, None, None, cnt)
        assert rc == 0
        counts.append(cnt[0])

    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(band, range(y0, y0 + rows)))
    dt = time.perf_counter() - t0
    rays = int(sum(counts))
    return {"value": rays / dt, "unit": "rays/s", "cores": threads, "kind": "port",
            "sample": f"oracle/rt_oracle.c (CPU restatement, gcc -O2) on rows {y0}..{y0 + rows - 1} of the same {width}x{height} frame at {spp} spp: "
                      f"{rays} rays in {dt:.1f} s on {threads} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=16)
    ap.add_argument("--watchdog", type=int, default=900, help="seconds after which a run that is still going prints every thread's stack and exits (0 = off): a rank stuck in a collective must not hang the job")
    args = ap.parse_args()

    # this pool's host driver supports only dmabuf IPC: without it RCCL's cross-process handles fail (hipIpcGetMemHandle: invalid
    # argument).  The launcher exports it already; kept here so that a rank started from a bare environment behaves the same
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # the image's own compiler library in place of the older one PyTorch bundles (rtamd/lib.py; +3 % on the generated kernel)
    from rtamd import lib as _lib_early
    jit_compiler = _lib_early.prefer_system_compiler()
    import torch
    import torch.distributed as dist
    from rtamd import abi, lib, scenes, shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.watchdog > 0:
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog, exit=True)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path to time")
    # BENCH_REHEARSAL=1: rehearse the N > 1 code path on a box with ONE GPU - every rank on device 0, gloo instead of RCCL
    # (RCCL refuses two ranks on one device).  The numbers of such a run mean nothing; the driver never sets it.
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    # BENCH_ONE_RANK_COLLECTIVE=1 (with --gpus 1): run the N > 1 step - pack, RCCL all-reduce, unpack - as written, over a ONE-rank
    # RCCL communicator, so that the `nccl` branch executes on a one-GPU box too.  Its line is marked; the driver never sets it.
    one_rank_collective = world == 1 and os.environ.get("BENCH_ONE_RANK_COLLECTIVE") == "1"
    if one_rank_collective:
        os.environ.setdefault("MASTER_PORT", "29531")
    collective = world > 1 or one_rank_collective
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not rehearsal and torch.cuda.device_count() < world:      # fail loudly, never fold ranks onto fewer devices
            raise SystemExit(f"--gpus {world} but this node shows {torch.cuda.device_count()} GPU(s) to rank {rank}")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus asked for {args.gpus}")

    W, H, SPP = args.width, args.height, args.spp
    ctx = lib.Context(local)
    s, cam, _ = scenes.scene04(SPP, seed=1)
    s_begin, s_end = shard.sample_range(rank, world, SPP)
    prm = scenes.path_params(abi.RT_PATH_LIGHTS, SPP, seed=1, sample_begin=s_begin, sample_end=s_end)
    # A real side stream, made current for every step: the library's kernels are enqueued on ITS handle and torch orders the
    # all-reduce (and RCCL's internal stream) against the same stream.  Torch's default stream has handle 0, which the
    # library reads as "use my own non-blocking stream" - that stream is ordered against nothing torch does, so passing 0
    # here would let RCCL reduce a half-written accumulator.
    stream = torch.cuda.Stream()
    sp = stream.cuda_stream
    assert sp != 0
    accum = torch.zeros(H * W * 4, dtype=torch.float32, device="cuda")
    bgra = torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda")
    # the collective carries the three data floats of a pixel, packed (99.5 MB at 4K), not the float4 accumulator (132.7 MB)
    rgb = torch.zeros(H * W, 3, dtype=torch.float32, device="cuda") if collective else None
    torch.cuda.synchronize()
    empty_share = shard.is_empty((s_begin, s_end))
    # Host wait between render and collective: only in the one-GPU rehearsal (BENCH_HOST_SYNC=1 forces it, =0 removes it).
    # A real run orders render -> all-reduce -> resolve on ONE stream and never blocks the host inside a step.
    host_sync = os.environ.get("BENCH_HOST_SYNC", "1" if rehearsal else "0") == "1"
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]      # render begin / end, all-reduce begin / end, resolve end - all on `stream`

    def step():
        with torch.cuda.stream(stream):
            ev[0].record(stream)
            if empty_share:
                accum.zero_()                                        # spp < world: this rank adds nothing
            else:
                ctx.path_accumulate_device(s, cam, prm, W, H, accum.data_ptr(), sp)
            ev[1].record(stream)
            if collective:
                if host_sync:
                    ctx.sync(sp)
                a4 = accum.view(H * W, 4)
                rgb.copy_(a4[:, :3])
                ev[2].record(stream)
                dist.all_reduce(rgb, op=dist.ReduceOp.SUM)            # the one collective: per-GPU sums of the sample shares
                ev[3].record(stream)
                a4[:, :3].copy_(rgb)
            else:
                ev[2].record(stream); ev[3].record(stream)
            ctx.resolve_device(accum.data_ptr(), W, H, float(SPP), 1, bgra.data_ptr(), 0, None, sp)
            ev[4].record(stream)

    def fence():
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    # one-time setup, like loading the library: the library compiles a kernel specialised for this scene (hiprtc, ~0.7 s).  Its
    # default is to do that when it sees a scene for the second time; here it is asked to do it at the first render, which
    # is the first warmup step (with --warmup 0: a tiny 16x16 render, so that the compile is never inside the timed region)
    os.environ["RT_MI355_JIT"] = "2"          # the documented switch: specialise at the first render of a scene
    if args.warmup == 0:
        ctx.render_path(s, cam, scenes.path_params(abi.RT_PATH_LIGHTS, 1, seed=1), 16, 16)
    if collective:                            # the communicator's lazy set-up (RCCL: rings, IPC handles) is one-time too
        with torch.cuda.stream(stream):
            dist.all_reduce(torch.zeros(1024, dtype=torch.float32, device="cuda"), op=dist.ReduceOp.SUM)
    for _ in range(args.warmup):
        step()
    fence()
    kernel_ms, rays, phases = [], 0, []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        st = ctx.sync(sp)          # waits for this step (also collects the kernel's ray counter and HIP-event time)
        kernel_ms.append(st["kernel_ms"]); rays = st["rays"]
        # stream-side phases of the step (HIP events on the step's stream): render, gap + pack before the collective, the collective, unpack + resolve
        phases.append([ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), ev[2].elapsed_time(ev[3]), ev[3].elapsed_time(ev[4])])
    fence()
    dt = time.perf_counter() - t0

    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    r = torch.tensor([float(rays)], dtype=torch.float64, device="cuda")
    if collective:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
    dt = float(t.item()); rays_step = float(r.item())
    ph = torch.tensor(np.mean(phases, axis=0), dtype=torch.float64, device="cuda")
    per_rank = torch.tensor([float(rays), float(np.mean(kernel_ms))], dtype=torch.float64, device="cuda")
    if collective:
        gathered = [torch.zeros_like(per_rank) for _ in range(world)]
        dist.all_gather(gathered, per_rank)
        ph_max = ph.clone(); dist.all_reduce(ph_max, op=dist.ReduceOp.MAX)
    else:
        gathered, ph_max = [per_rank], ph

    if rehearsal and world > 1 and rank == 0:
        # the composed frame (N sample-range shards + all-reduce + resolve) against the same frame rendered by one context alone
        with torch.cuda.stream(stream):
            solo_acc = torch.zeros_like(accum); solo = torch.zeros_like(bgra)
            ctx.path_accumulate_device(s, cam, scenes.path_params(abi.RT_PATH_LIGHTS, SPP, seed=1), W, H, solo_acc.data_ptr(), sp)
            ctx.resolve_device(solo_acc.data_ptr(), W, H, float(SPP), 1, solo.data_ptr(), 0, None, sp)
        ctx.sync(sp)
        diff = (solo.to(torch.int16) - bgra.to(torch.int16)).abs()
        assert int(diff.max()) <= 1 and float((diff > 0).float().mean()) < 1e-3, "rehearsal: composed frame differs from the single-context frame"
        print(f"rehearsal: composed {world}-rank frame == single-context frame (max byte diff {int(diff.max())}, {float((diff > 0).float().mean()):.2e} of bytes differ)", file=sys.stderr)

    if rank == 0:
        form, unit_log2 = ctx.last_path_form()
        body = {0: "rt::path_body (one lane per pixel)", 1: "rt::path_queue_body", 2: f"rt::path_items_body (persistent waves, work counter, {1 << unit_log2} pixels per unit)", 3: "rt::path_sorted_body"}.get(form, "?")
        kernel_name = {2: f"k_path_jit (scene-specialised {body}<4, true>, hiprtc)", 1: f"generic {body}<4, true>", 0: "rt::k_path_tiled<4>"}.get(ctx.last_path_kernel()[0], "?")
        fpr = scenes.flops_per_ray(s, 100)                           # 17*3 + 12*5 + 38 + 100 = 249
        k_ms = float(np.mean(kernel_ms))                             # rank 0's tracing kernel (+slice sum), HIP events on its stream
        achieved = rays * fpr / (k_ms * 1e-3) / 1e12
        # roofline.traffic is NOT measured in this run: PMC counters need their own rocprofv3 passes (tools/collect_profiles.sh runs
        # this very command under --pmc FETCH_SIZE and --pmc WRITE_SIZE); the committed summary of those passes is quoted,
        # with its source, when it is for this workload - else null
        traffic = None
        pj = os.path.join(ROOT, "profiles", "r03_pmc_hbm_traffic.json")
        if world == 1 and os.path.exists(pj):
            try:
                pm = json.load(open(pj))
                if pm.get("workload") == f"{W}x{H}x{SPP}":
                    traffic = pm.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "rays/sec (whole node) on 04 path-tracing scene, 1/2/4/8 GPUs; % FP32 VALU peak",
            "value": rays_step * args.steps / dt, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"04_PathTracingLights scene (5 planes, 3 spheres, 1 emissive rectangle), {W}x{H}, {SPP} spp, max depth 6, seed 1",
                       "jit_compiler": jit_compiler, "samples_per_step": W * H * SPP, "rays_per_step": rays_step, "samples_per_sec": W * H * SPP * args.steps / dt,
                       "parallelism": (f"sample-range x{world}, one all-reduce(sum) of the packed RGB sums ({W * H * 12 / 1e6:.1f} MB)" + (" [REHEARSAL: all ranks on one GPU, gloo]" if rehearsal else "") + (" [ONE-RANK RCCL communicator: BENCH_ONE_RANK_COLLECTIVE=1]" if one_rank_collective else "")) if collective else "single GPU",
                       "rays_per_rank": [int(g[0].item()) for g in gathered], "kernel_ms_per_rank": [round(float(g[1].item()), 3) for g in gathered],
                       "phase_ms_rank0": {"render": round(float(ph[0]), 3), "pack": round(float(ph[1]), 3), "allreduce": round(float(ph[2]), 3), "unpack_resolve": round(float(ph[3]), 3)},
                       "phase_ms_max_over_ranks": {"render": round(float(ph_max[0]), 3), "pack": round(float(ph_max[1]), 3), "allreduce": round(float(ph_max[2]), 3), "unpack_resolve": round(float(ph_max[3]), 3)},
                       "host_sync_before_collective": bool(host_sync and collective)},
            "roofline": {"bound": "valu", "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_VALU_TFLOPS,
                         "traffic": traffic, "traffic_source": ("profiles/r03_pmc_hbm_traffic.json: separate rocprofv3 --pmc passes of this command (FETCH_SIZE, WRITE_SIZE), not this run; algorithmic bytes are 16 B/pixel = 132.7 MB" if traffic is not None else None),
                         "kernel": kernel_name, "kernel_ms": k_ms, "kernel_ms_steps": [round(x, 3) for x in kernel_ms], "flops_per_ray": fpr, "rays_per_launch": rays,
                         "note": "FP32 VALU-bound path (no MFMA, negligible HBM traffic): algorithmic FLOPs/ray x rays / kernel time vs the FP32 vector peak"},
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = min(16, os.cpu_count() or 1)
            out["cpu_baseline"] = cpu_baseline_reference(W, H, SPP, threads) or cpu_baseline(scenes, W, H, SPP, args.cpu_rows, threads)
        print(json.dumps(out), flush=True)
    if collective:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
