"""The N>1 composition (sample-range shards + one all-reduce(sum) + resolve), rehearsed with world_size 2 on the
gloo backend.  No GPU here, so each rank's shard is rendered by the CPU oracle standing in for the kernel; what is
under test is the partition, the collective and the resolve order that bench.py uses on RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
from rtamd import abi, scenes, shard
import orc as oracle

W, H, SPP = 64, 40, 7      # odd spp: ragged ranges


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s, cam, _ = scenes.scene04(SPP, seed=3)
    b, e = shard.sample_range(rank, world, SPP)
    prm = scenes.path_params(abi.RT_PATH_LIGHTS, SPP, seed=3, sample_begin=b, sample_end=e)
    r = oracle.render_path(s, cam, prm, W, H)
    acc = torch.zeros(H, W, 4)
    acc[:, :, :3] = torch.from_numpy(r["sum"]); acc[:, :, 3] = e - b
    dist.all_reduce(acc, op=dist.ReduceOp.SUM)
    rays = torch.tensor([float(r["rays"])], dtype=torch.float64)
    dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.save(os.path.join(out_dir, "acc.npy"), acc.numpy()); np.save(os.path.join(out_dir, "rays.npy"), rays.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_ranges_cover_every_sample_once():
    for world in (1, 2, 3, 4, 8):
        for spp in (0, 1, 7, 32, 1024, 1000):
            r = shard.all_ranges(world, spp)
            full = [x for x in r if not shard.is_empty(x)]
            assert sum(e - b for b, e in r) == spp and all(a[1] == b[0] for a, b in zip(full, full[1:]))
            assert (not full) or (full[0][0] == 0 and full[-1][1] == spp)
            assert max(e - b for b, e in r) - min(e - b for b, e in r) <= 1
            # rt_path_params reads (0, 0) as "the whole frame": no share may look like that unless it IS the whole frame
            assert all(x != (0, 0) or spp == 0 for x in r)
    with pytest.raises(ValueError):
        shard.sample_range(2, 2, 8)


def test_two_ranks_reproduce_the_single_process_frame(tmp_path):
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    acc = np.load(tmp_path / "acc.npy"); rays = float(np.load(tmp_path / "rays.npy")[0])
    s, cam, prm = scenes.scene04(SPP, seed=3)
    full = oracle.render_path(s, cam, prm, W, H)
    assert rays == full["rays"]
    assert (acc[:, :, 3] == SPP).all()
    # only the float summation order differs (sum of 2 partial sums vs one running sum)
    assert np.allclose(acc[:, :, :3], full["sum"], rtol=2e-6, atol=1e-6)
    a = oracle.resolve(acc[:, :, :3], float(SPP), 1); b = oracle.resolve(full["sum"], float(SPP), 1)
    assert (np.abs(a.astype(int) - b.astype(int)) <= 1).all() and (a == full["bgra"]).mean() > 0.999


def _bench(args, env_extra):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_bench_refuses_a_launch_that_does_not_match_gpus():
    """bench.py must not fold an N-GPU request onto fewer ranks, and has no CPU path to time: non-zero exit and a message, no JSON line."""
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {})                      # started bare: WORLD_SIZE is 1
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and "--nproc-per-node 2" in r.stderr and r.stdout.strip() == ""
    r = _bench(["--gpus", "1", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and r.stdout.strip() == ""
    if not torch.cuda.is_available():
        r = _bench(["--steps", "1", "--warmup", "0"], {})
        assert r.returncode != 0 and "needs an MI355X" in r.stderr and r.stdout.strip() == ""
