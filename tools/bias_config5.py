"""HIP vs the oracle on the config 5 scene (18,194 primitives, 04 estimator) at equal seeds: scene queries per sample and mean colour."""
import os, sys, time
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from rtamd import abi, lib, scenes
import orc as oracle
ctx = lib.Context()
W, H = 196, 140
for spp, seed in ((32, 1), (32, 2), (32, 3)):
    s, cam, _ = scenes.scene_stress(spp=spp, seed=seed)
    prm = scenes.path_params(abi.RT_PATH_LIGHTS, spp, max_depth=8, seed=seed)
    t0 = time.time(); g = ctx.render_path(s, cam, prm, W, H); o = oracle.render_path(s, cam, prm, W, H, threads=16)
    n = W * H * spp
    print(f"config 5 scene, {spp} spp seed {seed}: HIP {g['rays'] / n:.5f} oracle {o['rays'] / n:.5f} rel {g['rays'] / o['rays'] - 1:+.2e}; mean HIP {g['rgb'].mean():.6f} oracle {(o['sum'] / spp).mean():.6f} ({time.time() - t0:.0f} s)", flush=True)
