"""03-type fuzz scenes on the large ground sphere, generic kernel only: python3 tools/fuzz_path_big.py seed ...   (seed 0 = the reference's 03 scene)"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from rtamd import abi, lib, scenes
import orc as oracle
src = open(os.path.join(ROOT, "tools", "fuzz_path.py")).read().replace("sys.exit(main())", "")
exec(compile(src, "fuzz_path", "exec"))
ctx = lib.Context()
if os.environ.get("FUZZ_ARITH") == "ref": ctx.set_option(abi.RT_OPT_ARITHMETIC, abi.RT_ARITH_REFERENCE)      # the reference's arithmetic for rays and spheres
W, H, SPP = 96, 64, 64
flags = int(os.environ.get("FUZZ_FLAGS", "8"))
tot_g = tot_o = 0
for seed in map(int, sys.argv[1:]):
    if seed == 0: s, cam, prm = scenes.scene03(SPP, seed=7)
    else:
        s, cam, variant = FAMILIES[os.environ.get("FUZZ_FAMILY", "")](seed); prm = scenes.path_params(variant, SPP, seed=seed, **({"t_max": float(os.environ["FUZZ_TMAX"])} if os.environ.get("FUZZ_TMAX") else {}), **({"t_min": float(os.environ["FUZZ_TMIN"])} if os.environ.get("FUZZ_TMIN") else {}), **({"max_depth": int(os.environ["FUZZ_DEPTH"])} if os.environ.get("FUZZ_DEPTH") else {}))
    ctx.set_debug_flags(flags)
    try: g = ctx.render_path(s, cam, prm, W, H)
    finally: ctx.set_debug_flags(0)
    o = oracle.render_path(s, cam, prm, W, H, threads=16)
    tot_g += g["rays"]; tot_o += o["rays"]
    mo = (o["sum"] / SPP).reshape(-1, 3).mean(axis=0); mg = g["rgb"].reshape(-1, 3).mean(axis=0)
    print(f"seed {seed:3d}: rel q {(g['rays'] - o['rays']) / o['rays']:+.2e}  mean rel {np.abs(mg / mo - 1).max():.1e}", flush=True)
print(f"all: rel q {(tot_g - tot_o) / tot_o:+.2e}")
