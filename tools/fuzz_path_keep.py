"""A fuzz scene reduced to chosen primitives: python3 tools/fuzz_path_keep.py seed i,j,k [i,j,k ...]   (FUZZ_FAMILY / FUZZ_ARITH as in fuzz_path.py)"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from rtamd import abi, lib, scenes
import orc as oracle
src = open(os.path.join(ROOT, "tools", "fuzz_path.py")).read().replace("sys.exit(main())", "")
exec(compile(src, "fuzz_path", "exec"))
ctx = lib.Context()
if os.environ.get("FUZZ_ARITH") == "ref": ctx.set_option(abi.RT_OPT_ARITHMETIC, abi.RT_ARITH_REFERENCE)
W, H, SPP = 96, 64, int(os.environ.get("FUZZ_SPP", "64"))
seed = int(sys.argv[1])
s, cam, variant = FAMILIES[os.environ.get("FUZZ_FAMILY", "")](seed)
prims = list(s.prims)
for keep in sys.argv[2:]:
    idx = [int(k) for k in keep.split(",")]
    s.prims = [prims[i] for i in idx]; s._c = None
    prm = scenes.path_params(variant, SPP, seed=seed, **({"t_max": float(os.environ["FUZZ_TMAX"])} if os.environ.get("FUZZ_TMAX") else {}), **({"t_min": float(os.environ["FUZZ_TMIN"])} if os.environ.get("FUZZ_TMIN") else {}), **({"max_depth": int(os.environ["FUZZ_DEPTH"])} if os.environ.get("FUZZ_DEPTH") else {}))
    ctx.set_debug_flags(8 | int(os.environ.get("FUZZ_XFLAGS", "0")))
    try: g = ctx.render_path(s, cam, prm, W, H)
    finally: ctx.set_debug_flags(0)
    o = oracle.render_path(s, cam, prm, W, H, threads=16)
    print(f"seed {seed} prims {keep}: rel q {(g['rays'] - o['rays']) / o['rays']:+.2e} (oracle {o['rays'] / (W * H * SPP):.4f} q/s)", flush=True)
