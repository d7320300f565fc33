"""Which primitive of a fuzz scene makes the GPU's scene-query count leave the oracle's: python3 tools/fuzz_path_diag.py seed [seed ...]
Renders the scene whole, then with each primitive (after the first) left out, and with every material made diffuse."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from rtamd import abi, lib, scenes
import orc as oracle
src = open(os.path.join(ROOT, "tools", "fuzz_path.py")).read().replace("sys.exit(main())", "")
exec(compile(src, "fuzz_path", "exec"))
ctx = lib.Context()
W, H, SPP = 96, 64, 64
NAMES = {abi.RT_PRIM_SPHERE: "sphere", abi.RT_PRIM_PLANE: "plane", abi.RT_PRIM_RECT: "rect", abi.RT_PRIM_BOX: "box", abi.RT_PRIM_TRIANGLE: "triangle"}
MATS = {abi.RT_MAT_DIFFUSE: "diffuse", abi.RT_MAT_METAL: "metal", abi.RT_MAT_DIELECTRIC: "dielectric", abi.RT_MAT_LIGHT: "light"}
def rel(s, cam, prm):
    ctx.set_debug_flags(8)
    try: g = ctx.render_path(s, cam, prm, W, H)
    finally: ctx.set_debug_flags(0)
    o = oracle.render_path(s, cam, prm, W, H, threads=16)
    return (g["rays"] - o["rays"]) / o["rays"], o["rays"] / (W * H * SPP)
for seed in map(int, sys.argv[1:]):
    s, cam, variant = FAMILIES[os.environ.get("FUZZ_FAMILY", "")](seed)
    prm = scenes.path_params(variant, SPP, seed=seed, **({"t_max": float(os.environ["FUZZ_TMAX"])} if os.environ.get("FUZZ_TMAX") else {}), **({"t_min": float(os.environ["FUZZ_TMIN"])} if os.environ.get("FUZZ_TMIN") else {}), **({"max_depth": int(os.environ["FUZZ_DEPTH"])} if os.environ.get("FUZZ_DEPTH") else {}))
    r, q = rel(s, cam, prm)
    print(f"seed {seed}: whole scene rel q {r:+.2e} (oracle {q:.4f} queries/sample); camera {list(cam.position)} {list(cam.orientation_deg)} fov {cam.fov_deg:.1f}")
    prims = list(s.prims)
    for i, p in enumerate(prims):
        m = s.materials[p.material]
        desc = f"{NAMES[p.type]} {[round(float(x), 3) for x in p.f[:9]]} flags {p.flags} {MATS[m.type]} rough {m.roughness:.2f} ior {m.refraction_index:.3f}"
        if i == 0: print(f"  prim 0 kept: {desc}"); continue
        s.prims = prims[:i] + prims[i + 1:]; s._c = None
        r, q = rel(s, cam, prm)
        print(f"  without prim {i}: rel q {r:+.2e}   ({desc})", flush=True)
    s.prims = prims; s._c = None
