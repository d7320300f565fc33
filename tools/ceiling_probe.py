"""Ceiling probe of the headline kernel: the shipped bounce loop's instruction stream with every lane alive.

python3 tools/ceiling_probe.py real|uniform [spp]      (under rocprofv3 --pmc for the counters; tools/ceiling_probe.sh runs both)

real    : BASELINE config 4's scene as it is (three material classes + the light, early path ends at the light)
uniform : the same nine primitives, same camera, but ONE material class - every surface diffuse, the light rectangle too.  No path
          ends early (nothing emits), so every lane of every wave runs exactly max_depth + 1 scene queries per sample, all 64 lanes
          regenerate in the same iteration, and shading takes one branch.  Same generated scene query (same geometry), same
          shading code for the diffuse class, same regeneration: its cycles per 64 rays are what this instruction mix costs when
          nothing diverges - the bound for any scheduling / compaction scheme that keeps the arithmetic."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
os.environ.setdefault("RT_MI355_JIT", "2")
from rtamd import abi, lib, scenes

which = sys.argv[1] if len(sys.argv) > 1 else "real"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
s, cam, prm = scenes.scene04(spp)
if which == "uniform":
    u = scenes.Scene()
    white = u.diffuse((0.73, 0.73, 0.73))
    for p in s.prims:
        q = abi.rt_prim(type=p.type, material=white, flags=p.flags)
        for i in range(len(p.f)):
            q.f[i] = p.f[i]
        u.prims.append(q)
    s = u
ctx = lib.Context()
W, H = 3840, 2160
for rep in range(3):
    r = ctx.render_path(s, cam, prm, W, H)
    print(f"{which} {W}x{H} spp {spp} kernel {ctx.last_path_kernel()[0]}: {r['kernel_ms']:.2f} ms rays {r['rays']} rays/sample {r['rays'] / (W * H * spp):.4f} "
          f"ns/kray {r['kernel_ms'] * 1e6 / r['rays'] * 1e3:.3f} frac(249) {r['rays'] * 249 / (r['kernel_ms'] * 1e-3) / 157.3e12:.4f}", flush=True)
