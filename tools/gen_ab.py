"""Generated-query generations on the headline frame (item form): python3 tools/gen_ab.py [spp]"""
import os, sys, subprocess
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if len(sys.argv) > 2:
    sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
    os.environ["RT_MI355_JIT"] = "2"; os.environ["RT_MI355_JIT_GEN"] = sys.argv[2]
    from rtamd import lib, scenes
    ctx = lib.Context(); spp = int(sys.argv[1])
    s, cam, prm = scenes.scene04(spp)
    for rep in range(3):
        r = ctx.render_path(s, cam, prm, 3840, 2160)
        print(f"gen {sys.argv[2]} spp {spp}: {r['kernel_ms']:.2f} ms rays {r['rays']} ns/kray {r['kernel_ms'] * 1e6 / r['rays'] * 1e3:.3f} frac {r['rays'] * 249 / (r['kernel_ms'] * 1e-3) / 157.3e12:.4f} mean {r['rgb'].mean():.9f}", flush=True)
else:
    for g in ("1", "2", "3", "1", "3"):
        subprocess.run([sys.executable, __file__, sys.argv[1] if len(sys.argv) > 1 else "256", g], check=True)
