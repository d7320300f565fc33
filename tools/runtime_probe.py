import os, sys
sys.path.insert(0, "software-raytracing_amd")
mode = sys.argv[1]
os.environ["RT_MI355_JIT"] = "2"; os.environ["RT_MI355_JIT_DUMP"] = f"gpurun_out/jit_{mode}"
if mode == "torch":
    import torch; torch.cuda.init()
if mode == "torch+comgr":      # the system compiler library made the process-wide provider of amd_comgr_* before torch's runtime loads its own copy
    import ctypes
    ctypes.CDLL("/opt/rocm/lib/libamd_comgr.so.3", mode=ctypes.RTLD_GLOBAL)
    import torch; torch.cuda.init()
    x = torch.ones(1024, device="cuda"); print("torch works:", float((x * 2).sum()))
from rtamd import abi, lib, scenes
ctx = lib.Context()
s, cam, prm = scenes.scene04(1024, seed=1)
best = min(ctx.render_path(s, cam, prm, 3840, 2160)["kernel_ms"] for _ in range(3))
import hashlib
print(mode, f"{best:.2f} ms", hashlib.md5(open(f"gpurun_out/jit_{mode}.co", "rb").read()).hexdigest(), flush=True)
for l in open("/proc/self/maps"):
    if any(k in l for k in ("libamdhip64", "libhiprtc", "libamd_comgr", "libhsa-runtime")) and "r-xp" in l: print("   ", l.split()[-1])
