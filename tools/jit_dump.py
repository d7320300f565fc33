"""Generate + compile (hiprtc cross-compiles without a device) the scene-specialised kernel of a reference scene and leave source, code object and
disassembly under a prefix: python3 tools/jit_dump.py 04|03|02|01 [form: 16 item, 0 one lane per pixel, 8 queue, 32 sorted] [prefix]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
from rtamd import abi, lib, scenes
which = sys.argv[1] if len(sys.argv) > 1 else "04"
form = int(sys.argv[2]) if len(sys.argv) > 2 else 16
pre = sys.argv[3] if len(sys.argv) > 3 else f"/tmp/jit_{which}_{form}"
L = lib.load()
L.rt_debug_jit_probe.argtypes = [C.POINTER(abi.rt_scene), C.c_int32, C.c_char_p, C.c_int32]
note = C.create_string_buffer(4096)
sc, variant = {"04": (scenes.scene04(32)[0], 4), "03": (scenes.scene03(16)[0], 3), "02": (scenes.scene02(16)[0], 2), "01": (scenes.scene01()[0], 1)}[which]
os.environ["RT_MI355_JIT_DUMP"] = pre
rc = L.rt_debug_jit_probe(C.byref(sc.c()), variant + (form if variant > 2 else 0), note, 4096)
print("probe rc", rc, note.value.decode()[:300])
dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--mcpu=gfx950", pre + ".co"], capture_output=True, text=True).stdout
open(pre + ".s", "w").write(dis)
print("wrote", pre + ".s", len(dis.splitlines()), "lines")
