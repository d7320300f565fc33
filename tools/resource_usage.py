"""Kernel resource-usage report (no GPU needed): the ahead-of-time kernels from hipcc's -Rpass-analysis=kernel-resource-usage
('make -C software-raytracing_amd asm') and the scene-specialised kernels of the four reference scenes, generated and compiled by
rt_debug_jit_probe (hiprtc cross-compiles without a device) and read from the code object's notes.
python3 tools/resource_usage.py > profiles/rNN_kernel_resource_usage.txt"""
import ctypes as C, os, re, subprocess, sys, tempfile
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
PKG = os.path.join(ROOT, "software-raytracing_amd")
sys.path.insert(0, PKG)
subprocess.run(["make", "-C", PKG, "asm"], check=True, capture_output=True)
txt = open(os.path.join(PKG, "build", "resource_usage.txt")).read()
print("# Kernel resource usage (gfx950). Ahead-of-time kernels of librt_mi355.so: hipcc -Rpass-analysis=kernel-resource-usage ('make asm');")
print("# scene-specialised kernels (hiprtc, via rt_debug_jit_probe): the code objects' metadata.  tools/resource_usage.py regenerates this file.")
print("# columns: kernel | SGPRs | VGPRs | scratch bytes/lane | waves/SIMD | static LDS bytes/block (dynamic LDS is set at launch)")
cur = {}
for l in txt.splitlines():
    m = re.search(r"remark: .*?Function Name: (\S+)", l)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")}
        continue
    for key in ("TotalSGPRs", "VGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
        m = re.search(r"remark: .*?\s" + re.escape(key) + r": (\d+)", l)
        if m and cur and key not in cur:
            cur[key] = m.group(1)
            if key == "LDS Size [bytes/block]":
                print(" | ".join([cur["name"]] + [f"{k}: {cur.get(k, '?')}" for k in ("TotalSGPRs", "VGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]")]))
                cur = {}
from rtamd import abi, lib, scenes
L = lib.load()
L.rt_debug_jit_probe.argtypes = [C.POINTER(abi.rt_scene), C.c_int32, C.c_char_p, C.c_int32]
note = C.create_string_buffer(4096)
print("# scene-specialised builds (variant + form: 16 item form, 0 one lane per pixel, 8 queue, 32 outcome-sorted)")
with tempfile.TemporaryDirectory() as tmp:
    for name, sc, variant in (("04 scene, item form", scenes.scene04(32)[0], 4 + 16), ("04 scene, one lane per pixel", scenes.scene04(32)[0], 4), ("04 scene, queue form", scenes.scene04(32)[0], 4 + 8),
                              ("04 scene, outcome-sorted form", scenes.scene04(32)[0], 4 + 32), ("03 scene, item form", scenes.scene03(16)[0], 3 + 16), ("03 scene, one lane per pixel", scenes.scene03(16)[0], 3),
                              ("02 scene (k_whitted_jit + k_whitted_shadow_jit)", scenes.scene02(16)[0], 2)):
        pre = os.path.join(tmp, "k")
        os.environ["RT_MI355_JIT_DUMP"] = pre
        L.rt_debug_jit_probe(C.byref(sc.c()), variant, note, 4096)
        meta = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", pre + ".co"], capture_output=True, text=True).stdout
        for blk in meta.split(".agpr_count")[1:]:
            g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
            v = int(g("vgpr_count"))
            print(f"{g('name')} [{name}] | TotalSGPRs: {g('sgpr_count')} | VGPRs: {v} | ScratchSize [bytes/lane]: {g('private_segment_fixed_size')} | spilled VGPRs: {g('vgpr_spill_count')} | "
                  f"Occupancy [waves/SIMD]: {min(8, 512 // (8 * ((v + 7) // 8)))} | LDS Size [bytes/block]: {g('group_segment_fixed_size')}")
        os.remove(pre + ".co")
