#!/usr/bin/env python3
"""ISA histogram of a code object (llvm-objdump -d output): instruction counts by opcode and encoding class, with the
issue cost of each class from tools/ubench (cycles per wave-instruction per SIMD, normalised to 2.4 GHz), per address range.

usage: isa_hist.py file.s [name=lo-hi ...]   (lo/hi = instruction line numbers in the listing, 1-based, inclusive)"""
import re
import sys
from collections import Counter

FULL = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32", "v_xor_b32", "v_and_b32", "v_or_b32",
        "v_lshrrev_b32", "v_lshlrev_b32", "v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_cndmask_b32", "v_ashrrev_i32", "v_not_b32"}
TRANS = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_exp_f32", "v_log_f32", "v_rcp_iflag_f32"}
COST = {"full": 2.45, "vop3_fma": 3.0, "half": 4.3, "cmp": 4.6, "trans": 8.3, "salu": 0.0, "lds": 0.0, "vmem": 0.0, "other": 0.0}


def classify(op, text):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if not op.startswith("v_"):
        return "other"
    if base in TRANS:
        return "trans"
    if base.startswith("v_cmp"):
        return "cmp"
    sgpr_operand = re.search(r"(?<![a-z_\[])s\d+|s\[\d+:\d+\]", text.split(",", 1)[1] if "," in text else "") is not None
    if op.endswith("_sdwa") or op.endswith("_dpp"):
        return "half"
    if base in FULL and not op.endswith("_e64"):
        return "half" if sgpr_operand and base != "v_cndmask_b32" else "full"
    if base in ("v_fma_f32",) or (base in FULL and op.endswith("_e64")):
        return "half" if sgpr_operand else "vop3_fma"
    return "half"


def main():
    lines = [l for l in open(sys.argv[1]) if re.match(r"^\s+[a-z_0-9]+\s", l) and "//" in l]
    ranges = {"all": (1, len(lines))}
    for a in sys.argv[2:]:
        n, r = a.split("=")
        lo, hi = r.split("-")
        ranges[n] = (int(lo), int(hi))
    for name, (lo, hi) in ranges.items():
        cls, ops = Counter(), Counter()
        for l in lines[lo - 1:hi]:
            t = l.split("//")[0].strip()
            op = t.split()[0]
            c = classify(op, t)
            cls[c] += 1
            ops[(op, c)] += 1
        valu = sum(v for k, v in cls.items() if k in ("full", "vop3_fma", "half", "cmp", "trans"))
        cyc = sum(COST[k] * v for k, v in cls.items())
        print(f"== {name}: instructions {lo}..{hi}: {hi - lo + 1} total, {valu} VALU, modelled issue cost {cyc:.0f} cycles@2.4GHz")
        print("   by class: " + ", ".join(f"{k} {v}" for k, v in sorted(cls.items(), key=lambda kv: -kv[1])))
        print("   by opcode: " + ", ".join(f"{op}[{c}] {v}" for (op, c), v in ops.most_common(40)))


if __name__ == "__main__":
    main()
