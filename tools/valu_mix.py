#!/usr/bin/env python3
"""The VALU instruction mix of tk_flat_kernel by issue-cost class (static: the kernel's assembly, no GPU needed), and -- with the
micro-benchmark's table -- the kernel's VALU-issue floor:

  python tools/valu_mix.py                         -> profiles/r04_valu_mix.json   (classes as tools/ubench/valu2.hip measures them)
  python tools/valu_mix.py --clk profiles/ubench/r04_valu2.json --insts 313205612 [--waves 7]
      floor_ms = insts_per_launch x sum_c(share_c x clk_c) / (SIMDs x clock); the clock comes from the micro-benchmark itself
      (s_memtime ticks per event-timed millisecond).

The dynamic counter (SQ_INSTS_VALU) gives the total; the split between classes is the STATIC one of the kernel's code, which is
what there is without per-opcode counters -- the hot loops (classification, rules, the per-batch look-up) dominate both."""
import argparse
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CLASSES = [  # (class name as in valu2.hip, regex on the mnemonic / full line)
    ("v_mov_b32_dpp (wave_shr / wave_shl)", r"^v_mov_b32_dpp"),
    ("v_add_u32_dpp (row_shr / row_bcast)", r"_dpp\b"),
    ("sdwa", r"_sdwa\b"),
    ("v_readlane / v_readfirstlane", r"^v_(readlane|readfirstlane|writelane)_b32"),
    ("v_mad_u64_u32", r"^v_mad_[ui]64_[ui]32"),
    ("v_mul_lo_u32", r"^v_mul_(lo|hi)_[ui]32"),
    ("64-bit shift / v_lshl_add_u64", r"^v_(lshlrev|lshrrev|ashrrev)_[bi]64|^v_lshl_add_u64|^v_(add|sub)_(co_)?u64"),
    ("v_alignbit_b32 / v_alignbyte_b32", r"^v_align(bit|byte)_b32"),
    ("v_perm_b32", r"^v_perm_b32"),
    ("v_bitop3_b32", r"^v_bitop3_b32"),
    ("v_bcnt / v_mbcnt / v_ffbl", r"^v_(bcnt|mbcnt|ffbl|ffbh)_"),
    ("v_min3 / v_max3 / v_min", r"^v_(min|max)3?_[ui](16|32)"),
    ("v_cmp -> sgpr pair", r"^v_cmp\w*_e64"),
    ("v_cmp (vcc) + v_cndmask", r"^v_cmp|^v_cndmask"),
    ("alu32 three-operand (and_or / add3 / bfe / lshl_or: 64-bit encoding)", r"^v_(and_or|or3|add3|xad|lshl_or|lshl_add|add_lshl|bfe|bfi|xor3|mad_u32_u24|mad_i32_i24|sad)_|_e64$"),
    ("alu32 (add / xor / and_or / shift)", r"^v_"),
]


def kernel_body(asm_path, name="_Z14tk_flat_kernel10TkFlatArgs"):
    lines = open(asm_path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.section") or "uses_flat_scratch" in lines[i])
    return [l.strip() for l in lines[start + 1:end] if l.strip() and not l.strip().startswith((".", ";"))]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clk", default="")
    ap.add_argument("--insts", type=float, default=0.0)
    ap.add_argument("--waves", default="7")
    ap.add_argument("--kernel-ms", type=float, default=0.0)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_valu_mix.json"))
    a = ap.parse_args()
    asm = os.path.join(ROOT, "gpurun_out", "asm", "flat.s")
    os.makedirs(os.path.dirname(asm), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", "-o", asm,
                           os.path.join(ROOT, "tekken-rs_amd", "csrc", "tk_flat.hip")], stderr=subprocess.DEVNULL)
    body = kernel_body(asm)
    mix = collections.Counter()
    for l in body:
        if not l.startswith("v_"):
            continue
        mn = l.split()[0]
        probe = mn + (" _dpp" if (" row_" in l or " wave_" in l or "quad_perm" in l) and "_dpp" not in mn else "") + (" _sdwa" if "_sel:" in l and "_sdwa" not in mn else "")
        for cname, rx in CLASSES:
            if re.search(rx, mn) or (rx in (r"_dpp\b", r"_sdwa\b") and re.search(rx, probe)):
                mix[cname] += 1
                break
    total = sum(mix.values())
    res = {"kernel": "tk_flat_kernel", "static_valu_instructions": total, "other_instructions": {"salu": sum(1 for l in body if l.startswith("s_")),
           "lds": sum(1 for l in body if l.startswith("ds_")), "vmem": sum(1 for l in body if l.startswith(("global_", "buffer_", "flat_", "scratch_")))},
           "classes": {k: {"count": v, "share": round(v / total, 4)} for k, v in mix.most_common()},
           "note": "static mix of the kernel's code (tools/valu_mix.py); SQ_INSTS_VALU gives the dynamic total, not the split"}
    if a.clk:
        ub = json.load(open(a.clk))["classes"]
        w = a.waves
        clk_mix, rows = 0.0, {}
        for k, v in res["classes"].items():
            # (the cheaper of the two well-occupied points: this is a FLOOR; the 7-wave point alone is noisy -- its blocks are not all
            # resident at once, see resident_share in the micro-benchmark's output)
            e = ub.get(k, ub["alu32 (add / xor / and_or / shift)"])
            c = min(e["4"]["clk"], e[w]["clk"])
            rows[k] = c
            clk_mix += v["share"] * c
        ghz = sorted(x[w]["shader_GHz"] for x in ub.values())[len(ub) // 2]
        res["issue_cost"] = {"waves_per_simd": int(w), "clk_per_class": rows, "clk_mix": round(clk_mix, 3), "shader_GHz": round(ghz, 3),
                             "source": os.path.relpath(a.clk, ROOT), "guide": "MI355X_MICROARCH.md: SIMD-32, a wave64 VALU instruction issues over 2 cycles; one wave alone sustains 4"}
        if a.insts:
            floor = a.insts * clk_mix / 1024.0 / (ghz * 1e9) * 1e3
            res["floor"] = {"valu_insts_per_launch": a.insts, "floor_ms": round(floor, 4)}
            if a.kernel_ms:
                res["floor"]["kernel_ms"] = a.kernel_ms
                res["floor"]["frac"] = round(floor / a.kernel_ms, 4)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
