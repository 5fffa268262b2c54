#!/usr/bin/env python3
"""Estimated VALU issue time of a kernel listing under the measured per-instruction costs (tools/ubench/inst_cost.hip, MI355X,
7 waves/SIMD): python tools/isa_cost.py file.s symbol-substring [first_line last_line]"""
import re
import sys
from collections import defaultdict

FAST = ("v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_add_u32", "v_sub_u32",
        "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b64")
SLOW = {"v_rcp_f32": 8.5, "v_rcp_iflag_f32": 8.5, "v_sqrt_f32": 8.6, "v_rsq_f32": 8.5, "v_permlane32_swap_b32": 8.1,
        "v_permlane16_swap_b32": 8.1, "v_rcp_f64": 17, "v_sqrt_f64": 17, "v_div_scale_f64": 8, "v_div_fmas_f64": 8, "v_div_fixup_f64": 8}


def cost(line):
    op = line.split()[0]
    base = re.sub(r"_e32$|_e64$|_dpp$|_sdwa$", "", op)
    if base in SLOW:
        return SLOW[base]
    if base.startswith("v_pk_"):
        return 4.9
    if base.endswith("_f64") or "_f64_" in base:
        return 5.1
    sgpr = re.search(r"[ ,]s\[?\d", line) or "vcc" in line or "exec" in line
    dpp = op.endswith("_dpp")
    if base in FAST and not sgpr and not dpp:
        return 3.5 if base == "v_fmac_f32" else 2.9
    return 4.4


def main():
    path, pat = sys.argv[1], sys.argv[2]
    txt = open(path).read()
    m = re.search(r"^(_Z\w*%s\w*):\s*; @\1\n(.*?)\n\s*\.end_amdhsa_kernel" % re.escape(pat), txt, re.S | re.M)
    lines = m.group(2).split(".section")[0].splitlines()
    lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    hi = int(sys.argv[4]) if len(sys.argv) > 4 else len(lines)
    if len(sys.argv) <= 3:  # default: the straight-line hot block = everything before the first loop header
        for i, l in enumerate(lines):
            if "Loop Header: Depth=1" in l:
                hi = i
                break
    tot = defaultdict(float)
    cnt = defaultdict(int)
    for l in lines[lo:hi]:
        s = l.strip()
        if not s.startswith("v_"):
            continue
        c = cost(s)
        op = s.split()[0]
        key = op + (" (sgpr)" if c == 4.4 and re.sub(r"_e32$|_e64$", "", op) in FAST else "")
        tot[key] += c
        cnt[key] += 1
    total = sum(tot.values())
    print("%s lines %d..%d: %d VALU, estimated %.0f SIMD cycles" % (m.group(1)[:60], lo, hi, sum(cnt.values()), total))
    for k in sorted(tot, key=lambda k: -tot[k])[:30]:
        print("   %-34s %4d  %7.0f  %4.1f%%" % (k, cnt[k], tot[k], 100 * tot[k] / total))


if __name__ == "__main__":
    main()
