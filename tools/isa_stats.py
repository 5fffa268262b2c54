#!/usr/bin/env python3
"""Static instruction mix of kernels in a hipcc -S listing: python tools/isa_stats.py file.s [symbol-substring ...]"""
import re
import sys


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    txt = open(path).read()
    # kernel bodies: from 'sym:' to s_endpgm; metadata (.vgpr_count etc.) from the .amdhsa block / comments
    for m in re.finditer(r"^(_Z\w+):\s*; @\1\n(.*?)\n\s*\.end_amdhsa_kernel", txt, re.S | re.M):
        sym, body = m.group(1), m.group(2)
        if pats and not any(p in sym for p in pats):
            continue
        code = body.split(".section")[0]
        ins = [l.strip().split()[0] for l in code.splitlines() if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";"))]
        valu = [i for i in ins if i.startswith("v_")]
        pk = [i for i in valu if i.startswith("v_pk_")]
        salu = [i for i in ins if i.startswith("s_")]
        vmem = [i for i in ins if i.startswith(("global_", "buffer_", "flat_", "scratch_"))]
        lds = [i for i in ins if i.startswith("ds_")]
        vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", body)
        sc = re.search(r"; ScratchSize: (\d+)", txt[m.end():m.end() + 4000])
        occ = re.search(r"; Occupancy: (\d+)", txt[m.end():m.end() + 4000])
        print("%-90s VALU %5d (pk %4d)  SALU %5d  VMEM %4d  LDS %4d  vgpr %s scratch %s occ %s" % (
            sym[:90], len(valu), len(pk), len(salu), len(vmem), len(lds), vg.group(1) if vg else "?", sc.group(1) if sc else "?",
            occ.group(1) if occ else "?"))


if __name__ == "__main__":
    main()
