#!/usr/bin/env python3
"""Registers, spills, LDS and scratch of every kernel in a gfx950 assembly listing (hipcc --cuda-device-only -S):  python tools/kernel_resources.py kernels.s [filter]"""
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size: *\d+", txt, re.S):
    blk = m.group(0)
    g = lambda k: (re.search(r"\." + k + r": *(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    short = re.sub(r"^_ZN3crh12_GLOBAL__N_1\d+", "", name)[:70]
    if flt in name:
        print(f"{short:72s} vgpr {g('vgpr_count'):>4s} agpr {g('agpr_count'):>3s} sgpr {g('sgpr_count'):>4s} spill {g('vgpr_spill_count'):>3s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>5s}")
