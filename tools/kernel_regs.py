#!/usr/bin/env python3
"""Prints VGPR / spill / scratch figures and the static VALU count of every kernel in a gfx950 assembly file
(hipcc -S --cuda-device-only).  Usage: tools/kernel_regs.py step.s"""
import re
import sys

s = open(sys.argv[1]).read()
meta = {}
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:", s, re.S):
    blk = m.group(0)
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    get = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
    meta[name] = (get("vgpr_count"), get("vgpr_spill_count"), get("private_segment_fixed_size"), get("sgpr_count"))
for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)s_endpgm", s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    valu = sum(1 for l in body.split("\n") if l.strip().startswith("v_"))
    v = meta.get(name, ("?",) * 4)
    print("%-70s vgpr %s spill %s scratch %s B sgpr %s valu %d" % (name[:70], v[0], v[1], v[2], v[3], valu))
