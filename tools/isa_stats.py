#!/usr/bin/env python
"""Per-kernel instruction statistics of a gfx950 assembly listing (hipcc -S --cuda-device-only):
VGPR/SGPR/LDS use and counts of memory / LDS / barrier instructions.  usage: isa_stats.py file.s [filter]"""
import re
import sys
from collections import Counter

text = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)^\s+\.end_amdhsa_kernel", text, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name:
        continue
    ops = Counter(re.findall(r"^\s+(global_load_\w+|global_store_\w+|global_atomic_\w+|ds_\w+|buffer_\w+|s_barrier|scratch_\w+)", body, re.M))
    vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", body)
    lds = re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", body)
    ninstr = len(re.findall(r"^\s+[vs]_\w+|^\s+ds_|^\s+global_|^\s+buffer_", body, re.M))
    print("%s\n   vgpr %s lds %s instrs %d  %s" % (name, vg.group(1) if vg else "?", lds.group(1) if lds else "?", ninstr,
                                                  " ".join("%s:%d" % kv for kv in sorted(ops.items()))))
