#!/usr/bin/env python3
"""Print one kernel's body from a device assembly file (tools/kernel_isa.sh): tools/isa_fn.py FILE.s MANGLED_SUBSTRING [regex filter]"""
import re, sys
f, key = sys.argv[1], sys.argv[2]
flt = re.compile(sys.argv[3]) if len(sys.argv) > 3 else None
on = False; n = 0
for line in open(f):
    if not on:
        if line.startswith("_ZN") and key in line and line.rstrip().split(":")[0].endswith(key) or (line.startswith("_ZN") and key in line.split(":")[0]):
            on = True; print(line.rstrip()[:160])
        continue
    if line.startswith(".Lfunc_end"): break
    s = line.strip()
    if not s or s.startswith(";"): continue
    n += 1
    if flt is None or flt.search(s): print(f"{n:5d}  {s[:150]}")
