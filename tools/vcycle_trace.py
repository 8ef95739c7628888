#!/usr/bin/env python3
"""Per-launch view of one V-cycle from a rocprofv3 --kernel-trace of tools/vcycle_bench.py: mean duration by (kernel, grid size),
i.e. per LEVEL, and the gaps between consecutive launches.   python tools/vcycle_trace.py <dir with *_kernel_trace.csv>"""
import csv, glob, os, sys, collections
if len(sys.argv) < 2 or not os.path.isdir(sys.argv[1]):
    sys.exit("usage: python tools/vcycle_trace.py <directory holding a rocprofv3 --kernel-trace of tools/vcycle_bench.py (*_kernel_trace.csv)>")
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
if not rows:
    sys.exit(f"no *kernel_trace.csv under {sys.argv[1]}")
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("void ", "").replace("mgcfd::exact::", "").replace("mgcfd::fast::", "")
    return n.split("(")[0][:60]
# the last 40 % of the launches: steady cycles
rows = rows[int(len(rows) * 0.6):]
agg = collections.defaultdict(list)
gaps = collections.defaultdict(list)
prev = None
for r in rows:
    key = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1))
    agg[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    if prev is not None:
        gaps[key].append(int(r["Start_Timestamp"]) - int(prev["End_Timestamp"]))
    prev = r
tot = sum(sum(v) for v in agg.values())
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"{len(rows)} launches, kernel time {tot/1e3:.1f} us of a {span/1e3:.1f} us span ({100.0*tot/span:.1f} %)")
for key, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    g = gaps.get(key, [0])
    print(f"{key[0]:60s} wgs={key[1]:5d} n={len(v):4d} mean={sum(v)/len(v)/1e3:7.2f} us  share={100.0*sum(v)/tot:5.1f} %  gap before: {sum(g)/max(len(g),1)/1e3:5.2f} us")
