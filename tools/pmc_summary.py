#!/usr/bin/env python3
"""Mean per-launch value of every counter rocprofv3 --pmc collected for the flux kernels."""
import collections, csv, glob, os, sys
agg = collections.defaultdict(list)
for d in sys.argv[1:]:
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_flux" in r["Kernel_Name"] or "k_indirect" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"{k:45s} {c:34s} launches={len(v):3d} mean={sum(v)/len(v):.5g}")
