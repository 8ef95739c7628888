#!/usr/bin/env python3
"""Print a rocprofv3 --stats kernel summary (directory given) as a compact table."""
import csv, glob, os, sys
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            print(f'{r["Name"][:64]:64s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:8.2f} pct={r["Percentage"]}')
