#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc csv passes: python tools/pmc_summary.py <kernel substring> <dir>..."""
import csv
import glob
import sys
from collections import defaultdict

sub = sys.argv[1]
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        dur = defaultdict(lambda: [0, 0])
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if sub not in k:
                continue
            key = (k.split("(")[0][-70:], r["Grid_Size"])
            e = acc[key][r["Counter_Name"]]
            e[0] += 1
            e[1] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur[key][0] += 1
                dur[key][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        for key in sorted(acc):
            n, t = dur[key]
            print("%s grid %s: %d launches, avg %.1f us" % (key[0], key[1], n, t / n / 1e3))
            for c, (m, v) in sorted(acc[key].items()):
                print("    %-28s %16.1f per launch" % (c, v / m))
