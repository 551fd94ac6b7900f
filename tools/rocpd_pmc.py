#!/usr/bin/env python3
"""Per-kernel PMC counter averages from a rocprofv3 rocpd database:  python tools/rocpd_pmc.py db --match conv_mfma"""
import argparse
import sqlite3
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--match", default="")
a = ap.parse_args()
cur = sqlite3.connect(a.db).cursor()
agg = defaultdict(lambda: defaultdict(list))
for name, disp, cname, val, dur in cur.execute("select name, dispatch_id, counter_name, counter_value, duration from pmc_events"):
    if a.match in name:
        key = name.split("(")[0][-60:]
        agg[key][cname].append((disp, val, dur))
for k, cs in agg.items():
    print(k)
    for cname, vals in sorted(cs.items()):
        per = defaultdict(float)
        for disp, val, dur in vals:
            per[disp] += val
        v = list(per.values())
        print("   %-28s n=%d avg=%.4g" % (cname, len(v), sum(v) / len(v)))
