#!/usr/bin/env python3
"""Per-kernel PMC counter averages from a rocprofv3 rocpd database.

    python tools/rocpd_pmc.py db --match conv_mfma [--by-grid]     per-kernel averages per dispatch (summed over XCDs / instances)
    python tools/rocpd_pmc.py db --totals --iterations N            counter totals over ALL dispatches, divided by N iterations,
                                                                    with the per-kernel breakdown (whole-iteration HBM bytes)
"""
import argparse
import re
import sqlite3
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--match", default="")
ap.add_argument("--by-grid", action="store_true")
ap.add_argument("--totals", action="store_true")
ap.add_argument("--iterations", type=int, default=1)
a = ap.parse_args()
cur = sqlite3.connect(a.db).cursor()


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"^void ", "", name).split("(")[0][:110]


grids = {}
durs = {}
if a.by_grid:                                                          # pmc_events carries no launch geometry: join on dispatch_id
    for disp, gx, wx, dur in cur.execute("select dispatch_id, grid_x, workgroup_x, duration from kernels"):
        grids[disp] = gx // max(wx, 1)
        durs[disp] = dur
agg = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))      # key -> counter -> dispatch -> value
for name, disp, cname, val in cur.execute("select name, dispatch_id, counter_name, counter_value from pmc_events"):
    if a.match in name:
        key = short(name)
        if a.by_grid:
            key += "  grid=%s" % grids.get(disp, "?")
        agg[key][cname][disp] += val
if a.totals:
    tot = defaultdict(float)
    per = defaultdict(lambda: defaultdict(float))
    for k, cs in agg.items():
        for cname, d in cs.items():
            s = sum(d.values())
            tot[cname] += s
            per[cname][k] += s
    for cname, t in tot.items():
        print("%s: total %.6g over all dispatches = %.6g per iteration (%d iterations)" % (cname, t, t / a.iterations, a.iterations))
        for k, s in sorted(per[cname].items(), key=lambda kv: -kv[1])[:40]:
            print("   %14.6g per iteration  %5.1f %%  %s" % (s / a.iterations, 100.0 * s / t, k))
else:
    for k, cs in agg.items():
        print(k)
        if durs:                                                        # duration of the SAME (counter-collecting) dispatches, ns
            dd = [durs[x] for x in next(iter(cs.values())) if x in durs]
            if dd:
                print("   %-28s n=%d avg=%.6g" % ("duration_ns", len(dd), sum(dd) / len(dd)))
        for cname, d in sorted(cs.items()):
            v = list(d.values())
            print("   %-28s n=%d avg=%.6g" % (cname, len(v), sum(v) / len(v)))
