#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd SQLite) kernel trace: per-kernel calls / total / avg / share, like `--stats`.

    python tools/rocpd_stats.py x_results.db [--by-grid] [--tags tags.json] > profiles/xxx.txt

--tags: the id -> layer table a run with DPI_PROFILE_TAGS=<path> wrote (deep_prior_interpolation_amd/ops.py).  Every conv
launch of that run is bracketed by `dpi_marker_kernel` dispatches whose workgroup count is the id (1 = end of scope); dispatches
are walked in host launch order (dispatch_id) and each kernel between a marker and its end marker gets the layer appended to its
row key — so the 25->16 forward conv at 256x128x128 has its own row even though three layers share its template and launch grid.
"""
import argparse
import json
import re
import sqlite3
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:150]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--by-grid", action="store_true", help="split kernels by launch grid")
    ap.add_argument("--tags", default=None, help="id -> layer JSON written by a DPI_PROFILE_TAGS run")
    ap.add_argument("--top", type=int, default=60)
    a = ap.parse_args()
    cur = sqlite3.connect(a.db).cursor()
    rows = cur.execute("select name, duration, grid_x, grid_y, grid_z, workgroup_x, vgpr_count, lds_size, sgpr_count, dispatch_id "
                       "from kernels order by dispatch_id").fetchall()
    tags = None
    if a.tags:
        with open(a.tags) as fp:
            tags = {int(k): v for k, v in json.load(fp).items()}
    agg = defaultdict(lambda: [0, 0, None])
    total = 0
    scope = None
    nmark = 0
    for name, dur, gx, gy, gz, wx, vg, lds, sg, _disp in rows:
        if "dpi_marker_kernel" in name:
            nmark += 1
            i = gx // max(wx, 1)
            scope = None if (tags is None or i == 1) else tags.get(i, "tag %d" % i)
            continue                                       # markers are profiling aids: not part of the totals
        key = short(name) + ("  grid=(%d,%d,%d)" % (gx // max(wx, 1), gy, gz) if a.by_grid else "")
        if scope is not None:
            key += "  [" + scope + "]"
        e = agg[key]
        e[0] += 1
        e[1] += dur
        e[2] = (vg, lds, sg)
        total += dur
    print("total kernel time %.3f ms over %d dispatches%s" % (total / 1e6, len(rows) - nmark,
                                                              " (+%d marker dispatches, excluded)" % nmark if nmark else ""))
    print("%8s %12s %10s %6s  %5s %6s %5s  %s" % ("calls", "total_ms", "avg_us", "%", "vgpr", "lds", "sgpr", "kernel"))
    for key, (n, t, res) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
        print("%8d %12.3f %10.1f %6.2f  %5s %6s %5s  %s" % (n, t / 1e6, t / n / 1e3, 100.0 * t / total, res[0], res[1], res[2], key))


if __name__ == "__main__":
    main()
