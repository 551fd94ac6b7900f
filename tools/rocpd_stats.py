#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd SQLite) kernel trace: per-kernel calls / total / avg / share, like `--stats`.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--by-grid] [--skip-first N] > profiles/xxx.txt
"""
import argparse
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
    ap.add_argument("--top", type=int, default=60)
    a = ap.parse_args()
    cur = sqlite3.connect(a.db).cursor()
    rows = cur.execute("select name, duration, grid_x, grid_y, grid_z, workgroup_x, vgpr_count, lds_size, sgpr_count from kernels").fetchall()
    agg = defaultdict(lambda: [0, 0, None])
    total = 0
    for name, dur, gx, gy, gz, wx, vg, lds, sg in rows:
        key = short(name) + ("  grid=(%d,%d,%d)" % (gx // max(wx, 1), gy, gz) if a.by_grid else "")
        e = agg[key]
        e[0] += 1
        e[1] += dur
        e[2] = (vg, lds, sg)
        total += dur
    print("total kernel time %.3f ms over %d dispatches" % (total / 1e6, len(rows)))
    print("%8s %12s %10s %6s  %5s %6s %5s  %s" % ("calls", "total_ms", "avg_us", "%", "vgpr", "lds", "sgpr", "kernel"))
    for key, (n, t, res) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
        print("%8d %12.3f %10.1f %6.2f  %5s %6s %5s  %s" % (n, t / 1e6, t / n / 1e3, 100.0 * t / total, res[0], res[1], res[2], key))


if __name__ == "__main__":
    main()
