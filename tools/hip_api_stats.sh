#!/bin/bash
# HIP API call histogram of one bench run (rocprofv3 --hip-trace); usage: tools/hip_api_stats.sh [bench args]
repo=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/hipapi
rocprofv3 --hip-trace --stats -d /tmp/hipapi -o run -- python3 $repo/bench.py "$@" > /tmp/hipapi.log 2>&1
db=$(find /tmp/hipapi -name "*.db" | head -1)
python3 - "$db" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if 'region' in t.lower() or 'api' in t.lower() or 'memory' in t.lower()][:20])
for t in ("regions", "rocpd_region", "region"):
    if t in tabs or True:
        try:
            rows = cur.execute("select name, count(*), sum(end-start)/1e6 from regions group by name order by 2 desc limit 25").fetchall()
            for r in rows: print(r)
            break
        except Exception as e:
            print("query failed", e); break
try:
    for r in cur.execute("select name, count(*), sum(size) from memory_copies group by name order by 2 desc limit 10"): print("memcpy", r)
except Exception as e:
    print("memcpy query failed", e)
PY
