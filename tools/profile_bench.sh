#!/bin/bash
# usage: tools/profile_bench.sh <out-name> [bench args...]   -> gpurun_out/<out-name>.txt (per-kernel stats of one bench run)
set -u
name=$1; shift
repo=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$name
rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o run -- python3 $repo/bench.py "$@" > /tmp/prof_$name.log 2>&1
grep -m1 '"metric"' /tmp/prof_$name.log | cut -c1-200
db=$(find /tmp/prof_$name -name "*.db" | head -1)
python3 $repo/tools/rocpd_stats.py "$db" > $repo/gpurun_out/$name.txt 2>&1 || tail -5 /tmp/prof_$name.log
python3 $repo/tools/rocpd_stats.py "$db" --by-grid --top 400 > $repo/gpurun_out/${name}_bygrid.txt 2>&1
head -45 $repo/gpurun_out/$name.txt | cut -c1-150
