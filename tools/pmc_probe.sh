#!/bin/bash
# usage: tools/pmc_probe.sh <out-name> <match> <script.py> COUNTER [COUNTER...]  -> gpurun_out/<out-name>.txt
set -u
name=$1; match=$2; script=$3; shift 3
repo=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$name
rocprofv3 --pmc "$@" -d /tmp/pmc_$name -o run -- python3 $repo/$script > /tmp/pmc_$name.log 2>&1
db=$(find /tmp/pmc_$name -name "*.db" | head -1)
python3 $repo/tools/rocpd_pmc.py "$db" --match "$match" > $repo/gpurun_out/$name.txt 2>&1 || tail -5 /tmp/pmc_$name.log
cat $repo/gpurun_out/$name.txt
