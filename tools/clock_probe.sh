#!/bin/bash
# samples GPU clocks / power while the dominant conv runs in a loop
repo=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $repo/gpurun_out
python3 $repo/tools/loop_conv.py 10 > /tmp/loop.log 2>&1 &
pid=$!
sleep 5
for i in 1 2 3; do rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|fclk|Power" ; sleep 1; done
wait $pid
cat /tmp/loop.log | tail -2
rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Power"
