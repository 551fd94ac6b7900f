set -u
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
bash tools/profile_r06.sh r06 stats layers timeline pmc_iter 2>&1 | tail -30
BENCH_EXTRA="--precision bf16" PRECISION=bf16 bash tools/profile_r06.sh r06_bf16 stats layers pmc_iter 2>&1 | tail -20
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c3
rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o run -- python3 $repo/bench.py --workload c3 > /tmp/prof_c3.log 2>&1
db=$(find /tmp/prof_c3 -name "*.db" | head -1)
python3 $repo/tools/rocpd_stats.py "$db" > $repo/gpurun_out/r06_c3_kernel_stats.txt 2>&1
head -5 $repo/gpurun_out/r06_c3_kernel_stats.txt | cut -c1-150
grep -m1 '"metric"' /tmp/prof_c3.log | cut -c1-300
