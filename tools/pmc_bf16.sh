# HBM counters (FETCH_SIZE, WRITE_SIZE: separate passes) of an iteration with --precision bf16, per launch grid for the bf16 kernels and summed over
# the iteration.  Run on the GPU box: bash tools/pmc_bf16.sh   ->  gpurun_out/r02_bf16v2_pmc_*.txt, r02_bf16v2_hbm_iteration_*.txt
set -u
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export DPI_OVERLAP_WGRAD=0
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c -d /tmp/pmc_$c -o run -- python3 $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --precision bf16 > /tmp/pmc_$c.log 2>&1
  db=$(find /tmp/pmc_$c -name "*.db" | head -1)
  python3 $repo/tools/rocpd_pmc.py "$db" --match conv_bf16 --by-grid > $repo/gpurun_out/r02_bf16v2_pmc_$c.txt 2>&1 || tail -5 /tmp/pmc_$c.log
  python3 $repo/tools/rocpd_pmc.py "$db" --totals --iterations 4 > $repo/gpurun_out/r02_bf16v2_hbm_iteration_$c.txt 2>&1
  head -40 $repo/gpurun_out/r02_bf16v2_pmc_$c.txt | cut -c1-150
done
