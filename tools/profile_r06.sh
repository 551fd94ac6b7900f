#!/bin/bash
# usage: tools/profile_r06.sh <out-prefix> [steps: stats layers timeline pmc_iter]   (run on the GPU box; round 6: as tools/profile_r05.sh;
#   FAMILY_MATCH = name filter of the per-grid counter table of the backward-weight family — `bwd_weight` also matches the bf16 kernels)
#   like tools/profile_r04.sh; adds `timeline` (tools/timeline.py of the default concurrent schedule: who runs beside whom).
#   stats / layers / pmc_iter run with DPI_OVERLAP_WGRAD=0 (every kernel alone: a duration is the kernel's own).
set -u
pre=$1; shift
steps=${@:-stats layers timeline}
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
ITERS=6
BENCH="$repo/bench.py --mode eager --steps 3 --warmup 1 --no-cpu-baseline --no-other-modes --no-c3-extra ${BENCH_EXTRA:-}"
trace() {
  local name=$1
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o run -- python3 $BENCH > /tmp/prof_$name.log 2>&1
  grep -m1 '"metric"' /tmp/prof_$name.log | cut -c1-200
  find /tmp/prof_$name -name "*.db" | head -1
}
for s in $steps; do
  case $s in
    stats)
      export DPI_OVERLAP_WGRAD=0; unset DPI_PROFILE_TAGS
      db=$(trace ${pre}_stats | tail -1)
      python3 $repo/tools/rocpd_stats.py "$db" > $out/${pre}_kernel_stats.txt 2>&1
      python3 $repo/tools/rocpd_stats.py "$db" --by-grid --top 400 > $out/${pre}_kernel_stats_by_grid.txt 2>&1
      head -30 $out/${pre}_kernel_stats.txt | cut -c1-170 ;;
    layers)
      export DPI_OVERLAP_WGRAD=0; export DPI_PROFILE_TAGS=/tmp/${pre}_tags.json
      db=$(trace ${pre}_layers | tail -1)
      python3 $repo/tools/rocpd_stats.py "$db" --tags /tmp/${pre}_tags.json --top 400 > $out/${pre}_kernel_stats_layers.txt 2>&1
      cp /tmp/${pre}_tags.json $out/ 2>/dev/null
      unset DPI_PROFILE_TAGS
      head -40 $out/${pre}_kernel_stats_layers.txt | cut -c1-200 ;;
    timeline)
      unset DPI_OVERLAP_WGRAD; unset DPI_PROFILE_TAGS
      db=$(trace ${pre}_timeline | tail -1)
      python3 $repo/tools/rocpd_stats.py "$db" > $out/${pre}_kernel_stats_overlap_on.txt 2>&1
      python3 $repo/tools/timeline.py "$db" --list > $out/${pre}_timeline.txt 2>&1
      head -14 $out/${pre}_timeline.txt ;;
    pmc_iter)
      export DPI_OVERLAP_WGRAD=0; unset DPI_PROFILE_TAGS
      for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/pmc_${pre}_$c
        rocprofv3 --pmc $c -d /tmp/pmc_${pre}_$c -o run -- python3 $BENCH > /tmp/pmc_${pre}_$c.log 2>&1
        db=$(find /tmp/pmc_${pre}_$c -name "*.db" | head -1)
        python3 $repo/tools/rocpd_pmc.py "$db" --totals --iterations $ITERS > $out/${pre}_hbm_iteration_$c.txt 2>&1 || tail -5 /tmp/pmc_${pre}_$c.log
        python3 $repo/tools/rocpd_pmc.py "$db" --match ${FAMILY_MATCH:-bwd_weight} --by-grid > $out/${pre}_family_$c.txt 2>&1
        head -24 $out/${pre}_hbm_iteration_$c.txt | cut -c1-170
      done
      python3 $repo/tools/make_traffic_json.py $out/${pre} $ITERS ${PRECISION:-fp32} > $out/${pre}_traffic.json && head -c 1200 $out/${pre}_traffic.json ;;
  esac
done
