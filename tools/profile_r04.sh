#!/bin/bash
# usage: [BENCH_EXTRA="--precision bf16" PRECISION=bf16] tools/profile_r04.sh <out-prefix> [steps: stats layers overlap pmc_iter]   (run on the GPU box)
#   PRECISION labels the traffic JSON (bench.py quotes it only for the same --precision and the same kernel sources: csrc_sha256)
#   -> gpurun_out/<prefix>_kernel_stats.txt, _by_grid.txt, _layers.txt, _overlap_on.txt, _hbm_iteration_{FETCH,WRITE}_SIZE.txt,
#      _family_{FETCH,WRITE}_SIZE.txt, <prefix>_traffic.json
# Every rocprofv3 invocation has `python3 <script>` directly after `--`; counter passes are separate runs whose only trace domain
# is the kernel trace the counters attach to.  The profiled command is bench.py's default workload (eager loop) with 3 timed steps; bench.py
# itself runs 3 untimed iterations before them (first touch, the per-family timing pass, one warm-up): 6 iterations per file.
set -u
pre=$1; shift
steps=${@:-stats layers}
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
ITERS=6
# --mode eager: a fixed number of iterations per file whatever the precision (the bf16 line replays a graph by default and then adds an
# eager tail for the per-launch events: 10 iterations); the kernels are the same in both loop modes
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
    overlap)
      unset DPI_OVERLAP_WGRAD; unset DPI_PROFILE_TAGS
      db=$(trace ${pre}_overlap | tail -1)
      python3 $repo/tools/rocpd_stats.py "$db" > $out/${pre}_kernel_stats_overlap_on.txt 2>&1
      head -12 $out/${pre}_kernel_stats_overlap_on.txt | cut -c1-170 ;;
    pmc_iter)        # whole-iteration HBM bytes and the per-kernel breakdown: FETCH_SIZE and WRITE_SIZE in separate passes
      export DPI_OVERLAP_WGRAD=0; unset DPI_PROFILE_TAGS
      for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/pmc_${pre}_$c
        rocprofv3 --pmc $c -d /tmp/pmc_${pre}_$c -o run -- python3 $BENCH > /tmp/pmc_${pre}_$c.log 2>&1
        db=$(find /tmp/pmc_${pre}_$c -name "*.db" | head -1)
        python3 $repo/tools/rocpd_pmc.py "$db" --totals --iterations $ITERS > $out/${pre}_hbm_iteration_$c.txt 2>&1 || tail -5 /tmp/pmc_${pre}_$c.log
        python3 $repo/tools/rocpd_pmc.py "$db" --match conv_bwd_weight --by-grid > $out/${pre}_family_$c.txt 2>&1
        head -24 $out/${pre}_hbm_iteration_$c.txt | cut -c1-170
      done
      python3 $repo/tools/make_traffic_json.py $out/${pre} $ITERS ${PRECISION:-fp32} > $out/${pre}_traffic.json && head -c 1200 $out/${pre}_traffic.json ;;
  esac
done
