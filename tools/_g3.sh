mkdir -p gpurun_out/r3c
for mv in 1048576 100000000 4194305; do
DPI_OVERLAP_MAX_VOXELS=$mv python bench.py --steps 20 --no-cpu-baseline --no-other-modes --no-c3-extra > gpurun_out/r3c/bench_mv$mv.json 2>/dev/null
python - <<P
import json
r=json.load(open('gpurun_out/r3c/bench_mv$mv.json')); ro=r['roofline']
print($mv, r['ms_per_step'], ro['kernel'][:40], ro['frac'], ro['isolated']['frac'])
P
done
