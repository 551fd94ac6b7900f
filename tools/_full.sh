python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k q4 2>&1 | tail -2
bash tools/profile_r03.sh r03nt stats 2>&1 | grep "total kernel\|conv_q4\|bn_bwd\|chain_\|noise" | cut -c1-130
for i in 1 2; do python bench.py --steps 20 --no-cpu-baseline --no-other-modes --no-c3-extra 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print(r['ms_per_step'])"; done
