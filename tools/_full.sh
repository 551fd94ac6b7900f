python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "q4" 2>&1 | tail -3
for q in 0 1; do echo "== q4 $q"; python tools/bench_conv.py --q4 $q --cases dec0_67_4 out_25_1 enc0_64_4 --which bwd_data --reps 20 2>/dev/null | grep -v "^case"; done
