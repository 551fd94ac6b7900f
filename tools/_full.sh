python -m pytest tests/test_gpu_ops.py tests/test_gpu_bench_size.py -x -q -m gpu -k "conv" 2>&1 | tail -2
python tools/bench_conv.py --cases sc0_67_25 sc0_64_25 res0_25_16_k1 sc1_137_51 --reps 20 2>/dev/null
