python -m pytest tests/test_gpu_ops.py tests/test_gpu_bench_size.py -x -q -m gpu 2>&1 | tail -3
C="res0_25_16 dec0_67_4 enc0_64_4 enc0_8_13 enc0_4_8 out_25_1 res1_51_32 enc1_17_26 dec1_137_8 res2_105_64 enc3_71_106"
for w in 0 1; do echo "== wide $w"; DPI_BW_WIDE=$w python tools/bench_conv.py --cases $C --which bwd_weight --reps 20 2>/dev/null | grep -v "^case"; done
