python -m pytest tests/test_gpu_snr_parity.py -x -q -m gpu -s -k "mid_size or full_length or plateau" 2>&1 | grep -v "^$" | tail -25
python -m pytest tests/test_gpu_nets.py -x -q -m gpu -s -k "skip2d" 2>&1 | tail -4
