mkdir -p gpurun_out/r3f
python tools/snr_spread_gpu.py --seeds $(seq 0 47) --out gpurun_out/r3f/snr_spread_gpu48.json 2>&1 | tail -3
