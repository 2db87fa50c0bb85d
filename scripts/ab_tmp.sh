timeout 600 python -m pytest tests/test_conv_gemm_gpu.py -x -q -m gpu 2>&1 | tail -1
python scripts/bench_conv_layers.py 2>/dev/null | cut -c1-100
python scripts/device_phase_times.py 2>/dev/null | tail -16
