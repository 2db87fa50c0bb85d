timeout 900 python -m pytest tests/test_conv_gemm_gpu.py -x -q -m gpu -k linear 2>&1 | tail -2
for v in 0 1; do echo "== SCDA_GEMM_NO_MPART=$v"; if [ $v = 1 ]; then export SCDA_GEMM_NO_MPART=1; else unset SCDA_GEMM_NO_MPART; fi
python scripts/bench_conv_layers.py 2>/dev/null | grep -E "fc6|fc7"
python scripts/device_phase_times.py 2>/dev/null | grep -E "det_backward|step_begin"; done
