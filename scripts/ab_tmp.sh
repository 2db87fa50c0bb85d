python scripts/bench_conv_layers.py 2>/dev/null | grep -E "fc6|fc7"
for i in 1 2; do python scripts/device_phase_times.py 2>/dev/null | grep -E "det_backward|step_begin"; done
