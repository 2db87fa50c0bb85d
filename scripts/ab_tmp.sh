timeout 900 python -m pytest tests/test_conv_gemm_gpu.py -x -q -m gpu 2>&1 | tail -2
python scripts/bench_conv_layers.py 2>/dev/null
python scripts/device_phase_times.py 2>/dev/null | tail -16
for c in vgg16 resnet50; do python bench.py --config $c --steps 20 --warmup 6 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$c', d['value'], d['ms_per_step'], d['roofline']['iteration']['frac'])"; done
