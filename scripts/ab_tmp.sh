timeout 1200 python -m pytest tests/test_maskrcnn_gpu.py -x -q -m gpu 2>&1 | tail -2
for c in maskrcnn resnet50; do python bench.py --config $c --steps 10 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$c', d['value'], d['ms_per_step'], d['config'].get('mask_rois'), d['roofline']['iteration'])"; done
