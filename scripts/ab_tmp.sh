R=$(pwd)
timeout 600 python -m pytest tests/test_conv_gemm_gpu.py -x -q -m gpu 2>&1 | tail -1
SCDA_OPS_LIB=$R/scda_amd/libscda_ops_spb2.so timeout 600 python -m pytest tests/test_conv_gemm_gpu.py -x -q -m gpu 2>&1 | tail -1
for lib in "" $R/scda_amd/libscda_ops_spb2.so "" $R/scda_amd/libscda_ops_spb2.so; do
echo "== lib '$lib'"; SCDA_OPS_LIB=$lib python scripts/device_phase_times.py 2>/dev/null | grep -E "backbones|det_backward|crops|phase1|phase2|phase3|phase4|step_begin"; done
echo "== conv layers base"; python scripts/bench_conv_layers.py 2>/dev/null | grep -E "dec_|conv5"
echo "== conv layers SPB2"; SCDA_OPS_LIB=$R/scda_amd/libscda_ops_spb2.so python scripts/bench_conv_layers.py 2>/dev/null | grep -E "dec_|conv5"
for lib in "" $R/scda_amd/libscda_ops_spb2.so; do SCDA_OPS_LIB=$lib python bench.py --config resnet50 --steps 10 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('resnet50 lib=$lib', d['value'], d['ms_per_step'])"; done
