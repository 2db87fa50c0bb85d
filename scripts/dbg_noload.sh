for i in 1 2; do
echo OLD; SCDA_LIB_PATH=$PWD/gpurun_tmp_old.so python scripts/bench_conv_layers.py 2>&1 | grep -v amdgpu | head -12
echo NEW; python scripts/bench_conv_layers.py 2>&1 | grep -v amdgpu | head -12
done
