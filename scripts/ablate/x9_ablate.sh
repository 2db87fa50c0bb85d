#!/bin/bash
# Timing builds of the bf16 x 9 GEMM with parts of its K loop compiled out (csrc/conv_gemm.hip X9_ABLATE; results wrong by construction,
# only the time is read): builds scripts/ablate/x9_libs/libscda_ops_<bits>.so HERE (hipcc cross-compiles), `run` times FC6's three
# products with each on the GPU box.   scripts/ablate/x9_ablate.sh build "1 2 4 ..." | run
set -e
cd "$(dirname "$0")/../.."
CS=scda_amd/csrc; OUT=scripts/ablate/x9_libs
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function -Wno-unused-value -fhip-fp32-correctly-rounded-divide-sqrt"
if [ "$1" = build ]; then
  mkdir -p $OUT
  for b in $2; do
    ( /opt/rocm/bin/hipcc $FLAGS -DX9_ABLATE=$b -c $CS/conv_gemm.hip -o $OUT/conv_gemm_$b.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libscda_ops_$b.so $CS/detection_ops.o $CS/box_ops.o $OUT/conv_gemm_$b.o $CS/conv_wino.o $CS/nn_ops.o $CS/image_ops.o && rm $OUT/conv_gemm_$b.o ) &
  done
  wait
  ls -la $OUT
else
  for f in $OUT/libscda_ops_*.so; do
    echo "== $f"
    SCDA_OPS_LIB=$PWD/$f python scripts/time_gemm_x9.py x9only 2>&1 | grep -v amdgpu.ids
  done
fi
