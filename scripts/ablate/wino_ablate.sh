#!/bin/bash
# timing ablations of conv_wino_kernel's K loop (results are WRONG by construction; only the time is read):
#   bash scripts/ablate/wino_ablate.sh            builds scripts/ablate/build/libscda_ops_A<n>.so (git-ignored, travels with gpurun) for n in 4 8 16 28 (here, no GPU needed)
#   bash scripts/ablate/wino_ablate.sh run        times conv2_2 / conv3_2 / conv4_2 forward with each (on the GPU box)
set -u
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/../..; pwd)}
O=$R/scripts/ablate/build; mkdir -p $O
if [ "${1:-}" = "run" ]; then
  for n in 0 4 8 16 28; do
    lib=$O/libscda_ops_A$n.so; [ $n = 0 ] && lib=$R/scda_amd/libscda_ops.so
    echo "== ablate $n"; SCDA_OPS_LIB=$lib python $R/scripts/bench_wino.py conv2_2 conv3_2 conv4_2 2>&1 | grep -v amdgpu | cut -c1-8,48-110
  done
  for n in ${WG_SET:-0 32 64 128 224 256 512 992}; do      # the weight gradient's knobs (256 no epilogue, 512 no operand transforms, 992 all)
    lib=$O/libscda_ops_A$n.so; [ $n = 0 ] && lib=$R/scda_amd/libscda_ops.so
    echo "== wgrad ablate $n"; SCDA_OPS_LIB=$lib python $R/scripts/bench_wino.py conv2_2 conv3_2 conv4_2 2>&1 | grep -v amdgpu | sed "s/ GFLOP.*| wgrad/ wgrad/"
  done
  exit 0
fi
cd $R/scda_amd/csrc
for n in ${BUILD_SET:-4 8 16 28 32 64 128 224 256 512 992}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function -DSCDA_WINO_ABLATE=$n -c conv_wino.hip -o $O/conv_wino_A$n.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libscda_ops_A$n.so detection_ops.o box_ops.o conv_gemm.o $O/conv_wino_A$n.o nn_ops.o image_ops.o && echo built $n
done
