#!/bin/bash
# tools/build_plugin_variant.sh <tag> [hipcc flags for cnn_wino.hip ...]: a build of the evaluator plugin that differs from the product
# in compile-time switches of the convolution kernels -> tools/variants/<tag>/libsprl_amd_torch.so (git-ignored; travels to the GPU
# box).  On the box an A/B copies a variant over sprl_amd/libsprl_amd_torch.so of the scratch snapshot and runs bench.py --allow-lab.
set -e
TAG=$1; shift
cd "$(dirname "$0")/.."
OUT=tools/variants/$TAG
mkdir -p $OUT
TORCH=$(python3 -c 'import torch, os; print(os.path.dirname(torch.__file__))')
C=sprl_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize "$@" -c -o $OUT/cnn_wino.o $C/cnn_wino.hip
g++ -std=c++17 -O2 -fPIC -shared -w -D_GLIBCXX_USE_CXX11_ABI=1 -DUSE_ROCM -D__HIP_PLATFORM_AMD__ \
    -I$TORCH/include -I$TORCH/include/torch/csrc/api/include -I/opt/rocm/include \
    -o $OUT/libsprl_amd_torch.so $C/torch_eval.cpp $C/cnn_epilogue.o $OUT/cnn_wino.o $C/cnn_train.o \
    -L$TORCH/lib -Wl,--disable-new-dtags -Wl,-rpath,$TORCH/lib -Wl,--no-as-needed -ltorch -ltorch_cpu -ltorch_hip -lc10 -lc10_hip -lamdhip64
rm -f $OUT/cnn_wino.o
ls -la $OUT/libsprl_amd_torch.so
