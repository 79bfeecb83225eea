#!/bin/bash
# Builds the library and its two measurement variants: tools/libtc_stamps.so (cycle stamps in the large-tile
# GEMM, -DTC_GEMM_STAMPS) and, with "base" as first argument, tools/ab/libtc_base.so (a copy of the CURRENT
# build to compare later builds against with tools/gemm_ab.py).  Extra arguments go to the stamps compile.
set -e
cd "$(dirname "$0")/../transcodon_amd/csrc"
make -j8 2>&1 | grep -v "^/opt" || true
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form"
OTHERS="runtime.o rowops.o train_ops.o gemm_tn.o attention.o attention_bwd.o encoder.o encoder_train.o"
if [ "$1" = "base" ]; then
    shift
    mkdir -p ../../tools/ab
    cp libtranscodon_hip.so ../../tools/ab/libtc_base.so
fi
/opt/rocm/bin/hipcc $FLAGS -DTC_GEMM_STAMPS "$@" -c gemm.hip -o /tmp/gemm_stamps.o
/opt/rocm/bin/hipcc ${FLAGS/-mllvm -amdgpu-mfma-vgpr-form/} -DTC_GEMM_STAMPS "$@" -c gemm_w.hip -o /tmp/gemm_w_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libtc_stamps.so $OTHERS /tmp/gemm_stamps.o /tmp/gemm_w_stamps.o
