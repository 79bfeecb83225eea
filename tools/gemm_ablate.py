#!/usr/bin/env python3
"""Times tc_gemm (bf16, bias epilogue, M = 131072) from ablation builds of gemm.hip
(results are wrong by construction, only the time matters).  Build them first, e.g.

    cd transcodon_amd/csrc && mkdir -p ../../tools/abl && for v in "" -DTC_ABL_NOREADS -DTC_ABL_NODMA \
        "-DTC_ABL_NOREADS -DTC_ABL_NODMA" "-DTC_ABL_NOREADS -DTC_ABL_NODMA -DTC_ABL_NOBAR"; do \
        hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form $v -shared \
        -o ../../tools/abl/abl_$(echo "full$v" | tr -d ' -').so gemm.hip runtime.hip; done
"""
import ctypes as C
import glob
import os
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
vp = C.c_void_p
dev = "cuda"
M = 131072
for path in sorted(glob.glob(os.path.join(here, "abl", "abl_*.so"))):
    lib = C.CDLL(path)
    lib.tc_gemm.restype = C.c_int
    lib.tc_gemm.argtypes = [C.c_int, C.c_int, vp, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                            C.c_float, vp, vp, C.c_int, C.c_int, vp]
    row = []
    for (N, K) in [(3072, 3072), (3072, 768)]:
        a = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
        b = torch.randn(N, device=dev) * 0.1
        out = torch.empty(M, N + 64, device=dev, dtype=torch.bfloat16)
        s = torch.cuda.current_stream().cuda_stream

        def run():
            assert lib.tc_gemm(1, 0, a.data_ptr(), K, w.data_ptr(), b.data_ptr(), None, out.data_ptr(), N + 64, M, N, K,
                               1.0, None, None, 0, 0, s) == 0
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        row.append(f"N={N} K={K}: {us:8.1f} us {2 * M * N * K / us / 1e6:7.1f} TF/s")
    print(f"{os.path.basename(path):28s} " + "   ".join(row), flush=True)
