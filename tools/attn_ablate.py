#!/usr/bin/env python3
"""Times tc_attn_fwd (bf16, 256 sequences x 512 tokens, 12 heads) from variant builds of
attention.hip and compares each result with the first variant's.  Build them first, e.g.

    cd transcodon_amd/csrc && mkdir -p ../../tools/abl && F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 \
        -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans -shared" && \
        hipcc $F -o ../../tools/abl/attn_0base.so attention.hip runtime.hip && \
        hipcc $F -DTC_ATTN_MAX3 -o ../../tools/abl/attn_1max3.so attention.hip runtime.hip
"""
import ctypes as C
import glob
import os

import torch

here = os.path.dirname(os.path.abspath(__file__))
vp = C.c_void_p
dev = "cuda"
H, D = 12, 64
E = H * D
ref = {}
for rnd in range(2):
    for path in sorted(glob.glob(os.path.join(here, "abl", "attn_*.so"))):
        lib = C.CDLL(path)
        lib.tc_attn_fwd.restype = C.c_int
        lib.tc_attn_fwd.argtypes = [C.c_int, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
        row = []
        for (B, T) in [(256, 512), (64, 1024)]:
            torch.manual_seed(5)
            qkv = torch.randn(B * T, 3 * E, device=dev).bfloat16()
            out = torch.empty(B * T, E, device=dev, dtype=torch.bfloat16)
            s = torch.cuda.current_stream().cuda_stream

            def run():
                assert lib.tc_attn_fwd(1, qkv.data_ptr(), 3 * E, None, out.data_ptr(), B, T, H, D, s) == 0
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            key = (B, T)
            if key not in ref:
                ref[key] = out.float().clone()
            err = (out.float() - ref[key]).abs().max().item()
            row.append(f"B={B} T={T}: {us:7.1f} us {4.0 * B * H * T * T * D / us / 1e6:7.1f} TF/s  dmax {err:.2e}")
        print(f"{os.path.basename(path):24s} " + "   ".join(row), flush=True)
