#!/usr/bin/env python3
"""Fixed cost per workgroup of the bf16 attention forward: times tc_attn_fwd_q2 from builds of attention.hip whose
key loop stops after the first N tiles (-DTC_ABL_NKT=N; results are then wrong, only the time is read).

    cd transcodon_amd/csrc && mkdir -p ../../tools/abl && F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 \
        -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans -shared" && \
        for n in 0 1 2 4 8; do hipcc $F -DTC_ABL_NKT=$n -o ../../tools/abl/attn_nkt$n.so attention.hip runtime.hip; done
"""
import ctypes as C
import glob
import os
import re

import torch

here = os.path.dirname(os.path.abspath(__file__))
vp = C.c_void_p
H, D = 12, 64
E = H * D
for (B, T) in [(256, 512), (128, 1024)]:
    torch.manual_seed(5)
    qkv = torch.randn(B * T, 3 * E, device="cuda").bfloat16()
    out = torch.empty(B * T, E, device="cuda", dtype=torch.bfloat16)
    s = torch.cuda.current_stream().cuda_stream
    for path in sorted(glob.glob(os.path.join(here, "abl", "attn_*.so"))):
        lib = C.CDLL(path)
        lib.tc_attn_fwd_q2.restype = C.c_int
        lib.tc_attn_fwd_q2.argtypes = [C.c_int, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]

        def run():
            assert lib.tc_attn_fwd_q2(1, qkv.data_ptr(), 3 * E, None, out.data_ptr(), B, T, H, D, s) == 0
        best = 1e9
        for rep in range(3):
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
        print(f"B={B} T={T} {os.path.basename(path):16s} {best:7.1f} us", flush=True)
