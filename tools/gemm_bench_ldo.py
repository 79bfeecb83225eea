"""Large-tile tc_gemm against the leading dimensions of the result and of the activation
operand: which row strides are slow?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transcodon_amd import _lib  # noqa: E402

dev = "cuda"
M = 131072
lib = _lib.load()
strm = torch.cuda.current_stream().cuda_stream


def timeit(run):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


which = sys.argv[1:] or ["out_bf16", "out_f32", "lda"]
if "out_bf16" in which:
    for (N, K) in [(3072, 768), (2304, 768)]:
        a = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
        b = torch.randn(N, device=dev) * 0.1
        for pad in (0, 64, 128, 1024):
            buf = torch.empty(M, N + pad, device=dev, dtype=torch.bfloat16)
            for epi, name in ((_lib.EPI_BIAS, "bias"), (_lib.EPI_BIAS_GELU, "gelu")):
                def run():
                    assert lib.tc_gemm(1, epi, a.data_ptr(), K, w.data_ptr(), b.data_ptr(), None,
                                       buf.data_ptr(), N + pad, M, N, K, 1.0, None, None, 0, 0, strm) == 0
                us = timeit(run)
                print(f"bf16 out {name} N={N} K={K} ldo={N + pad}: {us:8.1f} us {2 * M * N * K / us / 1e6:8.1f} TF/s", flush=True)
if "out_f32" in which:
    for (N, K) in [(768, 768), (768, 3072)]:
        a = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
        b = torch.randn(N, device=dev) * 0.1
        for pad in (0, 32, 64, 256):
            buf = torch.randn(M, N + pad, device=dev)

            def run():
                assert lib.tc_gemm(1, _lib.EPI_BIAS_RESIDUAL, a.data_ptr(), K, w.data_ptr(), b.data_ptr(),
                                   buf.data_ptr(), buf.data_ptr(), N + pad, M, N, K, 1.0, None, None, 0, 0, strm) == 0
            us = timeit(run)
            print(f"f32 residual in place N={N} K={K} ldo={N + pad}: {us:8.1f} us {2 * M * N * K / us / 1e6:8.1f} TF/s", flush=True)
if "lda" in which:
    N, K = 768, 3072
    w = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
    b = torch.randn(N, device=dev) * 0.1
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for pad in (0, 64, 128, 1024):
        a = torch.randn(M, K + pad, device=dev).bfloat16()

        def run():
            assert lib.tc_gemm(1, _lib.EPI_BIAS, a.data_ptr(), K + pad, w.data_ptr(), b.data_ptr(), None,
                               out.data_ptr(), N, M, N, K, 1.0, None, None, 0, 0, strm) == 0
        us = timeit(run)
        print(f"bf16 N={N} K={K} lda={K + pad}: {us:8.1f} us {2 * M * N * K / us / 1e6:8.1f} TF/s", flush=True)
