#!/usr/bin/env python3
"""A/B timing of two builds of libtranscodon_hip.so on the encoder's GEMM shapes, in ONE process with
interleaved rounds (timings of separate processes or devices are not comparable).

    python tools/gemm_ab.py [base.so] [new.so] [--only qkv,fc1,...] [--rounds 5]

Defaults: tools/ab/libtc_base.so against transcodon_amd/csrc/libtranscodon_hip.so.  Shapes use the
forward's leading dimensions (FFN intermediate rows 3136 elements apart).  Also checks that both
builds produce the same result.
"""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transcodon_amd import _lib  # noqa: E402


def load(path):
    lib = C.CDLL(path)
    res, args = _lib._SIGNATURES["tc_gemm"]
    lib.tc_gemm.restype, lib.tc_gemm.argtypes = res, args
    lib.tc_last_error.restype = C.c_char_p
    res, args = _lib._SIGNATURES["tc_gemm_train"]
    lib.tc_gemm_train.restype, lib.tc_gemm_train.argtypes = res, args
    if hasattr(lib, "tc_gemm_residual_split"):
        res, args = _lib._SIGNATURES["tc_gemm_residual_split"]
        lib.tc_gemm_residual_split.restype, lib.tc_gemm_residual_split.argtypes = res, args
    if hasattr(lib, "tc_gemm_ln"):
        res, args = _lib._SIGNATURES["tc_gemm_ln"]
        lib.tc_gemm_ln.restype, lib.tc_gemm_ln.argtypes = res, args
    if hasattr(lib, "tc_gemm_residual_ln"):
        res, args = _lib._SIGNATURES["tc_gemm_residual_ln"]
        lib.tc_gemm_residual_ln.restype, lib.tc_gemm_residual_ln.argtypes = res, args
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("base", nargs="?", default=os.path.join(ROOT, "tools", "ab", "libtc_base.so"))
    ap.add_argument("new", nargs="?", default=os.path.join(ROOT, "transcodon_amd", "csrc", "libtranscodon_hip.so"))
    ap.add_argument("--only", default="qkv,qkv_bias,out,fc1,fc1_bias,fc2")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--M", type=int, default=131072)
    ap.add_argument("--fold-both", action="store_true", help="the *_fold shapes run the folded product in both builds")
    a = ap.parse_args()
    libs = {"base": load(a.base), "new": load(a.new)}
    dev = "cuda"
    M, E, F, T = a.M, 768, 3072, 512
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(M, E, device=dev, generator=g).bfloat16()
    mid = torch.randn(M, F + 64, device=dev, generator=g).bfloat16()
    res = torch.randn(M, E, device=dev, generator=g)
    ang = torch.arange(T, dtype=torch.float32)[:, None] * (1.0 / 10000 ** (torch.arange(0, 64, 2).float() / 64))[None]
    cos, sin = ang.cos().to(dev).contiguous(), ang.sin().to(dev).contiguous()
    strm = torch.cuda.current_stream().cuda_stream
    # name: (A, lda, N, K, epilogue, ldo, f32 out)
    shapes = {"qkv": (x, E, 3 * E, E, _lib.EPI_QKV_ROPE, 3 * E, False),
              "qkv_bias": (x, E, 3 * E, E, _lib.EPI_BIAS, 3 * E, False),
              "out": (x, E, E, E, _lib.EPI_BIAS_RESIDUAL, E, True),
              "fc1": (x, E, F, E, _lib.EPI_BIAS_GELU, F + 64, False),
              "fc1_bias": (x, E, F, E, _lib.EPI_BIAS, F + 64, False),
              "out_bias": (x, E, E, E, _lib.EPI_BIAS, E, False),
              "fc2_bias": (mid, F + 64, E, F, _lib.EPI_BIAS, E, False),
              "sq_bias": (mid, F + 64, F, F, _lib.EPI_BIAS, F + 64, False),
              "fc2": (mid, F + 64, E, F, _lib.EPI_BIAS_RESIDUAL, E, True),
              # the new build also writes the bf16 copy and the row sums of a folded LayerNorm
              "out_ln": (x, E, E, E, _lib.EPI_BIAS_RESIDUAL, E, True),
              "fc2_ln": (mid, F + 64, E, F, _lib.EPI_BIAS_RESIDUAL, E, True),
              # the new build keeps the residual stream as two bf16 planes, updated in place
              "out_split": (x, E, E, E, _lib.EPI_BIAS_RESIDUAL, E, True),
              "fc2_split": (mid, F + 64, E, F, _lib.EPI_BIAS_RESIDUAL, E, True)}
    # training products (tc_gemm_train): GELU with the saved pre-activation, and the dGELU data gradient
    shapes["fc1_save"] = (x, E, F, E, _lib.EPI_BIAS_GELU_SAVE, F + 64, False)
    shapes["fc2_dgelu"] = (x, E, F, E, _lib.EPI_DGELU, F + 64, False)
    # the new build runs these with a LayerNorm folded in (row pairs + column sums in the epilogue)
    shapes["qkv_fold"] = shapes["qkv"]
    shapes["fc1_fold"] = shapes["fc1"]
    ln_rows = torch.rand(M, 2, device=dev, generator=g) + 0.5
    plane_hi = torch.randn(M, E, device=dev, generator=g).bfloat16()
    plane_lo = (torch.randn(M, E, device=dev, generator=g) * 1e-3).bfloat16()
    copy16 = torch.empty(M, E, device=dev, dtype=torch.bfloat16)
    pre = torch.randn(M, F + 64, device=dev, generator=g).bfloat16()      # saved pre-activation (written or read)
    row_acc = torch.zeros(M, 2, device=dev, dtype=torch.int64)
    for name in a.only.split(","):
        A, lda, N, K, epi, ldo, f32 = shapes[name]
        w = (torch.randn(N, K, device=dev, generator=g) * 0.03).bfloat16()
        b = torch.randn(N, device=dev, generator=g) * 0.1
        outs = {k: torch.zeros(M, ldo, device=dev, dtype=torch.float32 if f32 else torch.bfloat16) for k in libs}

        def run(k):
            if name.endswith("_ln") and k == "new":
                rc = libs[k].tc_gemm_residual_ln(1, A.data_ptr(), lda, w.data_ptr(), b.data_ptr(), res.data_ptr(),
                                                 outs[k].data_ptr(), copy16.data_ptr(), row_acc.data_ptr(), ldo, M, N, K, strm)
                assert rc == 0, libs[k].tc_last_error()
                return
            if name.endswith("_fold") and (k == "new" or a.fold_both):
                rc = libs[k].tc_gemm_ln(1, epi, A.data_ptr(), lda, w.data_ptr(), b.data_ptr(), b.data_ptr(), ln_rows.data_ptr(),
                                        outs[k].data_ptr(), ldo, M, N, K, 0.125, cos.data_ptr(), sin.data_ptr(), T, 64, strm)
                assert rc == 0, libs[k].tc_last_error()
                return
            if name.endswith("_split") and (k == "new" or hasattr(libs[k], "tc_gemm_residual_split")):
                rc = libs[k].tc_gemm_residual_split(1, A.data_ptr(), lda, w.data_ptr(), b.data_ptr(), planes[k][0].data_ptr(),
                                                    planes[k][1].data_ptr(), row_accs[k].data_ptr(), ldo, M, N, K, strm)
                assert rc == 0, libs[k].tc_last_error()
                return
            if epi in (_lib.EPI_BIAS_GELU_SAVE, _lib.EPI_DGELU):
                save = epi == _lib.EPI_BIAS_GELU_SAVE
                rc = libs[k].tc_gemm_train(1, epi, A.data_ptr(), lda, w.data_ptr(), b.data_ptr() if save else None,
                                           None if save else pre.data_ptr(), outs[k].data_ptr(), pre.data_ptr() if save else None,
                                           ldo, M, N, K, 1, strm)
                assert rc == 0, libs[k].tc_last_error()
                return
            rc = libs[k].tc_gemm(1, epi, A.data_ptr(), lda, w.data_ptr(), b.data_ptr(), res.data_ptr() if f32 else None,
                                 outs[k].data_ptr(), ldo, M, N, K, 0.125, cos.data_ptr(), sin.data_ptr(), T, 64, strm)
            assert rc == 0, libs[k].tc_last_error()

        # the plane form updates its residual in place: every build gets its own copy, compared after the first call
        planes = {k: (plane_hi.clone(), plane_lo.clone()) for k in libs}
        row_accs = {k: torch.zeros(M, 2, device=dev, dtype=torch.int64) for k in libs}
        for k in libs:
            run(k)
        torch.cuda.synchronize()
        if name.endswith("_split") and hasattr(libs["base"], "tc_gemm_residual_split"):
            same = all(torch.equal(planes["base"][i], planes["new"][i]) for i in range(2)) and torch.equal(row_accs["base"], row_accs["new"])
            maxd = float((planes["base"][0].float() - planes["new"][0].float()).abs().max())
        else:
            same = torch.equal(outs["base"][:, :N], outs["new"][:, :N])
            maxd = float((outs["base"][:, :N].float() - outs["new"][:, :N].float()).abs().max())
        times = {k: [] for k in libs}
        for _ in range(a.rounds):
            for k in libs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    run(k)
                e1.record()
                torch.cuda.synchronize()
                times[k].append(e0.elapsed_time(e1) / a.iters * 1e3)
        med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
        mn = {k: min(v) for k, v in times.items()}
        print(f"{name:9s} N={N:4d} K={K:4d}: base {med['base']:7.1f} us (min {mn['base']:7.1f})  new {med['new']:7.1f} us "
              f"(min {mn['new']:7.1f})  new/base {med['new'] / med['base']:.3f}  {2.0 * M * N * K / med['new'] / 1e6:7.1f} TF/s  "
              f"identical={same} max|d|={maxd:.3g}", flush=True)


if __name__ == "__main__":
    main()
