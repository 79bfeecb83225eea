"""Torch-tensor front ends of the C-ABI kernels (device pointers + current HIP stream).

PyTorch only provides device memory and the stream here; all arithmetic runs in
libtranscodon_hip.so.  Every function requires CUDA(HIP) tensors and raises otherwise.
"""

import torch

from . import _lib
from ._lib import TC_BF16, TC_F32, check

AA_ORDER = "FLIMVPTAY_HQNKDECWRSG"      # amino-acid codes of tc_codon_decode
_AA_CODE = {c: i for i, c in enumerate(AA_ORDER)}


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return TC_BF16
    if t.dtype == torch.float32:
        return TC_F32
    raise TypeError(f"unsupported element type {t.dtype}")


def _dev(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.HipLibraryError("transcodon_amd kernels need tensors on the GPU (no CPU fallback)")
        if t is not None and not t.is_contiguous():
            raise ValueError("tensor must be contiguous")


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def embed(tokens, species, tok_table, species_table, pad_idx=1):
    """x[B*T,E] f32, pad_flags[B*T] u8, bad[1] i32   (model.py:178-206)."""
    tokens = tokens.to(torch.int32).contiguous()
    species = species.to(torch.int32).contiguous()
    _dev(tokens, species, tok_table, species_table)
    B, T = tokens.shape
    E = tok_table.shape[1]
    x = torch.empty(B * T, E, dtype=torch.float32, device=tokens.device)
    pad = torch.empty(B * T, dtype=torch.uint8, device=tokens.device)
    bad = torch.zeros(1, dtype=torch.int32, device=tokens.device)
    check(_lib.load().tc_embed_fwd(_p(tokens), _p(species), _p(tok_table), _p(species_table), _p(x),
                                   _p(pad), _p(bad), B, T, E, tok_table.shape[0],
                                   species_table.shape[0], pad_idx, _stream()), "tc_embed_fwd")
    return x, pad, bad


def layernorm(x, gamma, beta, eps=1e-5, out_dtype=torch.float32):
    _dev(x, gamma, beta)
    M, E = x.shape
    out = torch.empty(M, E, dtype=out_dtype, device=x.device)
    f32 = out if out_dtype == torch.float32 else None
    b16 = out if out_dtype == torch.bfloat16 else None
    check(_lib.load().tc_layernorm_fwd(_p(x), _p(gamma), _p(beta), _p(f32), _p(b16), M, E, eps,
                                       _stream()), "tc_layernorm_fwd")
    return out


def gemm(a, w, bias, epilogue=_lib.EPI_BIAS, residual=None, out=None, q_scale=1.0,
         rope_cos=None, rope_sin=None, T=0, head_dim=0):
    """epilogue(a[M,K] @ w[N,K]^T + bias[N]); see tc_gemm."""
    _dev(a, w, bias, residual, out, rope_cos, rope_sin)
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and a.dtype == w.dtype
    if out is None:
        odt = torch.float32 if epilogue == _lib.EPI_BIAS_RESIDUAL else a.dtype
        out = torch.empty(M, N, dtype=odt, device=a.device)
    check(_lib.load().tc_gemm(_dt(a), epilogue, _p(a), a.stride(0), _p(w), _p(bias), _p(residual),
                              _p(out), out.stride(0), M, N, K, q_scale, _p(rope_cos), _p(rope_sin),
                              T, head_dim, _stream()), "tc_gemm")
    return out


def key_mask_words(pad_flags, B, T):
    _dev(pad_flags)
    words = torch.empty(B, (T + 63) // 64, dtype=torch.int64, device=pad_flags.device)
    check(_lib.load().tc_key_mask_words(_p(pad_flags), _p(words), B, T, _stream()), "tc_key_mask_words")
    return words


def attention(qkv, B, T, H, key_words=None, q2=False):
    """qkv[B*T, 3*H*64] (q scaled, q/k rotated) -> out[B*T, H*64].  q2: the q columns carry
    log2(e) / sqrt(64) instead of 1 / sqrt(64) (tc_attn_fwd_q2, bf16 only)."""
    _dev(qkv, key_words)
    E = H * 64
    out = torch.empty(B * T, E, dtype=qkv.dtype, device=qkv.device)
    fn = _lib.load().tc_attn_fwd_q2 if q2 else _lib.load().tc_attn_fwd
    check(fn(_dt(qkv), _p(qkv), qkv.stride(0), _p(key_words), _p(out), B, T, H, 64, _stream()),
          "tc_attn_fwd_q2" if q2 else "tc_attn_fwd")
    return out


def attention_scores(qkv, B, T, H, pad_flags=None):
    _dev(qkv, pad_flags)
    s = torch.empty(B, H, T, T, dtype=torch.float32, device=qkv.device)
    check(_lib.load().tc_attn_scores(_dt(qkv), _p(qkv), qkv.stride(0), _p(pad_flags), _p(s), B, T, H, 64,
                                     _stream()), "tc_attn_scores")
    return s


def head_logits(g, col0, E, gamma, beta, W, bias, eps=1e-5):
    _dev(g, gamma, beta, W, bias)
    M = g.shape[0]
    V = W.shape[0]
    out = torch.empty(M, V, dtype=torch.float32, device=g.device)
    check(_lib.load().tc_head_logits_fwd(_dt(g), _p(g), g.stride(0), col0, _p(gamma), _p(beta), _p(W),
                                         _p(bias), _p(out), M, E, V, eps, _stream()),
          "tc_head_logits_fwd")
    return out


def mean_pool(rep, start, length):
    """rep f32 [B,T,E]; start,length int32 [B] -> pooled f32 [B,E] (tc_mean_pool)."""
    start = start.to(torch.int32).contiguous()
    length = length.to(torch.int32).contiguous()
    _dev(rep, start, length)
    B, T, E = rep.shape
    out = torch.empty(B, E, dtype=torch.float32, device=rep.device)
    check(_lib.load().tc_mean_pool(_p(rep), _p(start), _p(length), _p(out), B, T, E, _stream()), "tc_mean_pool")
    return out


def mean_pool_packed(rep, row_start, length):
    """rep f32 [M,E] (packed rows); row_start, length int32 [n] -> pooled f32 [n,E] (tc_mean_pool_packed)."""
    row_start = row_start.to(torch.int32).contiguous()
    length = length.to(torch.int32).contiguous()
    _dev(rep, row_start, length)
    M, E = rep.shape
    n = row_start.numel()
    out = torch.empty(n, E, dtype=torch.float32, device=rep.device)
    check(_lib.load().tc_mean_pool_packed(_p(rep), _p(row_start), _p(length), _p(out), n, M, E, _stream()),
          "tc_mean_pool_packed")
    return out


def encode_amino_acids(aa_seq: str) -> bytes:
    """Protein letters -> tc_codon_decode codes; unknown letters raise KeyError like
    the reference's dict lookup (infer_utr.py:108)."""
    return bytes(_AA_CODE[c] for c in aa_seq)


def codon_decode(logits2d, aa_codes, aa_off, row0):
    """logits2d f32 [rows, >=10]; aa_codes u8 [n]; aa_off i32 [n_seq+1]; row0 i32 [n_seq]
    -> bases u8 [3n] (ASCII).  See tc_codon_decode."""
    _dev(logits2d, aa_codes, aa_off, row0)
    assert logits2d.dtype == torch.float32
    n = aa_codes.numel()
    bases = torch.empty(3 * n, dtype=torch.uint8, device=logits2d.device)
    check(_lib.load().tc_codon_decode(_p(logits2d), logits2d.stride(0), _p(aa_codes), _p(aa_off),
                                      _p(row0), _p(bases), row0.numel(), n, _stream()),
          "tc_codon_decode")
    return bases


def ce_loss(logits2d, labels, want_grad=False, grad_scale=1.0):
    """Returns (loss_sum f32[1], count i32[1], dlogits or None); mean = sum / count."""
    labels = labels.to(torch.int32).contiguous()
    _dev(logits2d, labels)
    M, V = logits2d.shape
    loss = torch.zeros(1, dtype=torch.float32, device=logits2d.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=logits2d.device)
    dl = torch.empty_like(logits2d) if want_grad else None
    check(_lib.load().tc_ce_loss_fwd_bwd(_p(logits2d), _p(labels), _p(loss), _p(cnt), _p(dl),
                                         grad_scale, M, V, _stream()), "tc_ce_loss_fwd_bwd")
    return loss, cnt, dl


# ---- training-step kernels -----------------------------------------------------------------
def gemm_train(a, w, epilogue, bias=None, aux=None, out=None, out2=None, splits=1):
    """tc_gemm_train: EPI_PLAIN_F32 (f32 out, optional split-K slabs [splits, M, N]),
    EPI_BIAS_GELU_SAVE (returns (gelu, pre)), EPI_DGELU (aux = pre-activation; out2: optional f32 [N]
    that receives += column sums of the result)."""
    _dev(a, w, bias, aux, out, out2)
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and a.dtype == w.dtype
    if out is None:
        if epilogue == _lib.EPI_PLAIN_F32:
            out = torch.empty((splits, M, N) if splits > 1 else (M, N), dtype=torch.float32, device=a.device)
        else:
            out = torch.empty(M, N, dtype=a.dtype, device=a.device)
    if epilogue == _lib.EPI_BIAS_GELU_SAVE and out2 is None:
        out2 = torch.empty(M, N, dtype=a.dtype, device=a.device)
    check(_lib.load().tc_gemm_train(_dt(a), epilogue, _p(a), a.stride(0), _p(w), _p(bias), _p(aux), _p(out),
                                    _p(out2), N, M, N, K, splits, _stream()), "tc_gemm_train")
    return (out, out2) if epilogue == _lib.EPI_BIAS_GELU_SAVE else out


def transpose_cast(x, out_dtype, want_rowmajor=False, want_colsum=False, colsum=None, pad_to=64):
    """x [M,N] -> (x^T [N, ldt] zero-padded to a multiple of pad_to columns, row-major copy or
    None, column sums f32 [N] or None)."""
    _dev(x, colsum)
    M, N = x.shape
    ldt = (M + pad_to - 1) // pad_to * pad_to
    out_t = torch.empty(N, ldt, dtype=out_dtype, device=x.device)
    rm = torch.empty(M, N, dtype=out_dtype, device=x.device) if want_rowmajor else None
    if want_colsum and colsum is None:
        colsum = torch.zeros(N, dtype=torch.float32, device=x.device)
    check(_lib.load().tc_transpose_cast(_dt(x), TC_BF16 if out_dtype == torch.bfloat16 else TC_F32, _p(x),
                                        x.stride(0), _p(out_t), ldt, _p(rm), N, _p(colsum), M, N, _stream()),
          "tc_transpose_cast")
    return out_t, rm, colsum


def reduce_slabs(slabs, out=None, accumulate=False):
    _dev(slabs, out)
    S = slabs.shape[0]
    n = slabs[0].numel()
    if out is None:
        out = torch.empty_like(slabs[0])
        accumulate = False
    check(_lib.load().tc_reduce_slabs(_p(slabs), _p(out), S, n, int(accumulate), _stream()), "tc_reduce_slabs")
    return out


def layernorm_bwd(x, gamma, dy, dgamma, dbeta, dx=None, accumulate=False, eps=1e-5, dx_bf16=None, dx_colsum=None):
    """x and dy may be column slices of wider buffers (row stride = leading dimension); dy f32 or bf16.
    dx_bf16 [M,E] bf16 / dx_colsum [E] f32 (accumulated): optional bf16 copy and column sums of the final dx."""
    _dev(gamma, dy, dgamma, dbeta, dx, dx_bf16, dx_colsum)
    if not x.is_cuda or x.stride(1) != 1 or dy.stride(1) != 1:
        raise ValueError("x must be a GPU tensor, x and dy with unit column stride")
    M, E = dy.shape
    if dx is None:
        dx = torch.empty(M, E, dtype=torch.float32, device=dy.device)
        accumulate = False
    check(_lib.load().tc_layernorm_bwd_dy(_dt(x), _p(x), x.stride(0), _p(gamma), _dt(dy), _p(dy), dy.stride(0), _p(dx),
                                          int(accumulate), _p(dgamma), _p(dbeta), _p(dx_bf16), _p(dx_colsum), M, E, eps,
                                          _stream()),
          "tc_layernorm_bwd")
    return dx


def dgelu(dy, u):
    _dev(dy, u)
    out = torch.empty_like(u)
    check(_lib.load().tc_dgelu(_dt(u), _p(dy), _p(u), _p(out), u.numel(), _stream()), "tc_dgelu")
    return out


def head_logits_bwd(g, col0, E, gamma, beta, W, dlogits, dW, dbias, eps=1e-5):
    _dev(g, gamma, beta, W, dlogits, dW, dbias)
    M, V = dlogits.shape
    d_ln = torch.empty(M, E, dtype=torch.float32, device=g.device)
    check(_lib.load().tc_head_logits_bwd(_dt(g), _p(g), g.stride(0), col0, _p(gamma), _p(beta), _p(W), _p(dlogits),
                                         _p(d_ln), _p(dW), _p(dbias), M, E, V, eps, _stream()),
          "tc_head_logits_bwd")
    return d_ln


def embed_bwd(tokens, species, dx, d_tok, d_sp, pad_idx=1):
    tokens = tokens.to(torch.int32).contiguous()
    species = species.to(torch.int32).contiguous()
    _dev(tokens, species, dx, d_tok, d_sp)
    B, T = tokens.shape
    E = dx.shape[-1]
    check(_lib.load().tc_embed_bwd(_p(tokens), _p(species), _p(dx), _p(d_tok), _p(d_sp), B, T, E, d_tok.shape[0],
                                   d_sp.shape[0], pad_idx, _stream()), "tc_embed_bwd")


def rope_bwd(dqkv, cos, sin, T, E, head_dim, q_scale):
    _dev(dqkv, cos, sin)
    check(_lib.load().tc_rope_bwd(_dt(dqkv), _p(dqkv), dqkv.stride(0), _p(cos), _p(sin), dqkv.shape[0], T, E,
                                  head_dim, q_scale, _stream()), "tc_rope_bwd")
    return dqkv


def attention_train(qkv, B, T, H, key_words=None):
    """Forward that also returns lse f32 [B,H,T] for attention_bwd (tc_attn_fwd_train)."""
    _dev(qkv, key_words)
    E = H * 64
    out = torch.empty(B * T, E, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(B, H, T, dtype=torch.float32, device=qkv.device)
    check(_lib.load().tc_attn_fwd_train(_dt(qkv), _p(qkv), qkv.stride(0), _p(key_words), _p(out), _p(lse),
                                        B, T, H, 64, _stream()), "tc_attn_fwd_train")
    return out, lse


def attention_bwd(qkv, out, d_out, lse, B, T, H, key_words=None, rope_cos=None, rope_sin=None, q_scale=1.0):
    """Gradient of attention() with respect to its qkv input (tc_attn_bwd) -> d_qkv like qkv; with
    rope tables, with respect to the unrotated, unscaled q/k/v projections."""
    _dev(qkv, out, d_out, lse, key_words, rope_cos, rope_sin)
    E = H * 64
    if out.shape != (B * T, E) or d_out.shape != (B * T, E) or not out.is_contiguous() or not d_out.is_contiguous():
        raise ValueError("attention_bwd: out and d_out must be contiguous [B*T, E]")
    if d_out.dtype != qkv.dtype or out.dtype != qkv.dtype:
        raise ValueError("attention_bwd: out/d_out dtype must match qkv")
    d_qkv = torch.empty(B * T, 3 * E, dtype=qkv.dtype, device=qkv.device)
    delta = torch.empty(B, H, T, dtype=torch.float32, device=qkv.device)
    check(_lib.load().tc_attn_bwd(_dt(qkv), _p(qkv), qkv.stride(0), _p(key_words), _p(out), _p(d_out), _p(lse),
                                  _p(delta), _p(d_qkv), d_qkv.stride(0), _p(rope_cos), _p(rope_sin), q_scale,
                                  B, T, H, 64, _stream()), "tc_attn_bwd")
    return d_qkv


def gemm_tn(a, b, splits=1):
    """slabs[s] = partial a^T b over the s-th part of the rows; a [M,I], b [M,J] bf16 (tc_gemm_tn)."""
    _dev(a, b)
    M, I = a.shape
    J = b.shape[1]
    assert b.shape[0] == M and a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16
    out = torch.empty(splits, I, J, dtype=torch.float32, device=a.device)
    check(_lib.load().tc_gemm_tn(_p(a), a.stride(0), _p(b), b.stride(0), _p(out), J, M, I, J, splits, _stream()),
          "tc_gemm_tn")
    return out


def colsum(x, want_bf16_copy=False, out=None):
    """Column sums f32 [N] of x [M,N] (added to ``out`` when given) and, for f32 x, a bf16 copy."""
    _dev(x, out)
    M, N = x.shape
    if out is None:
        out = torch.zeros(N, dtype=torch.float32, device=x.device)
    copy = torch.empty(M, N, dtype=torch.bfloat16, device=x.device) if want_bf16_copy else None
    check(_lib.load().tc_colsum(_dt(x), _p(x), x.stride(0), _p(copy), N, _p(out), M, N, _stream()), "tc_colsum")
    return (out, copy) if want_bf16_copy else out
