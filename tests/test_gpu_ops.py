"""Per-kernel parity through the C ABI against plain fp32 torch on the CPU.

Tolerances: the f32 kernels must agree with an fp64-accumulated reference to a few f32
ulps of the accumulated magnitude; the bf16 kernels are compared against the same fp32
arithmetic applied to the bf16-rounded inputs (so the only differences are accumulation
order and the bf16 rounding of the output)."""
import json
import math
import os

import numpy as np
import pytest
import torch

from oracle import transcodon_ref as O
from transcodon_amd import _lib, ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# ---------------------------------------------------------------- GEMM
def test_gemm_bf16_fragment_layout_exact():
    """A = shifted identity pattern with small integers, asymmetric W: exact in bf16, so
    any row/column swap or k-slot permutation error shows up as an integer mismatch."""
    M, N, K = 256, 256, 128
    a = torch.zeros(M, K)
    for i in range(M):
        a[i, (3 * i + 1) % K] = 1 + (i % 3)
        a[i, (7 * i + 5) % K] += 2
    w = torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 13 - 6
    w[:, 0] += torch.arange(N) % 5
    bias = torch.arange(N, dtype=torch.float32) % 7
    out = ops.gemm(a.to(DEV, torch.bfloat16), w.to(DEV, torch.bfloat16), bias.to(DEV))
    ref = a @ w.t() + bias
    assert torch.equal(out.float().cpu(), ref)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,N,K", [(131, 128, 128), (1000, 768, 768), (257, 384, 3072), (64, 2304, 768)])
def test_gemm_bias_and_gelu(dtype, M, N, K):
    a, w, b = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=0.05), _rand(N, seed=3)
    if dtype == torch.bfloat16:
        a, w = _bf(a), _bf(w)
    ref = (a.double() @ w.double().t() + b.double())
    for epi, f in ((_lib.EPI_BIAS, lambda x: x), (_lib.EPI_BIAS_GELU, lambda x: O.gelu(x))):
        out = ops.gemm(a.to(DEV, dtype), w.to(DEV, dtype), b.to(DEV), epi).float().cpu()
        want = f(ref).float()
        if dtype == torch.bfloat16:
            np.testing.assert_allclose(out.numpy(), want.numpy(), rtol=1e-2, atol=2e-2)
        else:
            np.testing.assert_allclose(out.numpy(), want.numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gemm_residual_in_place(dtype):
    M, N, K = 300, 256, 512
    a, w, b, r = _rand(M, K, seed=4), _rand(N, K, seed=5, scale=0.05), _rand(N, seed=6), _rand(M, N, seed=7)
    if dtype == torch.bfloat16:
        a, w = _bf(a), _bf(w)
    ref = (r.double() + (a.double() @ w.double().t() + b.double())).float()
    res = r.to(DEV).clone()
    out = ops.gemm(a.to(DEV, dtype), w.to(DEV, dtype), b.to(DEV), _lib.EPI_BIAS_RESIDUAL, residual=res, out=res)
    assert out.data_ptr() == res.data_ptr() and out.dtype == torch.float32
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=2e-5, atol=3e-5)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gemm_qkv_rope(dtype):
    B, T, H = 3, 50, 2
    E, D = H * 64, 64
    M = B * T
    x, w, b = _rand(M, E, seed=8), _rand(3 * E, E, seed=9, scale=0.1), _rand(3 * E, seed=10, scale=0.1)
    if dtype == torch.bfloat16:
        x, w = _bf(x), _bf(w)
    freqs = O.rope_freqs(D)
    ang = torch.arange(T, dtype=torch.float32)[:, None] * freqs[None, :]
    out = ops.gemm(x.to(DEV, dtype), w.to(DEV, dtype), b.to(DEV), _lib.EPI_QKV_ROPE, q_scale=0.125,
                   rope_cos=ang.cos().contiguous().to(DEV), rope_sin=ang.sin().contiguous().to(DEV),
                   T=T, head_dim=D).float().cpu()
    y = x @ w.t() + b
    q, k, v = y[:, :E] * 0.125, y[:, E:2 * E], y[:, 2 * E:]

    def rot(t):  # [M,E] -> rope per head with position = row % T
        t = t.view(B, T, H, D).permute(0, 2, 1, 3)
        return O.rope_rotate(t, freqs).permute(0, 2, 1, 3).reshape(M, E)

    ref = torch.cat([rot(q), rot(k), v], dim=1)
    tol = dict(rtol=1e-2, atol=2e-2) if dtype == torch.bfloat16 else dict(rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(out.numpy(), ref.numpy(), **tol)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,N,K", [(22037, 768, 768), (16384 + 5, 2304, 768), (33000, 512, 3072)])
def test_gemm_large_tile_kernel(dtype, M, N, K):
    """Shapes with >= 256 tiles of 256 x 256 take the persistent large-tile kernel; M is not a
    multiple of the tile so the row tail is exercised.  All four epilogues."""
    B_T = 61                                  # "sequence length" for the rotary positions
    a, w, b = _rand(M, K, seed=31), _rand(N, K, seed=32, scale=0.05), _rand(N, seed=33)
    if dtype == torch.bfloat16:
        a, w = _bf(a), _bf(w)
    y = (a @ w.t() + b)                       # fp32 CPU reference
    tol = dict(rtol=1e-2, atol=3e-2) if dtype == torch.bfloat16 else dict(rtol=5e-5, atol=5e-5)
    ad, wd, bd = a.to(DEV, dtype), w.to(DEV, dtype), b.to(DEV)
    out = ops.gemm(ad, wd, bd, _lib.EPI_BIAS).float().cpu()
    np.testing.assert_allclose(out.numpy(), y.numpy(), **tol)
    out = ops.gemm(ad, wd, bd, _lib.EPI_BIAS_GELU).float().cpu()
    np.testing.assert_allclose(out.numpy(), O.gelu(y).numpy(), **tol)
    r = _rand(M, N, seed=34)
    res = r.to(DEV).clone()
    ops.gemm(ad, wd, bd, _lib.EPI_BIAS_RESIDUAL, residual=res, out=res)
    np.testing.assert_allclose(res.cpu().numpy(), (r + y).numpy(), **tol)
    if N % 3 == 0 and (N // 3) % 64 == 0:
        E, D = N // 3, 64
        freqs = O.rope_freqs(D)
        ang = torch.arange(B_T, dtype=torch.float32)[:, None] * freqs[None, :]
        out = ops.gemm(ad, wd, bd, _lib.EPI_QKV_ROPE, q_scale=0.125, rope_cos=ang.cos().contiguous().to(DEV),
                       rope_sin=ang.sin().contiguous().to(DEV), T=B_T, head_dim=D).float().cpu()
        pos = torch.arange(M) % B_T
        cs, sn = ang.cos()[pos].repeat_interleave(2, -1), ang.sin()[pos].repeat_interleave(2, -1)   # [M,64]

        def rot(t):
            t = t.view(M, E // D, D)
            rh = torch.stack((-t[..., 1::2], t[..., 0::2]), -1).reshape(t.shape)
            return (t * cs[:, None, :] + rh * sn[:, None, :]).reshape(M, E)

        ref = torch.cat([rot(y[:, :E] * 0.125), rot(y[:, E:2 * E]), y[:, 2 * E:]], 1)
        np.testing.assert_allclose(out.numpy(), ref.numpy(), **tol)


@pytest.mark.parametrize("T", [256, 512, 1024])
def test_gemm_large_tile_rope_aligned_sequences(T):
    """Sequence length a multiple of the 256-row tile: the large-tile kernel takes the rotary
    cos/sin of a tile's first rows through LDS and steps the other row blocks by angle addition
    (rows 16 and 80 of the tables).  Checked against the table formula row by row; M is not a
    multiple of the tile, so the last tile's masked rows are exercised as well."""
    M, E, D = 128 * 256 + 200 * 256 + 77, 768, 64       # >= 256 tiles of 256 x 256
    N, K = 3 * E, 768
    a, w, b = _bf(_rand(M, K, seed=41)), _bf(_rand(N, K, seed=42, scale=0.05)), _rand(N, seed=43)
    freqs = O.rope_freqs(D)
    ang = torch.arange(T, dtype=torch.float32)[:, None] * freqs[None, :]
    out = ops.gemm(a.to(DEV, torch.bfloat16), w.to(DEV, torch.bfloat16), b.to(DEV), _lib.EPI_QKV_ROPE, q_scale=0.125,
                   rope_cos=ang.cos().contiguous().to(DEV), rope_sin=ang.sin().contiguous().to(DEV), T=T,
                   head_dim=D).float().cpu()
    y = a @ w.t() + b
    pos = torch.arange(M) % T
    cs, sn = ang.cos()[pos].repeat_interleave(2, -1), ang.sin()[pos].repeat_interleave(2, -1)

    def rot(t):
        t = t.view(M, E // D, D)
        rh = torch.stack((-t[..., 1::2], t[..., 0::2]), -1).reshape(t.shape)
        return (t * cs[:, None, :] + rh * sn[:, None, :]).reshape(M, E)

    ref = torch.cat([rot(y[:, :E] * 0.125), rot(y[:, E:2 * E]), y[:, 2 * E:]], 1)
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=1e-2, atol=3e-2)


@pytest.mark.parametrize("epi", ["bias", "gelu", "residual"])
def test_gemm_large_tile_padded_rows(epi):
    """Leading dimensions larger than the row length on both sides (the encoder pads its FFN
    intermediate): results land in the right columns and the padding is left untouched."""
    M, N, K, pad_a, pad_o = 66000, 1024, 512, 64, 72
    a, w, b = _bf(_rand(M, K, seed=51)), _bf(_rand(N, K, seed=52, scale=0.05)), _rand(N, seed=53)
    abuf = torch.full((M, K + pad_a), 7.0, dtype=torch.bfloat16, device=DEV)
    abuf[:, :K] = a.to(DEV, torch.bfloat16)
    y = a @ w.t() + b
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    wd, bd = w.to(DEV, torch.bfloat16), b.to(DEV)
    if epi == "residual":
        r = _rand(M, N + pad_o, seed=54)
        obuf = r.to(DEV).clone()
        rc = lib.tc_gemm(1, _lib.EPI_BIAS_RESIDUAL, abuf.data_ptr(), K + pad_a, wd.data_ptr(), bd.data_ptr(),
                         obuf.data_ptr(), obuf.data_ptr(), N + pad_o, M, N, K, 1.0, None, None, 0, 0, s)
        assert rc == 0
        got = obuf.cpu()
        np.testing.assert_allclose(got[:, :N].numpy(), (r[:, :N] + y).numpy(), rtol=1e-2, atol=3e-2)
        assert torch.equal(got[:, N:], r[:, N:])
    else:
        obuf = torch.full((M, N + pad_o), -3.0, dtype=torch.bfloat16, device=DEV)
        rc = lib.tc_gemm(1, _lib.EPI_BIAS if epi == "bias" else _lib.EPI_BIAS_GELU, abuf.data_ptr(), K + pad_a,
                         wd.data_ptr(), bd.data_ptr(), None, obuf.data_ptr(), N + pad_o, M, N, K, 1.0, None, None, 0, 0, s)
        assert rc == 0
        got = obuf.float().cpu()
        want = y if epi == "bias" else O.gelu(y)
        np.testing.assert_allclose(got[:, :N].numpy(), want.numpy(), rtol=1e-2, atol=3e-2)
        assert bool((got[:, N:] == -3.0).all())


def test_one_wave_per_simd_large_tile_equals_the_alternating_kernel(tmp_path):
    """gemm_w.hip (four waves of 128 x 128, accumulators in the accumulation registers; measured 6-12 % slower in
    wall time than the alternating kernel and therefore off unless TC_GEMM_W=1) accumulates every output element in
    the same k order as the alternating kernel: bit-identical results, a row tail in the last tile included."""
    import os
    import subprocess
    import sys
    M, N, K = 66000 + 37, 1024, 768
    g = torch.Generator().manual_seed(12)
    a = torch.randn(M, K, generator=g).to(DEV, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(DEV, torch.bfloat16)
    bias = torch.randn(N, generator=g).to(DEV)
    want = ops.gemm(a, w, bias, epilogue=_lib.EPI_BIAS)
    ref = a.float() @ w.float().T + bias
    assert float((want.float() - ref).abs().max()) <= 0.02 + 0.01 * float(ref.abs().max())
    torch.save({"a": a.cpu(), "w": w.cpu(), "bias": bias.cpu()}, tmp_path / "in.pt")
    code = ("import torch, sys; from transcodon_amd import _lib, ops; d = torch.load(sys.argv[1]); "
            "o = ops.gemm(d['a'].cuda(), d['w'].cuda(), d['bias'].cuda(), epilogue=_lib.EPI_BIAS); "
            "torch.save(o.cpu(), sys.argv[2])")
    env = dict(os.environ, TC_GEMM_W="1", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    subprocess.run([sys.executable, "-c", code, str(tmp_path / "in.pt"), str(tmp_path / "out.pt")], check=True, env=env, timeout=240)
    got = torch.load(tmp_path / "out.pt")
    assert torch.equal(got, want.cpu())


def test_gelu_fast_accuracy():
    """The bf16 GEMM epilogue uses a rational erf; check it against exact erf on a dense grid
    by running a GEMM with an identity weight (values land on the bf16 grid, error <= 1 bf16 ulp)."""
    K = N = 256
    M = 65536
    x = torch.linspace(-9, 9, M * K).view(M, K)
    xb = _bf(x)
    eye = torch.eye(N, K)
    out = ops.gemm(xb.to(DEV, torch.bfloat16), eye.to(DEV, torch.bfloat16), torch.zeros(N, device=DEV),
                   _lib.EPI_BIAS_GELU).float().cpu()
    ref = O.gelu(xb.double()).float()
    err = (out - ref).abs()
    bound = ref.abs() * 2.0 ** -8 + 1e-6
    assert bool((err <= bound).all()), float((err - bound).max())


def test_gemm_rejects_bad_shapes():
    a = torch.zeros(8, 64, device=DEV, dtype=torch.bfloat16)
    w = torch.zeros(100, 64, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(_lib.HipLibraryError):
        ops.gemm(a, w, torch.zeros(100, device=DEV))


# ---------------------------------------------------------------- row kernels
def test_embed_pad_zero_and_bad_index():
    V, S, E = 10, 16, 128
    tokt, spt = _rand(V, E, seed=11), _rand(S, E, seed=12)
    tok = torch.tensor([[0, 5, 6, 9, 2, 1, 1], [0, 8, 7, 4, 5, 6, 2]])
    sp = torch.tensor([3, 15])
    x, pad, bad = ops.embed(tok.to(DEV), sp.to(DEV), tokt.to(DEV), spt.to(DEV))
    ref = (tokt[tok] + spt[sp][:, None, :]) * (tok != 1)[..., None]
    assert torch.equal(x.cpu().view(2, 7, E), ref)
    assert pad.cpu().view(2, 7).tolist() == (tok == 1).int().tolist() and int(bad.item()) == 0
    tok[0, 2] = 10
    x, _, bad = ops.embed(tok.to(DEV), sp.to(DEV), tokt.to(DEV), spt.to(DEV))
    assert int(bad.item()) == 1 and float(x.view(2, 7, E)[0, 2].abs().sum()) == 0.0


@pytest.mark.parametrize("E", [128, 768, 1024])
def test_layernorm(E):
    M = 37
    x, g, b = _rand(M, E, seed=13, scale=3.0) + 0.5, _rand(E, seed=14), _rand(E, seed=15)
    ref = torch.nn.functional.layer_norm(x, (E,), g, b, 1e-5)
    y = ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), 1e-5, torch.float32).cpu()
    np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
    yb = ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), 1e-5, torch.bfloat16).float().cpu()
    np.testing.assert_allclose(yb.numpy(), ref.numpy(), rtol=8e-3, atol=8e-3)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_head_logits(dtype):
    M, E, V = 45, 768, 10
    gbuf = _rand(M, 2 * E, seed=16)
    if dtype == torch.bfloat16:
        gbuf = _bf(gbuf)
    gam, bet, W, bias = _rand(E, seed=17), _rand(E, seed=18), _rand(V, E, seed=19, scale=0.05), _rand(V, seed=20)
    for col0 in (0, E):
        ref = torch.nn.functional.layer_norm(gbuf[:, col0:col0 + E], (E,), gam, bet, 1e-5) @ W.t() + bias
        out = ops.head_logits(gbuf.to(DEV, dtype), col0, E, gam.to(DEV), bet.to(DEV), W.to(DEV), bias.to(DEV)).cpu()
        np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=1e-4, atol=1e-4)


def test_key_mask_words():
    B, T = 3, 150
    pad = torch.zeros(B, T, dtype=torch.uint8)
    pad[0, 100:] = 1
    pad[1, 64:] = 1
    pad[2, 7] = 1
    words = ops.key_mask_words(pad.to(DEV).view(-1), B, T).cpu().numpy().view(np.uint64)
    for b in range(B):
        for w in range(3):
            want = 0
            for j in range(64):
                k = 64 * w + j
                if k < T and pad[b, k] == 0:
                    want |= 1 << j
            assert int(words[b, w]) == want


# ---------------------------------------------------------------- attention
def _attn_ref(qkv, B, T, H, pad):
    E = H * 64
    q, k, v = (qkv[:, i * E:(i + 1) * E].view(B, T, H, 64).permute(0, 2, 1, 3).double() for i in range(3))
    s = q @ k.transpose(-1, -2)
    if pad is not None:
        s = s.masked_fill(pad.bool()[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    return (p @ v).permute(0, 2, 1, 3).reshape(B * T, E).float(), s.float()


LOG2E = 1.4426950408889634


def _q2_pair(qkv, E):
    """bf16 input whose q columns carry the extra factor log2(e) (what tc_attn_fwd_q2 takes), and the f32
    input with exactly those rounded values divided by log2(e) again, for the float64 reference."""
    x = qkv.clone()
    x[:, :E] *= LOG2E
    x = _bf(x)
    ref_in = x.double()
    ref_in[:, :E] /= LOG2E
    return x, ref_in


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32, "bf16_q2"])
@pytest.mark.parametrize("B,T,H,lens", [(2, 131, 2, None), (3, 256, 3, [256, 130, 1]), (2, 70, 1, [70, 64]),
                                        (1, 515, 2, None), (2, 300, 2, [300, 257]), (2, 200, 1, [0, 200])])
def test_attention_forward(dtype, B, T, H, lens):
    E = H * 64
    q2 = dtype == "bf16_q2"
    if q2:
        dtype = torch.bfloat16
    qkv = _rand(B * T, 3 * E, seed=21)
    qkv[:, :E] *= 0.25
    ref_in = None
    if q2:
        qkv, ref_in = _q2_pair(qkv, E)
    elif dtype == torch.bfloat16:
        qkv = _bf(qkv)
    pad = None
    if lens is not None:
        pad = torch.zeros(B, T, dtype=torch.uint8)
        for b, L in enumerate(lens):
            pad[b, L:] = 1
    ref, sref = _attn_ref(qkv if ref_in is None else ref_in, B, T, H, pad)
    words = ops.key_mask_words(pad.to(DEV).view(-1), B, T) if pad is not None else None
    out = ops.attention(qkv.to(DEV, dtype), B, T, H, words, q2=q2).float().cpu()
    keep = torch.ones(B * T, dtype=torch.bool) if pad is None else (pad.view(-1) == 0)
    tol = dict(rtol=2e-2, atol=2e-2) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out[keep].numpy(), ref[keep].numpy(), **tol)
    if q2:
        with pytest.raises(_lib.HipLibraryError):       # the base-2 entry is bf16 only
            ops.attention(qkv.to(DEV, torch.float32), B, T, H, words, q2=True)
        return
    sc = ops.attention_scores(qkv.to(DEV, dtype), B, T, H, pad.to(DEV).view(-1) if pad is not None else None).cpu()
    finite = torch.isfinite(sref)
    assert torch.equal(torch.isfinite(sc), finite)
    np.testing.assert_allclose(sc[finite].numpy(), sref[finite].numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("q2", [False, True])
def test_attention_online_softmax_rescale(q2):
    """A key far above the others late in the sequence forces the running-max rescale."""
    B, T, H = 1, 192, 1
    qkv = _bf(_rand(T, 192, seed=22))
    qkv[:, :64] *= 0.1
    qkv[150, 64:128] = _bf(qkv[5, :64] * 40.0)      # key 150 aligned with query 5
    ref_in = qkv
    if q2:
        qkv, ref_in = _q2_pair(qkv, 64)
    ref, _ = _attn_ref(ref_in, B, T, H, None)
    out = ops.attention(qkv.to(DEV, torch.bfloat16), B, T, H, q2=q2).float().cpu()
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("jump", [30.0, 120.0])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32, "bf16_q2"])
def test_attention_lazy_running_maximum_staircase(dtype, jump):
    """The bf16 kernel moves its softmax offset only when a key tile's maximum exceeds it by more
    than 2^8; scores that climb tile after tile (steps below and above that threshold, in both
    directions) must still give the exact softmax.  Checked against float64 on every query."""
    B, T, H = 2, 512, 1
    g = torch.Generator().manual_seed(31)
    q = torch.randn(B * T, 64, generator=g) * 0.05
    k = torch.randn(B * T, 64, generator=g) * 0.05
    v = torch.randn(B * T, 64, generator=g)
    q[:, 0] = 1.0
    # per 64-key tile a shift of the whole tile's scores, in natural-log units: +3.5 and +7 stay below the
    # lazy threshold (8 in log2 units = 5.5 nats) cumulatively crossing it, -20 falls far below, +30 jumps
    # jump = 120 nats (173 in log2 units) overflows an exponential taken against the first tile's offset: the
    # base-2 kernel's loop without tile maxima must notice and repeat itself with them
    steps = torch.tensor([0.0, 3.5, 7.0, 4.0, -20.0, jump, jump + 3.0, 12.0])
    k[:, 0] = steps.repeat_interleave(64).repeat(B)
    qkv = torch.cat([q, k, v], 1)
    if dtype == "bf16_q2":
        # base-2 entry: q carries log2(e); the reference sees exactly the rounded values
        qkv, ref_in = _q2_pair(qkv, 64)
        ref, _ = _attn_ref(ref_in, B, T, H, None)
        out = ops.attention(qkv.to(DEV, torch.bfloat16), B, T, H, q2=True).float().cpu()
        np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=2e-2, atol=2e-2)
        return
    if dtype == torch.bfloat16:
        qkv = _bf(qkv)
    ref, _ = _attn_ref(qkv, B, T, H, None)
    out = ops.attention(qkv.to(DEV, dtype), B, T, H).float().cpu()
    tol = dict(rtol=2e-2, atol=2e-2) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out.numpy(), ref.numpy(), **tol)
    o2, lse = ops.attention_train(qkv.to(DEV, dtype), B, T, H)
    s = (qkv[:, :64].double().view(B, T, 64) @ qkv[:, 64:128].double().view(B, T, 64).transpose(1, 2))
    np.testing.assert_allclose(lse.cpu().numpy()[:, 0], torch.logsumexp(s, -1).numpy(), rtol=1e-3, atol=2e-2)


def test_attention_rejects_sequences_beyond_the_bf16_limit():
    """The bf16 kernels keep one validity word per 64 keys in LDS: 8192 keys at most; the call
    must fail with an error instead of reading past the table."""
    T, H = 8192 + 64, 1
    qkv = torch.zeros(T, 3 * 64, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(_lib.HipLibraryError):
        ops.attention(qkv, 1, T, H)
    out = ops.attention(torch.zeros(8192, 3 * 64, device=DEV, dtype=torch.bfloat16), 1, 8192, H)   # the limit itself runs
    assert torch.isfinite(out.float()).all()


# ---------------------------------------------------------------- decode and loss
def test_decode_known_answers(golden_dir):
    kats = json.load(open(os.path.join(golden_dir, "decode_kat.json")))
    for k in kats:
        logits = torch.tensor(k["logits"], dtype=torch.float32)
        m = len(k["aa"])
        aa = torch.tensor(list(ops.encode_amino_acids(k["aa"])), dtype=torch.uint8, device=DEV)
        bases = ops.codon_decode(logits.to(DEV), aa, torch.tensor([0, m], dtype=torch.int32, device=DEV),
                                 torch.tensor([0], dtype=torch.int32, device=DEV))
        assert bytes(bases.cpu().numpy()).decode() == k["rna"], k["name"]


def test_decode_ragged_batch_against_oracle():
    rng = np.random.default_rng(3)
    letters = list(O.AA2CODONS.keys())
    seqs = ["".join(rng.choice(letters, size=n)) for n in (1, 17, 300, 2049)]
    width = 3 * 2049 + 9
    logits = torch.from_numpy(rng.standard_normal((len(seqs), width, 10)).astype(np.float32) * 3)
    starts = [0, 5, 9, 1]
    offs, row0 = [0], []
    for r, s in enumerate(seqs):
        offs.append(offs[-1] + len(s))
        row0.append(r * width + starts[r])
    aa = torch.tensor(list(b"".join(ops.encode_amino_acids(s) for s in seqs)), dtype=torch.uint8, device=DEV)
    bases = ops.codon_decode(logits.view(-1, 10).to(DEV), aa, torch.tensor(offs, dtype=torch.int32, device=DEV),
                             torch.tensor(row0, dtype=torch.int32, device=DEV))
    flat = bytes(bases.cpu().numpy()).decode()
    for r, s in enumerate(seqs):
        want = O.decode(logits[r, starts[r]:starts[r] + 3 * len(s)], s)
        assert flat[3 * offs[r]:3 * offs[r + 1]] == want


def test_ce_loss_matches_torch():
    M, V = 1000, 10
    logits = _rand(M, V, seed=23, scale=2.0)
    labels = torch.randint(0, V, (M,), generator=torch.Generator().manual_seed(1))
    labels[torch.rand(M, generator=torch.Generator().manual_seed(2)) < 0.7] = -100
    with torch.enable_grad():
        lg = logits.clone().requires_grad_(True)
        ref = torch.nn.functional.cross_entropy(lg, labels, ignore_index=-100)
        ref.backward()
    ref = ref.detach()
    n = int((labels != -100).sum())
    loss, cnt, dl = ops.ce_loss(logits.to(DEV), labels.to(DEV), want_grad=True, grad_scale=1.0 / n)
    assert int(cnt.item()) == n
    np.testing.assert_allclose(float(loss.item()) / n, float(ref), rtol=1e-5)
    np.testing.assert_allclose(dl.cpu().numpy(), lg.grad.numpy(), rtol=1e-4, atol=1e-7)


def test_codon_stats_kernel_equals_host_metrics(golden_dir):
    """GC content, codon recovery and CFD from tc_codon_stats on the device against the host functions,
    themselves pinned by outputs of the reference's evaluation scripts (tests/golden/evaluation.json)."""
    import json
    import os
    from transcodon_amd import evaluation as EV
    with open(os.path.join(golden_dir, "evaluation.json")) as fh:
        g = json.load(fh)
    usage = EV.CodonUsage({k: (v[0], v[1]) for k, v in g["codon_frequencies"].items()})
    cases = [c for c in g["cases"] if len(c["dna"]) % 3 == 0 and len(c["other"]) >= len(c["dna"])]
    assert len(cases) >= 3
    dna = [c["dna"] for c in cases]
    other = [c["other"][:len(c["dna"])] for c in cases]
    off = np.concatenate([[0], np.cumsum([len(d) // 3 for d in dna])]).astype(np.int32)
    up = lambda strs: torch.frombuffer(bytearray("".join(strs).encode()), dtype=torch.uint8).to(DEV)
    m = EV.device_metrics(up(dna), torch.from_numpy(off), ref_bases=up(other), usage=usage, threshold=0.3)
    for i, c in enumerate(cases):
        assert m["gc"][i] == pytest.approx(c["gc"], rel=1e-12)
        assert m["cfd"][i] == pytest.approx(c["cfd"], rel=1e-12)
        assert m["recovery"][i] == pytest.approx(EV.codon_similarity(c["dna"], other[i]), rel=1e-12)
    m5 = EV.device_metrics(up(dna), torch.from_numpy(off), usage=usage, threshold=0.5)
    assert m5["recovery"] is None
    for i, c in enumerate(cases):
        assert m5["cfd"][i] == pytest.approx(c["cfd_05"], rel=1e-12)
    with pytest.raises(KeyError):
        EV.device_metrics(up(["ATGNNN"]), torch.tensor([0, 2], dtype=torch.int32))


def test_minmax_kernel_equals_the_reference_lists_bit_for_bit(golden_dir):
    """%MinMax on the device (tc_codon_minmax) against the lists evaluation/minmax.py:9-79 itself produced for the
    fixture's sequences (window 18 and 5): same f64 operations in the same order, so equality is exact; the tail a
    window does not fit is None as in the reference."""
    import json
    import os
    from transcodon_amd import evaluation as EV
    with open(os.path.join(golden_dir, "evaluation.json")) as fh:
        g = json.load(fh)
    usage = EV.CodonUsage({k: (v[0], v[1]) for k, v in g["codon_frequencies"].items()})
    cases = [c for c in g["cases"] if len(c["dna"]) % 3 == 0]
    assert len(cases) >= 5 and max(len(c["dna"]) for c in cases) >= 900
    dna = [c["dna"] for c in cases]
    off = torch.from_numpy(np.concatenate([[0], np.cumsum([len(d) // 3 for d in dna])]).astype(np.int32))
    up = lambda strs: torch.frombuffer(bytearray("".join(strs).encode()), dtype=torch.uint8).to(DEV)
    for window, key in ((18, "minmax"), (5, "minmax_w5")):
        got = EV.device_min_max(up(dna), off, usage, window_size=window)
        n_exact = 0
        for c, lst in zip(cases, got):
            assert lst == c[key], (len(c["dna"]), window)
            n_exact += sum(v is not None for v in lst)
        assert n_exact > 300
    # RNA letters count as DNA; a letter outside the alphabet is an error, as on the host
    assert EV.device_min_max(up([dna[-1].replace("T", "U")]), off[-2:] - off[-2], usage) == [cases[-1]["minmax"]]
    with pytest.raises(KeyError):
        EV.device_min_max(up(["ATGNNN" * 10]), torch.tensor([0, 20], dtype=torch.int32), usage)
