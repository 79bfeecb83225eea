"""Backward-pass kernels against torch autograd on the CPU (fp32 reference)."""
import numpy as np
import pytest
import torch

from oracle import transcodon_ref as O
from transcodon_amd import _lib, ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rand(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M", [192, 1000, 40000])
def test_linear_backward_via_gemm(dtype, M):
    """dX = dY W and dW = dY^T X (split-K over the token dimension), db = colsum(dY)."""
    N, K = 384, 256
    x, w, dy = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=0.05), _rand(M, N, seed=3)
    if dtype == torch.bfloat16:
        x, w, dy = _bf(x), _bf(w), _bf(dy)
    tol = dict(rtol=2e-2, atol=2e-2) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-4)
    xd, wd, dyd = x.to(DEV, dtype), w.to(DEV, dtype), dy.to(DEV, dtype)
    # data gradient: A = dY [M,N], "W" = W^T [K,N]
    wt, _, _ = ops.transpose_cast(wd, dtype, pad_to=64)
    assert wt.shape == (K, N)
    dx = ops.gemm_train(dyd, wt, _lib.EPI_PLAIN_F32)
    np.testing.assert_allclose(dx.cpu().numpy(), (dy @ w).numpy(), **tol)
    # weight gradient: dW[N,K] = dY^T[N,M] . (X^T[K,M])^T, 4-way split over the tokens
    dyt, _, db = ops.transpose_cast(dyd, dtype, want_colsum=True)
    xt, _, _ = ops.transpose_cast(xd, dtype)
    assert dyt.shape[1] % 64 == 0 and float(dyt[:, M:].float().abs().sum()) == 0.0
    slabs = ops.gemm_train(dyt, xt, _lib.EPI_PLAIN_F32, splits=4)
    dw = ops.reduce_slabs(slabs)
    ref = dy.t() @ x
    scale = float(ref.abs().max())
    np.testing.assert_allclose(dw.cpu().numpy() / scale, ref.numpy() / scale, rtol=0, atol=3e-3 if dtype == torch.bfloat16 else 1e-5)
    np.testing.assert_allclose(db.cpu().numpy(), dy.sum(0).numpy(), rtol=1e-4, atol=1e-3)
    # accumulate into an existing gradient
    acc = torch.ones_like(dw)
    ops.reduce_slabs(slabs, out=acc, accumulate=True)
    np.testing.assert_allclose(acc.cpu().numpy(), dw.cpu().numpy() + 1.0, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gelu_save_and_dgelu_epilogues(dtype):
    M, N, K = 33000, 512, 256       # large enough for the persistent large-tile kernel in bf16
    x, w, b, dy = _rand(M, K, seed=4), _rand(N, K, seed=5, scale=0.08), _rand(N, seed=6), _rand(M, N, seed=7)
    if dtype == torch.bfloat16:
        x, w, dy = _bf(x), _bf(w), _bf(dy)
    tol = dict(rtol=2e-2, atol=2e-2) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-4)
    xd, wd = x.to(DEV, dtype), w.to(DEV, dtype)
    y, pre = ops.gemm_train(xd, wd, _lib.EPI_BIAS_GELU_SAVE, bias=b.to(DEV))
    u = x @ w.t() + b
    np.testing.assert_allclose(pre.float().cpu().numpy(), u.numpy(), **tol)
    np.testing.assert_allclose(y.float().cpu().numpy(), O.gelu(u).numpy(), **tol)
    # gradient through gelu fused in the data-gradient GEMM: d_u = (dY_next W2) * gelu'(u)
    w2 = _rand(K, N, seed=8, scale=0.08)          # next layer weight [out=K, in=N]
    dz = _rand(M, K, seed=9)
    if dtype == torch.bfloat16:
        w2, dz = _bf(w2), _bf(dz)
    w2t, _, _ = ops.transpose_cast(w2.to(DEV, dtype), dtype)     # [N, K]
    du = ops.gemm_train(dz.to(DEV, dtype), w2t, _lib.EPI_DGELU, aux=pre)
    pre_f = pre.float().cpu()
    with torch.enable_grad():
        uu = pre_f.clone().requires_grad_(True)
        (O.gelu(uu) * (dz @ w2)).sum().backward()
    np.testing.assert_allclose(du.float().cpu().numpy(), uu.grad.numpy(), **tol)
    # the same call can add the column sums of its result (the bias gradient of the layer below) to out2
    csum = torch.full((N,), 0.5, device=DEV)
    du_b = ops.gemm_train(dz.to(DEV, dtype), w2t, _lib.EPI_DGELU, aux=pre, out2=csum)
    assert torch.equal(du_b, du)
    want = 0.5 + uu.grad.double().sum(0).numpy()
    np.testing.assert_allclose(csum.cpu().numpy(), want, rtol=2e-2 if dtype == torch.bfloat16 else 1e-4,
                               atol=0.02 * float(np.abs(want).max()) if dtype == torch.bfloat16 else 1e-2)
    # standalone elementwise form
    du2 = ops.dgelu((dz @ w2).to(DEV), pre)
    np.testing.assert_allclose(du2.float().cpu().numpy(), uu.grad.numpy(), **tol)


@pytest.mark.parametrize("xdtype,dydtype", [(torch.float32, torch.float32), (torch.bfloat16, torch.float32),
                                            (torch.float32, torch.bfloat16), (torch.bfloat16, torch.bfloat16)])
@pytest.mark.parametrize("E", [128, 768])
def test_layernorm_backward(xdtype, dydtype, E):
    M = 777
    x, g, b, dy = _rand(M, E, seed=10, scale=2.0) + 0.3, _rand(E, seed=11) + 1.0, _rand(E, seed=12), _rand(M, E, seed=13)
    if xdtype == torch.bfloat16:
        x = _bf(x)
    if dydtype == torch.bfloat16:
        dy = _bf(dy)                      # the reference then sees exactly the values the kernel reads
    with torch.enable_grad():
        xx, gg, bb = x.clone().requires_grad_(True), g.clone().requires_grad_(True), b.clone().requires_grad_(True)
        torch.nn.functional.layer_norm(xx, (E,), gg, bb, 1e-5).backward(dy)
    dgam = torch.zeros(E, device=DEV)
    dbet = torch.zeros(E, device=DEV)
    carry = _rand(M, E, seed=14)
    dx = carry.to(DEV).clone()
    ops.layernorm_bwd(x.to(DEV, xdtype), g.to(DEV), dy.to(DEV, dydtype), dgam, dbet, dx=dx, accumulate=True)
    np.testing.assert_allclose(dx.cpu().numpy(), (carry + xx.grad).numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(dgam.cpu().numpy(), gg.grad.numpy(), rtol=1e-3, atol=2e-3)
    np.testing.assert_allclose(dbet.cpu().numpy(), bb.grad.numpy(), rtol=1e-3, atol=2e-3)
    dx2 = ops.layernorm_bwd(x.to(DEV, xdtype), g.to(DEV), dy.to(DEV, dydtype), dgam, dbet)
    np.testing.assert_allclose(dx2.cpu().numpy(), xx.grad.numpy(), rtol=1e-4, atol=1e-4)
    # fused outputs for the Linear layer below: bf16 copy of the final dx and its column sums (accumulated)
    copy = torch.zeros(M, E, dtype=torch.bfloat16, device=DEV)
    csum = torch.ones(E, device=DEV)
    dx3 = carry.to(DEV).clone()
    ops.layernorm_bwd(x.to(DEV, xdtype), g.to(DEV), dy.to(DEV, dydtype), dgam, dbet, dx=dx3, accumulate=True,
                      dx_bf16=copy, dx_colsum=csum)
    assert torch.equal(dx3, dx)
    assert torch.equal(copy, dx3.to(torch.bfloat16))
    np.testing.assert_allclose(csum.cpu().numpy(), 1.0 + dx3.double().sum(0).cpu().numpy(), rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("E,V", [(768, 10), (768, 3), (128, 10)])
def test_head_backward(dtype, E, V):
    M = 333
    gbuf = _rand(M, 2 * E, seed=15)
    if dtype == torch.bfloat16:
        gbuf = _bf(gbuf)
    gam, bet, W, dl = _rand(E, seed=16) + 1, _rand(E, seed=17), _rand(V, E, seed=18, scale=0.1), _rand(M, V, seed=19)
    col0 = E
    with torch.enable_grad():
        gg = gbuf[:, col0:].clone().requires_grad_(True)
        ww, ga, be = W.clone().requires_grad_(True), gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
        ln = torch.nn.functional.layer_norm(gg, (E,), ga, be, 1e-5)
        ln.retain_grad()
        (ln @ ww.t()).backward(dl)
    dW = torch.zeros(V, E, device=DEV)
    db = torch.zeros(V, device=DEV)
    d_ln = ops.head_logits_bwd(gbuf.to(DEV, dtype), col0, E, gam.to(DEV), bet.to(DEV), W.to(DEV), dl.to(DEV), dW, db)
    np.testing.assert_allclose(d_ln.cpu().numpy(), ln.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dW.cpu().numpy(), ww.grad.numpy(), rtol=1e-3, atol=2e-3)
    np.testing.assert_allclose(db.cpu().numpy(), dl.sum(0).numpy(), rtol=1e-4, atol=1e-3)
    # chain with the LayerNorm backward on the strided view of g
    dgam, dbet = torch.zeros(E, device=DEV), torch.zeros(E, device=DEV)
    gview = gbuf.to(DEV, dtype)[:, col0:]
    dg = ops.layernorm_bwd(gview, gam.to(DEV), d_ln, dgam, dbet)
    np.testing.assert_allclose(dg.cpu().numpy(), gg.grad.numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(dgam.cpu().numpy(), ga.grad.numpy(), rtol=1e-3, atol=2e-3)


def test_embedding_backward():
    B, T, E, V, S = 5, 37, 128, 10, 16
    rng = np.random.default_rng(0)
    tok = torch.from_numpy(rng.integers(0, V, (B, T)))
    tok[:, -5:] = 1
    sp = torch.tensor([3, 3, 0, 15, 7])
    dx = _rand(B * T, E, seed=20)
    d_tok, d_sp = torch.zeros(V, E, device=DEV), torch.zeros(S, E, device=DEV)
    ops.embed_bwd(tok.to(DEV), sp.to(DEV), dx.to(DEV), d_tok, d_sp)
    keep = (tok != 1).reshape(-1)
    ref_tok = torch.zeros(V, E).index_add_(0, tok.reshape(-1)[keep], dx[keep])
    ref_sp = torch.zeros(S, E).index_add_(0, sp.repeat_interleave(T)[keep], dx[keep])
    np.testing.assert_allclose(d_tok.cpu().numpy(), ref_tok.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(d_sp.cpu().numpy(), ref_sp.numpy(), rtol=1e-5, atol=1e-5)
    assert float(d_tok[1].abs().sum()) == 0.0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_rope_backward(dtype):
    B, T, H = 2, 29, 2
    E, D, M = H * 64, 64, B * T
    freqs = O.rope_freqs(D)
    ang = torch.arange(T, dtype=torch.float32)[:, None] * freqs[None, :]
    dqkv = _rand(M, 3 * E, seed=21)
    if dtype == torch.bfloat16:
        dqkv = _bf(dqkv)
    with torch.enable_grad():
        pre = _rand(M, 3 * E, seed=22).requires_grad_(True)

        def rot(t):
            t = t.view(B, T, H, D).permute(0, 2, 1, 3)
            return O.rope_rotate(t, freqs).permute(0, 2, 1, 3).reshape(M, E)

        out = torch.cat([rot(pre[:, :E] * 0.125), rot(pre[:, E:2 * E]), pre[:, 2 * E:]], 1)
        out.backward(dqkv)
    got = ops.rope_bwd(dqkv.to(DEV, dtype).clone(), ang.cos().contiguous().to(DEV), ang.sin().contiguous().to(DEV),
                       T, E, D, 0.125).float().cpu()
    tol = dict(rtol=1e-2, atol=1e-2) if dtype == torch.bfloat16 else dict(rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(got.numpy(), pre.grad.numpy(), **tol)


def _ref_attention(qkv, B, T, H, pad):
    """Plain attention on [B*T, 3E] (q already scaled/rotated), the maths of
    transcodon/multihead_attention.py:362-398."""
    E = H * 64
    q, k, v = (qkv[:, i * E:(i + 1) * E].view(B, T, H, 64).permute(0, 2, 1, 3) for i in range(3))
    s = q @ k.transpose(-1, -2)
    if pad is not None:
        s = s.masked_fill(pad[:, None, None, :], float("-inf"))
    p = torch.softmax(s, -1)
    return (p @ v).permute(0, 2, 1, 3).reshape(B * T, E), torch.logsumexp(s, -1)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,T,H,lens", [
    (2, 29, 2, None),
    (3, 200, 2, [200, 131, 64]),     # one sequence whose last key tiles are all padding
    (2, 512, 12, [512, 300]),
    (1, 129, 1, [70]),
])
def test_attention_backward(dtype, B, T, H, lens):
    E, M = H * 64, B * T
    qkv = _rand(M, 3 * E, seed=31, scale=0.7)
    qkv[:, :E] *= 0.35
    d_out = _rand(M, E, seed=32)
    pad = None
    if lens is not None:
        pad = torch.arange(T)[None, :] >= torch.tensor(lens)[:, None]
    if dtype == torch.bfloat16:
        qkv, d_out = _bf(qkv), _bf(d_out)
    with torch.enable_grad():
        x = qkv.clone().requires_grad_(True)
        ref_out, ref_lse = _ref_attention(x, B, T, H, pad)
        ref_out.backward(d_out)
    words = None
    if pad is not None:
        words = ops.key_mask_words(pad.to(DEV, torch.uint8).contiguous(), B, T)
    qd = qkv.to(DEV, dtype)
    out, lse = ops.attention_train(qd, B, T, H, words)
    np.testing.assert_allclose(lse.cpu().numpy(), ref_lse.detach().numpy(), rtol=1e-2 if dtype == torch.bfloat16
                               else 1e-5, atol=2e-2 if dtype == torch.bfloat16 else 1e-5)
    # the backward is fed the reference forward output so that its error is measured alone
    got = ops.attention_bwd(qd, ref_out.detach().to(DEV, dtype), d_out.to(DEV, dtype), lse, B, T, H, words)
    got = got.float().cpu()
    ref = x.grad
    for name, sl in (("dq", slice(0, E)), ("dk", slice(E, 2 * E)), ("dv", slice(2 * E, 3 * E))):
        g, r = got[:, sl], ref[:, sl]
        err = float((g - r).abs().max()) / max(float(r.abs().max()), 1e-6)
        assert err < (2e-2 if dtype == torch.bfloat16 else 2e-5), (name, err)
    if pad is not None:
        assert float(got[:, E:][pad.reshape(-1)].abs().max()) == 0.0   # padded keys get no gradient


@pytest.mark.parametrize("M,I,J,splits", [(64, 128, 128, 1), (1000, 256, 384, 3), (33000, 768, 128, 16), (77, 128, 256, 4),
                                          (33000, 768, 768, 28), (1000, 256, 512, 96), (4100, 768, 3072, 7),   # 256-tile kernel
                                          # whole 64-token tiles, >= 2 per split: alternating-group 256-tile kernel
                                          (4096, 768, 768, 28), (16384, 768, 3072, 7), (32768, 512, 512, 50),
                                          (65536, 2304, 768, 10)])
def test_gemm_tn_matches_matmul(M, I, J, splits):
    """dW = dY^T X straight from row-major operands (ragged last token tile, more splits than tiles)."""
    a, b = _bf(_rand(M, I, seed=41)), _bf(_rand(M, J, seed=42))
    slabs = ops.gemm_tn(a.to(DEV, torch.bfloat16), b.to(DEV, torch.bfloat16), splits)
    got = ops.reduce_slabs(slabs).cpu()
    ref = a.t().double() @ b.double()
    scale = float(ref.abs().max())
    np.testing.assert_allclose(got.numpy() / scale, ref.numpy() / scale, rtol=0, atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_colsum(dtype):
    M, N = 1234, 200
    x = _rand(M, N, seed=43)
    if dtype == torch.bfloat16:
        x = _bf(x)
    base = torch.ones(N, device=DEV)
    if dtype == torch.float32:
        out, copy = ops.colsum(x.to(DEV), want_bf16_copy=True, out=base)
        assert torch.equal(copy.cpu(), x.to(torch.bfloat16))
    else:
        out = ops.colsum(x.to(DEV, dtype), out=base)
    np.testing.assert_allclose(out.cpu().numpy(), 1.0 + x.double().sum(0).numpy(), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_attention_backward_with_fused_rope(dtype):
    """tc_attn_bwd with rope tables == tc_attn_bwd followed by tc_rope_bwd."""
    B, T, H = 2, 200, 2
    E, M = H * 64, B * T
    qkv = _rand(M, 3 * E, seed=51, scale=0.5).to(DEV, dtype)
    d_out = _rand(M, E, seed=52).to(DEV, dtype)
    freqs = O.rope_freqs(64)
    ang = torch.arange(T, dtype=torch.float32)[:, None] * freqs[None, :]
    cos, sin = ang.cos().contiguous().to(DEV), ang.sin().contiguous().to(DEV)
    pad = (torch.arange(T)[None, :] >= torch.tensor([T, 150])[:, None])
    words = ops.key_mask_words(pad.to(DEV, torch.uint8).contiguous(), B, T)
    out, lse = ops.attention_train(qkv, B, T, H, words)
    plain = ops.attention_bwd(qkv, out, d_out, lse, B, T, H, words)
    want = ops.rope_bwd(plain.clone(), cos, sin, T, E, 64, 0.125)
    got = ops.attention_bwd(qkv, out, d_out, lse, B, T, H, words, cos, sin, 0.125)
    np.testing.assert_array_equal(got.float().cpu().numpy(), want.float().cpu().numpy())
