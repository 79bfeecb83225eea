"""The allocation-free attention backward (tc_attn_bwd_static) against tc_attn_bwd, through the C ABI:
bit-identical d_qkv on two calls in a row, the second with a smaller input (the built-in scratch keeps
the first call's contents), plus the size limit."""
import numpy as np
import pytest
import torch

from transcodon_amd import _lib, ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _case(B, T, H, dtype, seed, pad_tail=0):
    g = torch.Generator().manual_seed(seed)
    E = H * 64
    qkv = (torch.randn(B * T, 3 * E, generator=g) * 0.5).to(dtype).to(DEV)
    d_out = torch.randn(B * T, E, generator=g).to(dtype).to(DEV)
    words = None
    if pad_tail:
        pad = torch.zeros(B, T, dtype=torch.uint8)
        pad[-1, T - pad_tail:] = 1
        words = ops.key_mask_words(pad.view(-1).to(DEV), B, T)
    out, lse = ops.attention_train(qkv, B, T, H, words)
    return qkv, out, d_out, lse, words


def _run(static, qkv, out, d_out, lse, words, B, T, H):
    lib = _lib.load()
    E = H * 64
    d_qkv = torch.full((B * T, 3 * E), float("nan"), dtype=qkv.dtype, device=DEV)
    dt = _lib.TC_BF16 if qkv.dtype == torch.bfloat16 else _lib.TC_F32
    p = lambda t: None if t is None else t.data_ptr()
    s = torch.cuda.current_stream().cuda_stream
    if static:
        rc = lib.tc_attn_bwd_static(dt, p(qkv), 3 * E, p(words), p(out), p(d_out), p(lse), p(d_qkv), 3 * E, None, None,
                                    1.0, B, T, H, 64, s)
    else:
        delta = torch.empty(B, H, T, dtype=torch.float32, device=DEV)
        rc = lib.tc_attn_bwd(dt, p(qkv), 3 * E, p(words), p(out), p(d_out), p(lse), p(delta), p(d_qkv), 3 * E, None,
                             None, 1.0, B, T, H, 64, s)
    _lib.check(rc, "tc_attn_bwd_static" if static else "tc_attn_bwd")
    torch.cuda.synchronize()
    return d_qkv


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_static_scratch_attention_backward_equals_workspace_form(dtype):
    big = (3, 200, 2)
    small = (1, 70, 2)
    for (B, T, H), pad_tail, seed in ((big, 37, 1), (small, 0, 2)):       # second call: smaller input
        args = _case(B, T, H, dtype, seed, pad_tail)
        want = _run(False, *args, B, T, H)
        got = _run(True, *args, B, T, H)
        assert not torch.isnan(got.float()).any()
        assert torch.equal(got, want), (B, T, H)
    # and once more at the first size: the scratch now holds the small call's values
    args = _case(*big, dtype, 3, 5)
    assert torch.equal(_run(True, *args, *big), _run(False, *args, *big))


def test_static_scratch_rejects_sizes_beyond_its_limit():
    lib = _lib.load()
    x = torch.zeros(16, device=DEV)
    # B*H*T = 4 194 305 floats: one more than the built-in scratch holds; rejected before any launch
    rc = lib.tc_attn_bwd_static(_lib.TC_BF16, x.data_ptr(), 3 * 64, None, x.data_ptr(), x.data_ptr(), x.data_ptr(),
                                x.data_ptr(), 3 * 64, None, None, 1.0, 4194305, 1, 1, 64, None)
    assert rc == -1 and b"exceeds the built-in scratch" in lib.tc_last_error()
