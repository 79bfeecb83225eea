"""Multi-head self-attention with rotary embeddings on the HIP kernels.

Same constructor, parameter names and forward signature as the reference's
transcodon/multihead_attention.py:68-408 for the branch the model uses (separate
q/k/v/out projections with bias, q scaled by head_dim**-0.5, RoPE on q and k, key padding
mask, fp32 softmax, no dropout).  The three projections run as one GEMM whose epilogue
applies bias, scale and rotation; scores never reach memory (tc_attn_fwd).
"""

import math
from typing import Optional, Tuple

import torch
from torch import Tensor, nn

from . import _lib, ops
from .rotary import RotaryEmbedding


def _param_key(params):
    return (_lib.weights_epoch[0],) + tuple((p.data_ptr(), p._version, p.dtype, str(p.device)) for p in params)


class MultiheadAttention(nn.Module):
    def __init__(self, embed_dim, num_heads, kdim=None, vdim=None, dropout=0.0, bias=True,
                 add_bias_kv=False, add_zero_attn=False, self_attention=False,
                 encoder_decoder_attention=False, rope_embedding=False):
        super().__init__()
        if (kdim not in (None, embed_dim)) or (vdim not in (None, embed_dim)):
            raise NotImplementedError("only self-attention with kdim == vdim == embed_dim")
        if add_bias_kv or add_zero_attn or encoder_decoder_attention or not bias:
            raise NotImplementedError("bias_kv / zero_attn / encoder-decoder attention are not built")
        if not rope_embedding:
            raise NotImplementedError("only the rotary-embedding attention is built")
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == embed_dim, "embed_dim must be divisible by num_heads"
        if self.head_dim != 64:
            raise NotImplementedError("the attention kernel is built for head_dim 64")
        self.scaling = self.head_dim ** -0.5
        self.rope_embedding = True
        self.rope = RotaryEmbedding(dim=self.head_dim)
        self.k_proj = nn.Linear(embed_dim, embed_dim, bias=True)
        self.v_proj = nn.Linear(embed_dim, embed_dim, bias=True)
        self.q_proj = nn.Linear(embed_dim, embed_dim, bias=True)
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=True)
        self.compute_dtype = None
        self._pack_cache = {}
        self.reset_parameters()

    def reset_parameters(self):
        g = 1 / math.sqrt(2)
        for lin in (self.k_proj, self.v_proj, self.q_proj):
            nn.init.xavier_uniform_(lin.weight, gain=g)
        nn.init.xavier_uniform_(self.out_proj.weight)
        nn.init.constant_(self.out_proj.bias, 0.0)

    # -- packed device weights for the kernels --------------------------------------------
    def packed(self, dtype):
        params = [self.q_proj.weight, self.k_proj.weight, self.v_proj.weight, self.q_proj.bias,
                  self.k_proj.bias, self.v_proj.bias, self.out_proj.weight, self.out_proj.bias]
        key = (dtype, _param_key(params))
        if self._pack_cache.get("key") != key:
            with torch.no_grad():
                self._pack_cache = {
                    "key": key,
                    "w_qkv": torch.cat([p.detach() for p in params[0:3]]).to(dtype).contiguous(),
                    "b_qkv": torch.cat([p.detach() for p in params[3:6]]).float().contiguous(),
                    "w_o": self.out_proj.weight.detach().to(dtype).contiguous(),
                    "b_o": self.out_proj.bias.detach().float().contiguous(),
                }
        return self._pack_cache

    def _dtype(self, x):
        return self.compute_dtype or (torch.bfloat16 if self.q_proj.weight.dtype == torch.bfloat16
                                      else torch.float32)

    def forward(self, query, key: Optional[Tensor] = None, value: Optional[Tensor] = None,
                key_padding_mask: Optional[Tensor] = None, incremental_state=None,
                need_weights: bool = True, static_kv: bool = False,
                attn_mask: Optional[Tensor] = None, before_softmax: bool = False,
                need_head_weights: bool = False) -> Tuple[Tensor, Optional[Tensor]]:
        """query [T,B,E] (key/value must be the same tensor) -> (attn [T,B,E], weights).

        ``weights`` is None unless need_head_weights=True, in which case it holds the
        pre-softmax scores [H,B,T,T] in fp32 with -inf at padded keys, exactly what the
        reference returns there (multihead_attention.py:400-403).  The head-averaged map the
        reference computes for need_weights=True is discarded by its only caller
        (model.py:223) and is not produced."""
        if incremental_state is not None or static_kv or attn_mask is not None or before_softmax:
            raise NotImplementedError("incremental decoding / attn_mask / before_softmax are not built")
        if (key is not None and key is not query) or (value is not None and value is not query):
            raise NotImplementedError("only self-attention (key is value is query)")
        T, B, E = query.size()
        assert E == self.embed_dim
        dt = self._dtype(query)
        pk = self.packed(dt)
        x = query.transpose(0, 1).contiguous().view(B * T, E).to(dt)
        cos, sin = self.rope.tables(T, query.device)
        qkv = ops.gemm(x, pk["w_qkv"], pk["b_qkv"], _lib.EPI_QKV_ROPE, q_scale=self.scaling,
                       rope_cos=cos, rope_sin=sin, T=T, head_dim=self.head_dim)
        pad = words = None
        if key_padding_mask is not None:
            assert key_padding_mask.shape == (B, T)
            pad = key_padding_mask.to(torch.uint8).contiguous().view(-1)
            words = ops.key_mask_words(pad, B, T)
        a = ops.attention(qkv, B, T, self.num_heads, words)
        out = ops.gemm(a, pk["w_o"], pk["b_o"], _lib.EPI_BIAS)
        out = out.view(B, T, E).transpose(0, 1).to(query.dtype)
        weights = None
        if need_head_weights:
            weights = ops.attention_scores(qkv, B, T, self.num_heads, pad).transpose(1, 0)
        return out, weights
