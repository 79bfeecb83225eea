"""Encoder building blocks on the HIP kernels.

Same class names, constructor arguments, parameter names and forward signatures as the
live classes of the reference's transcodon/modules.py (gelu :17-24, ESM1bLayerNorm
:27-41, TransformerLayer :44-105, RobertaLMHead :144-160, RNA_2D_RobertaLMHead :162-178).
Modules whose parameters exist in checkpoints but which the reference's forward never
calls (RNA_2D_Structure_Connector :201-245) are kept as parameter holders only.
"""

import torch
from torch import nn

from . import _lib, ops
from .multihead_attention import MultiheadAttention, _param_key


def gelu(x):
    """Exact erf GELU.  On the hot path it is fused into the fc1 / head GEMM epilogues;
    this standalone form is host-side convenience only."""
    return x * 0.5 * (1.0 + torch.erf(x / 1.4142135623730951))


class ESM1bLayerNorm(nn.LayerNorm):
    """nn.LayerNorm(E, eps=1e-5) parameters; forward on GPU tensors runs tc_layernorm_fwd."""

    def forward(self, x):
        if not x.is_cuda:
            raise _lib.HipLibraryError("ESM1bLayerNorm needs a GPU tensor (no CPU fallback)")
        shp = x.shape
        y = ops.layernorm(x.reshape(-1, shp[-1]).float().contiguous(), self.weight.detach().float(),
                          self.bias.detach().float(), self.eps)
        return y.view(shp).to(x.dtype)


class TransformerLayer(nn.Module):
    """Pre-LN block: x + MHA(LN(x)); x + fc2(gelu(fc1(LN(x))))."""

    def __init__(self, embed_dim, ffn_embed_dim, attention_heads, attention_dropout=0.0,
                 add_bias_kv=True, use_esm1b_layer_norm=False, rope_embedding=False):
        super().__init__()
        self.embed_dim, self.ffn_embed_dim = embed_dim, ffn_embed_dim
        self.attention_heads, self.attention_dropout = attention_heads, attention_dropout
        self.rope_embedding = rope_embedding
        self.self_attn = MultiheadAttention(embed_dim, attention_heads, add_bias_kv=add_bias_kv,
                                            add_zero_attn=False, dropout=attention_dropout,
                                            rope_embedding=rope_embedding)
        self.self_attn_layer_norm = ESM1bLayerNorm(embed_dim)
        self.fc1 = nn.Linear(embed_dim, ffn_embed_dim)
        self.fc2 = nn.Linear(ffn_embed_dim, embed_dim)
        self.final_layer_norm = ESM1bLayerNorm(embed_dim)
        self.compute_dtype = None
        self._pack_cache = {}

    def packed(self, dtype):
        params = [self.self_attn_layer_norm.weight, self.self_attn_layer_norm.bias,
                  self.final_layer_norm.weight, self.final_layer_norm.bias,
                  self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias]
        key = (dtype, _param_key(params))
        if self._pack_cache.get("key") != key:
            with torch.no_grad():
                f = lambda p: p.detach().float().contiguous()
                self._pack_cache = {
                    "key": key, "ln1_g": f(params[0]), "ln1_b": f(params[1]),
                    "ln2_g": f(params[2]), "ln2_b": f(params[3]),
                    "w_fc1": self.fc1.weight.detach().to(dtype).contiguous(), "b_fc1": f(params[5]),
                    "w_fc2": self.fc2.weight.detach().to(dtype).contiguous(), "b_fc2": f(params[7]),
                }
        out = dict(self._pack_cache)
        out.update(self.self_attn.packed(dtype))
        return out

    def _dtype(self):
        return self.compute_dtype or (torch.bfloat16 if self.fc1.weight.dtype == torch.bfloat16
                                      else torch.float32)

    def forward(self, x, self_attn_mask=None, self_attn_padding_mask=None, need_head_weights=False):
        """x [T,B,E] -> (x [T,B,E], attn); attn as MultiheadAttention.forward."""
        if self_attn_mask is not None:
            raise NotImplementedError("self_attn_mask is not built")
        if not x.is_cuda:
            raise _lib.HipLibraryError("TransformerLayer needs a GPU tensor (no CPU fallback)")
        T, B, E = x.shape
        dt = self._dtype()
        pk = self.packed(dt)
        res = x.transpose(0, 1).contiguous().view(B * T, E).float()
        h = ops.layernorm(res, pk["ln1_g"], pk["ln1_b"], self.self_attn_layer_norm.eps, dt)
        cos, sin = self.self_attn.rope.tables(T, x.device)
        qkv = ops.gemm(h, pk["w_qkv"], pk["b_qkv"], _lib.EPI_QKV_ROPE, q_scale=self.self_attn.scaling,
                       rope_cos=cos, rope_sin=sin, T=T, head_dim=64)
        pad = words = None
        if self_attn_padding_mask is not None:
            pad = self_attn_padding_mask.to(torch.uint8).contiguous().view(-1)
            words = ops.key_mask_words(pad, B, T)
        a = ops.attention(qkv, B, T, self.attention_heads, words)
        res = ops.gemm(a, pk["w_o"], pk["b_o"], _lib.EPI_BIAS_RESIDUAL, residual=res, out=res)
        h = ops.layernorm(res, pk["ln2_g"], pk["ln2_b"], self.final_layer_norm.eps, dt)
        mid = ops.gemm(h, pk["w_fc1"], pk["b_fc1"], _lib.EPI_BIAS_GELU)
        res = ops.gemm(mid, pk["w_fc2"], pk["b_fc2"], _lib.EPI_BIAS_RESIDUAL, residual=res, out=res)
        attn = None
        if need_head_weights:
            attn = ops.attention_scores(qkv, B, T, self.attention_heads, pad).transpose(1, 0)
        return res.view(B, T, E).transpose(0, 1).to(x.dtype), attn


class RobertaLMHead(nn.Module):
    """dense -> gelu -> LayerNorm -> tied projection to the vocabulary + bias."""

    def __init__(self, embed_dim, output_dim, weight):
        super().__init__()
        self.dense = nn.Linear(embed_dim, embed_dim)
        self.layer_norm = ESM1bLayerNorm(embed_dim)
        self.weight = weight
        self.bias = nn.Parameter(torch.zeros(output_dim))
        self.compute_dtype = None

    def forward(self, features):
        if not features.is_cuda:
            raise _lib.HipLibraryError("RobertaLMHead needs a GPU tensor (no CPU fallback)")
        dt = self.compute_dtype or (torch.bfloat16 if self.dense.weight.dtype == torch.bfloat16
                                    else torch.float32)
        shp = features.shape
        x = features.reshape(-1, shp[-1]).to(dt).contiguous()
        g = ops.gemm(x, self.dense.weight.detach().to(dt).contiguous(), self.dense.bias.detach().float(),
                     _lib.EPI_BIAS_GELU)
        out = ops.head_logits(g, 0, shp[-1], self.layer_norm.weight.detach().float(),
                              self.layer_norm.bias.detach().float(),
                              self.weight.detach().float().contiguous(), self.bias.detach().float(),
                              self.layer_norm.eps)
        return out.view(*shp[:-1], -1)


class RNA_2D_RobertaLMHead(nn.Module):
    """dense -> gelu -> LayerNorm -> Linear(E, output_dim) + extra bias."""

    def __init__(self, embed_dim, output_dim):
        super().__init__()
        self.dense = nn.Linear(embed_dim, embed_dim)
        self.layer_norm = ESM1bLayerNorm(embed_dim)
        self.out_proj = nn.Linear(embed_dim, output_dim)
        self.bias = nn.Parameter(torch.zeros(output_dim))
        self.compute_dtype = None

    def forward(self, features):
        if not features.is_cuda:
            raise _lib.HipLibraryError("RNA_2D_RobertaLMHead needs a GPU tensor (no CPU fallback)")
        dt = self.compute_dtype or (torch.bfloat16 if self.dense.weight.dtype == torch.bfloat16
                                    else torch.float32)
        shp = features.shape
        x = features.reshape(-1, shp[-1]).to(dt).contiguous()
        g = ops.gemm(x, self.dense.weight.detach().to(dt).contiguous(), self.dense.bias.detach().float(),
                     _lib.EPI_BIAS_GELU)
        bias = (self.out_proj.bias.detach().float() + self.bias.detach().float()).contiguous()
        out = ops.head_logits(g, 0, shp[-1], self.layer_norm.weight.detach().float(),
                              self.layer_norm.bias.detach().float(),
                              self.out_proj.weight.detach().float().contiguous(), bias,
                              self.layer_norm.eps)
        return out.view(*shp[:-1], -1)


class RelativePositionEncoder(nn.Module):
    """Parameter holder (rel_pos_emb [max_rel_pos+1, 1]); never called by the model."""

    def __init__(self, max_rel_pos=20):
        super().__init__()
        self.max_rel_pos = max_rel_pos
        self.rel_pos_emb = nn.Embedding(max_rel_pos + 1, 1)


class RNA_2D_Structure_Connector(nn.Module):
    """Parameter holder for checkpoint compatibility; the model's forward never uses it."""

    def __init__(self, embed_dim, max_rel_pos=20):
        super().__init__()
        self.embed_dim, self.max_rel_pos = embed_dim, max_rel_pos
        self.shared_proj = nn.Linear(embed_dim, embed_dim)
        self.rel_pos_encoder = RelativePositionEncoder(max_rel_pos)
        self.null_bias_layer = nn.Linear(embed_dim, 1)
