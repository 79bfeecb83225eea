"""Seeded weights for the oracle and the parity tests (TEST INFRASTRUCTURE ONLY).

``state_dict_spec`` restates the parameter names and shapes of the reference's
``ProteinBertModel.state_dict()`` (transcodon/model.py:102-173,
transcodon/modules.py:65-81,147-170,186,207-212,
transcodon/multihead_attention.py:106,115-119): 238 tensors at 12 layers,
including parameters that ``forward`` never touches.

``seeded_state_dict`` fills them from a numpy generator keyed by (seed, name),
so both sides of a parity test can rebuild the same 360 MB full-size weights
without committing them.  Distribution follows ``init_weights``
(pretraining.py:46-52): Linear/Embedding weights N(0, 0.02), Linear biases 0,
LayerNorm (1, 0), free-standing biases 0.  ``perturb=True`` additionally draws
biases and LayerNorm affine terms at random so that parity tests exercise them.
"""

import zlib
from typing import Dict, List, Tuple

import numpy as np
import torch

from .transcodon_ref import Config, rope_freqs


def state_dict_spec(cfg: Config) -> List[Tuple[str, Tuple[int, ...], str]]:
    """[(name, shape, kind)] with kind in {'w','b','ln_w','ln_b','freqs'}."""
    E, F, V, S = cfg.embed_dim, cfg.ffn_embed_dim, cfg.alphabet_size, cfg.struct_classes
    spec = []

    def lin(p, o, i):
        spec.append((p + ".weight", (o, i), "w"))
        spec.append((p + ".bias", (o,), "b"))

    def ln(p):
        spec.append((p + ".weight", (E,), "ln_w"))
        spec.append((p + ".bias", (E,), "ln_b"))

    spec.append(("embed_tokens.weight", (V, E), "w"))
    spec.append(("species_embedding.weight", (cfg.species_vocab_size, E), "w"))
    lin("struct_linear1", E, E)
    lin("struct_linear2", E, E)
    lin("struct_linear3", E, E)
    lin("connector.shared_proj", E, E)
    spec.append(("connector.rel_pos_encoder.rel_pos_emb.weight", (21, 1), "w"))
    lin("connector.null_bias_layer", 1, E)
    ln("norm")
    lin("existence_head.0", 128, E)
    lin("existence_head.2", 1, 128)
    for i in range(cfg.num_layers):
        p = f"layers.{i}."
        spec.append((p + "self_attn.rope.freqs", (cfg.head_dim // 2,), "freqs"))
        lin(p + "self_attn.k_proj", E, E)
        lin(p + "self_attn.v_proj", E, E)
        lin(p + "self_attn.q_proj", E, E)
        lin(p + "self_attn.out_proj", E, E)
        ln(p + "self_attn_layer_norm")
        lin(p + "fc1", F, E)
        lin(p + "fc2", E, F)
        ln(p + "final_layer_norm")
    ln("emb_layer_norm_after")
    spec.append(("lm_head.weight", (V, E), "tied"))
    spec.append(("lm_head.bias", (V,), "b"))
    lin("lm_head.dense", E, E)
    ln("lm_head.layer_norm")
    spec.append(("structure_linear.bias", (S,), "b"))
    lin("structure_linear.dense", E, E)
    ln("structure_linear.layer_norm")
    lin("structure_linear.out_proj", S, E)
    return spec


def seeded_state_dict(cfg: Config, seed: int, perturb: bool = False,
                      weight_std: float = 0.02) -> Dict[str, torch.Tensor]:
    sd: Dict[str, torch.Tensor] = {}
    for name, shape, kind in state_dict_spec(cfg):
        if kind == "tied":
            sd[name] = sd["embed_tokens.weight"]
            continue
        if kind == "freqs":
            sd[name] = rope_freqs(cfg.head_dim)
            continue
        rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
        if kind == "w":
            a = rng.standard_normal(shape, dtype=np.float32) * np.float32(weight_std)
        elif kind == "b":
            a = (rng.standard_normal(shape, dtype=np.float32) * np.float32(0.02)
                 if perturb else np.zeros(shape, np.float32))
        elif kind == "ln_w":
            a = (1.0 + 0.1 * rng.standard_normal(shape, dtype=np.float32)
                 if perturb else np.ones(shape, np.float32))
        else:  # ln_b
            a = (0.05 * rng.standard_normal(shape, dtype=np.float32)
                 if perturb else np.zeros(shape, np.float32))
        sd[name] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return sd


def trained_like_state_dict(cfg: Config, seed: int, common_offset: float = 0.12) -> Dict[str, torch.Tensor]:
    """Seeded weights with the features of a TRAINED checkpoint that N(0, 0.02) initialisation lacks
    (the released checkpoint is not available offline), so that parity is probed where bf16 kernels
    are stressed: LayerNorm gains ~ lognormal(0, 0.5) and shifts ~ N(0, 0.5); q/k projections six
    times larger (attention scores of tens of nats, peaked softmax rows); token / species embeddings
    of O(0.2); fc1 four times larger (pre-GELU values spanning +-10); v / out-proj / fc2 twice
    larger; and six outlier channels whose out-proj and fc2 biases keep one sign per channel, so the
    residual stream reaches |x| of 50-200 there, on top of a common offset (``common_offset`` per
    out-proj / fc2 bias element) that gives every row a non-zero mean: 0.12 puts |mean| / std of the
    LayerNorm inputs at 0.13, 0.6 at about 0.6.  Same names, shapes and generator keying as
    seeded_state_dict."""
    sd = seeded_state_dict(cfg, seed, perturb=True)
    E = cfg.embed_dim
    pick = np.random.default_rng([seed, 0x7A11ED])
    outliers = pick.choice(E, size=6, replace=False)
    signs = np.array([1, 1, -1, 1, 1, -1], np.float32)
    for name, shape, kind in state_dict_spec(cfg):
        if kind in ("tied", "freqs"):
            continue
        rng = np.random.default_rng([seed, zlib.crc32(name.encode()), 1])
        a = sd[name].numpy().copy()
        if kind == "ln_w":
            a = np.exp(0.5 * rng.standard_normal(shape, dtype=np.float32))
        elif kind == "ln_b":
            a = 0.5 * rng.standard_normal(shape, dtype=np.float32)
        elif name in ("embed_tokens.weight", "species_embedding.weight"):
            a *= 10.0
        elif name.endswith(("q_proj.weight", "k_proj.weight")):
            a *= 6.0
        elif name.endswith(("q_proj.bias", "k_proj.bias")):
            a = 0.3 * rng.standard_normal(shape, dtype=np.float32)
        elif name.endswith(("v_proj.weight", "out_proj.weight", "fc2.weight")) and name.startswith("layers."):
            a *= 2.0
        elif name.endswith("fc1.weight"):
            a *= 4.0
        elif name.endswith("fc1.bias"):
            a = 0.5 * rng.standard_normal(shape, dtype=np.float32)
        elif name.startswith("layers.") and name.endswith(("out_proj.bias", "fc2.bias")):
            a = 0.1 * rng.standard_normal(shape, dtype=np.float32) + np.float32(common_offset)
            a[outliers] = signs * rng.uniform(3.0, 8.0, size=6).astype(np.float32)
        sd[name] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    sd["lm_head.weight"] = sd["embed_tokens.weight"]
    return sd
