"""Model hyper-parameter namespace and weight initialisation."""

import argparse

import torch
from torch import nn


def default_args(**overrides) -> argparse.Namespace:
    """Argument namespace with the defaults of ProteinBertModel.add_args
    (reference transcodon/model.py:32-84) plus max_positions."""
    ns = argparse.Namespace(
        num_layers=12, embed_dim=768, attention_dropout=0.0, logit_bias=False, rope_embedding=True,
        ffn_embed_dim=3072, attention_heads=12, species_embedding=True, struct_label=True,
        species_vocab_size=1438, max_positions=2048)
    for k, v in overrides.items():
        setattr(ns, k, v)
    return ns


def init_weights(model: nn.Module, seed=None) -> nn.Module:
    """Linear / Embedding weights ~ N(0, 0.02), Linear biases 0, as the training module of
    the reference does after construction (pretraining.py:46-52)."""
    gen = None
    if seed is not None:
        gen = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, (nn.Linear, nn.Embedding)):
            with torch.no_grad():
                m.weight.copy_(torch.randn(m.weight.shape, generator=gen) * 0.02)
        if isinstance(m, nn.Linear):
            with torch.no_grad():
                m.bias.zero_()
    return model
