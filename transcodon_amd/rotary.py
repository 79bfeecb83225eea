"""Rotary position embedding state and angle tables.

The reference takes its rotary arithmetic from rotary-embedding-torch==0.8.6
(call sites transcodon/multihead_attention.py:14,106,358-360): a non-trainable
``freqs`` parameter of head_dim/2 inverse frequencies, theta = 10000, that shows up in
every layer's state_dict as ``self_attn.rope.freqs``.  Here the rotation itself is fused
into the q/k/v projection kernel; this module only keeps the parameter (so checkpoints
load) and builds the cos/sin tables the kernel reads.
"""

import torch
from torch import nn


class RotaryEmbedding(nn.Module):
    def __init__(self, dim: int, theta: float = 10000.0):
        super().__init__()
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2).float() / dim))
        self.freqs = nn.Parameter(freqs, requires_grad=False)
        self._tables = {}

    def tables(self, length: int, device):
        """(cos, sin) f32 [length, dim/2] on ``device`` for positions 0..length-1.
        Angles are formed in fp32 (position * freq) on the host, as the library does."""
        f = self.freqs.detach().float().cpu()
        key = (length, str(device), f.numpy().tobytes())
        hit = self._tables.get(key)
        if hit is None:
            ang = torch.arange(length, dtype=torch.float32)[:, None] * f[None, :]
            hit = (ang.cos().contiguous().to(device), ang.sin().contiguous().to(device))
            self._tables = {key: hit}          # keep only the latest size
        return hit
