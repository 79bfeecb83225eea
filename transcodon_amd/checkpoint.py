"""Checkpoint I/O compatible with the reference's files.

The reference stores the weights of ``CodonModel`` (a wrapper whose ``self.model`` is the
ProteinBertModel), so every key carries a ``model.`` prefix: safetensors files are read by
infer.py:248-250, Lightning ``.ckpt`` files are written by the ModelCheckpoint callback
(pretraining.py:214-224, checkpointing.py:16-26) with the tensors under ``state_dict``.
Both are read here with loaders that execute nothing from the file.
"""

import os
from typing import Dict

import torch

PREFIX = "model."


def read_state_dict(path: str) -> Dict[str, torch.Tensor]:
    """Tensors of a .safetensors file or of a torch / Lightning checkpoint, keys as stored."""
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    obj = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(obj, dict) and "state_dict" in obj and isinstance(obj["state_dict"], dict):
        obj = obj["state_dict"]
    if not isinstance(obj, dict) or not all(isinstance(v, torch.Tensor) for v in obj.values()):
        raise ValueError(f"{path}: no tensor state dict found")
    return obj


def strip_prefix(sd: Dict[str, torch.Tensor], prefix: str = PREFIX) -> Dict[str, torch.Tensor]:
    """Drop the wrapper prefix when every key has it; otherwise return the dict unchanged."""
    if sd and all(k.startswith(prefix) for k in sd):
        return {k[len(prefix):]: v for k, v in sd.items()}
    return sd


def load_into(model, path: str, strict: bool = True):
    """Load a reference checkpoint (with or without the ``model.`` prefix) into a ProteinBertModel."""
    sd = strip_prefix(read_state_dict(path))
    if "lm_head.weight" not in sd and "embed_tokens.weight" in sd:
        sd = dict(sd)
        sd["lm_head.weight"] = sd["embed_tokens.weight"]      # tied (model.py:164-168); some writers drop aliases
    res = model.load_state_dict(sd, strict=strict)
    store = getattr(model, "_flat_store", None)
    if store is not None:
        store.mark_dirty()          # the copy went through the parameters, not through the flat buffer
    return res


def save(model, path: str, prefix: str = PREFIX, step: int = None):
    """Write the model the way the reference's tools read it back: safetensors, with every key of
    the reference's state_dict (infer.py:249-250 loads it with strict=True, so the tied
    ``lm_head.weight`` is written as a copy of its own), or a Lightning-style
    ``{"state_dict": ..., "global_step": ...}`` torch checkpoint for any other extension."""
    sd = {prefix + k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()}
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    if path.endswith(".safetensors"):
        from safetensors.torch import save_file
        save_file({k: v.clone() for k, v in sd.items()}, path)      # clones: no shared storage
    else:
        torch.save({"state_dict": sd, "global_step": step}, path)
