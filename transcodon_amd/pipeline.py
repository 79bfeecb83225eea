"""Masked-LM batch pipeline on the device.

The reference's collate function is a chain of four blocks over token STRINGS run in DataLoader
worker processes (transcodon/pipeline.py): DataCollator (:143-205, 25 % of the positions 1..n-2 drawn
without replacement, 80 % -> <mask>, 10 % kept, 10 % -> random A/U/C/G), DataTrimmer (:208-263,
random window of max_positions tokens for longer sequences, taken AFTER masking), DataPadder
(:266-319, right-pad to the batch maximum) and DataPreprocessor (:322-389, int32 inputs, int64
labels with -100 off-mask, structure labels with -100 off-mask).  Here the host only stacks the
integer rows of a batch; masking, cropping, padding and label construction are three kernels
(tc_mlm_mask, tc_crop_pad) on the padded int batch.  Random draws come from counter-based
generators keyed by (seed, step): same distributions as the reference's numpy global generator, not
the same draws.
"""

from typing import Dict, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import check
from .train import mlm_mask

MAX_MASK_LEN = 16384          # tc_mlm_mask keeps a row's selection flags in LDS


def _stream():
    return torch.cuda.current_stream().cuda_stream


def crop_pad(x, lens, start, T_out, fill):
    """tc_crop_pad: x int32 [B,T_in] -> [B,T_out]; returns (out, lens_out)."""
    B, T_in = x.shape
    out = torch.empty(B, T_out, dtype=torch.int32, device=x.device)
    lens_out = torch.empty(B, dtype=torch.int32, device=x.device)
    check(_lib.load().tc_crop_pad(x.data_ptr(), out.data_ptr(), None if start is None else start.data_ptr(),
                                  lens.data_ptr(), lens_out.data_ptr(), B, T_in, T_out, int(fill), _stream()),
          "tc_crop_pad")
    return out, lens_out


class DevicePipeline:
    """collate(items, step) -> {"input" int32 [B,T], "labels" int32 [B,T], "struct_label" int32 [B,T],
    "organism" int64 [B], "lens" int32 [B], "struct_present" bool}, all on ``device``; the keys and
    meanings of the reference's PipelineOutput (pipeline.py:363-364)."""

    def __init__(self, device, max_positions=2048, mask_proportion=0.25, mask_percent=0.8, leave_percent=0.1,
                 seed=0, pad_idx=1, mask_idx=9, coding_ids=(5, 6, 7, 8)):
        if not torch.device(device).type == "cuda":
            raise _lib.HipLibraryError("DevicePipeline runs on the GPU only (no CPU fallback)")
        self.device = torch.device(device)
        self.max_positions, self.seed = int(max_positions), int(seed)
        self.mask_proportion, self.mask_percent, self.leave_percent = mask_proportion, mask_percent, leave_percent
        self.pad_idx, self.mask_idx, self.coding_ids = pad_idx, mask_idx, tuple(coding_ids)

    def collate(self, items: Sequence[Dict], step: int) -> Dict[str, torch.Tensor]:
        rng = np.random.default_rng((self.seed, int(step), 1))
        rows, structs = [], []
        for it in items:
            t, s = np.asarray(it["sequence"], np.int32), np.asarray(it["struct_label"], np.int32)
            if len(t) > MAX_MASK_LEN:      # beyond the mask kernel's row limit: window it first (the reference masks the whole row)
                o = int(rng.integers(0, len(t) - MAX_MASK_LEN))
                t, s = t[o:o + MAX_MASK_LEN], s[o:o + MAX_MASK_LEN]
            rows.append(t)
            structs.append(s)
        B = len(rows)
        lens = np.array([len(r) for r in rows], np.int32)
        T_in = int(lens.max())
        tok = np.full((B, T_in), self.pad_idx, np.int32)
        st = np.full((B, T_in), -100, np.int32)
        for b, (r, s) in enumerate(zip(rows, structs)):
            tok[b, :len(r)] = r
            st[b, :len(s)] = s
        raw_struct = bool((st != -100).any())
        dev = self.device
        tok_d, lens_d = torch.from_numpy(tok).to(dev), torch.from_numpy(lens).to(dev)
        st_d = torch.from_numpy(st).to(dev) if raw_struct else None
        # 1. corruption of the whole rows (DataCollator)
        masked, labels, st_lab = mlm_mask(tok_d, lens_d, seed=(self.seed * 1000003 + int(step)) & (2 ** 63 - 1),
                                          struct_labels=st_d, mask_proportion=self.mask_proportion,
                                          mask_percent=self.mask_percent, leave_percent=self.leave_percent,
                                          mask_idx=self.mask_idx, coding_ids=self.coding_ids)
        # 2. + 3. random window of max_positions for longer rows (DataTrimmer: start uniform in
        # [0, n - max_positions), pipeline.py:257), right-padded to the longest kept row (DataPadder)
        T_out = min(T_in, self.max_positions)
        start = None
        if T_in > self.max_positions:
            over = np.maximum(lens - self.max_positions, 0)
            start_np = np.where(over > 0, (rng.random(B) * over).astype(np.int64), 0).astype(np.int32)
            start = torch.from_numpy(start_np).to(dev)
        if start is None and T_out == T_in:
            inp, lab, lens_out = masked, labels, lens_d
            st_out = st_lab
        else:
            inp, lens_out = crop_pad(masked, lens_d, start, T_out, self.pad_idx)
            lab, _ = crop_pad(labels, lens_d, start, T_out, -100)
            st_out = crop_pad(st_lab, lens_d, start, T_out, -100)[0] if st_lab is not None else None
        # pretraining.py:115 asks the labels that come OUT of the pipeline: a batch whose structure labels all
        # fell outside the drawn positions or outside the crop window has no structure term.  Only rows that
        # carry structure labels at all pay for this device read (the shipped dataset.py:80-83 has none).
        struct_present = bool((st_out != -100).any()) if st_out is not None else False
        if st_out is None:
            st_out = torch.full_like(lab, -100)
        organism = torch.tensor([int(it["organism"]) for it in items], dtype=torch.int64, device=dev)
        return {"input": inp, "labels": lab, "struct_label": st_out, "organism": organism, "lens": lens_out,
                "struct_present": struct_present, "crop_start": start}
