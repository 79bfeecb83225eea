"""Sequence-level evaluation metrics of the reference's evaluation/ scripts, vectorised.

codon recovery (evaluation/codon_recover_rate.py:125-149), GC content (gc.py:7-21), codon
frequency distribution CFD (cfd.py:9-45) and %MinMax (minmax.py:9-79) over codon-index arrays
instead of per-codon Python loops.  The CAI weights and the minimum free energy of the
reference come from external packages (CAI, ViennaRNA) and are not rebuilt here.
"""

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

_BASE = np.full(256, -1, np.int16)
for _i, _c in enumerate("ACGT"):
    _BASE[ord(_c)] = _i
    _BASE[ord(_c.lower())] = _i
_BASE[ord("U")] = _BASE[ord("u")] = 3


def codon_ids(dna: str) -> np.ndarray:
    """Codon indices (16 a + 4 b + c over A,C,G,T/U = 0..3) of the complete codons of ``dna``."""
    b = _BASE[np.frombuffer(dna.encode("ascii"), np.uint8)]
    n = len(b) // 3
    if (b[:3 * n] < 0).any():
        raise KeyError("sequence contains a letter outside ACGTU")
    b = b[:3 * n].reshape(n, 3).astype(np.int64)
    return 16 * b[:, 0] + 4 * b[:, 1] + b[:, 2]


def gc_content(dna: str) -> float:
    """Percentage of G and C letters (gc.py:19-21; upper-case letters only, like str.count)."""
    a = np.frombuffer(dna.encode("ascii"), np.uint8)
    return float(((a == ord("G")) | (a == ord("C"))).sum()) / len(dna) * 100


def codon_similarity(dna1: str, dna2: str) -> float:
    """Share of identical codons over the common prefix (codon_recover_rate.py:137-149)."""
    m = min(len(dna1), len(dna2))
    a = np.frombuffer(dna1[:m].encode("ascii"), np.uint8)
    b = np.frombuffer(dna2[:m].encode("ascii"), np.uint8)
    full = m // 3
    same = int((a[:3 * full].reshape(full, 3) == b[:3 * full].reshape(full, 3)).all(1).sum())
    if m % 3 and (a[3 * full:] == b[3 * full:]).all():
        same += 1                                          # the trailing partial codon is compared too
    return same / (m / 3)


class CodonUsage:
    """Per-codon lookup tables of a ``{amino: (codons, frequencies)}`` usage table."""

    def __init__(self, codon_frequencies: Dict[str, Tuple[Sequence[str], Sequence[float]]]):
        self.freq = np.full(64, np.nan)
        self.max = np.full(64, np.nan)
        self.min = np.full(64, np.nan)
        self.avg = np.full(64, np.nan)
        for _, (codons, freqs) in codon_frequencies.items():
            f = [float(x) for x in freqs]
            for c, x in zip(codons, f):
                i = int(codon_ids(c)[0])
                self.freq[i], self.max[i], self.min[i] = x, max(f), min(f)
                self.avg[i] = sum(f) / len(f)
        self.relative = self.freq / self.max

    def _ids(self, dna: str) -> np.ndarray:
        ids = codon_ids(dna)
        if np.isnan(self.freq[ids]).any():
            raise KeyError("codon missing from the usage table")
        return ids


def cfd(dna: str, usage: CodonUsage, threshold: float = 0.3) -> float:
    """Percentage of codons whose frequency relative to the most used synonymous codon is below
    ``threshold`` (cfd.py:27-45)."""
    ids = usage._ids(dna)
    return float((usage.relative[ids] < threshold).sum()) / (len(dna) / 3) * 100


def min_max_percentage(dna: str, usage: CodonUsage, window_size: int = 18) -> List[Optional[float]]:
    """%MinMax profile (minmax.py:31-79): one value per window of ``window_size`` codons, then
    ``window_size // 2`` trailing None entries."""
    ids = usage._ids(dna)
    n = len(ids) - window_size + 1
    out: List[Optional[float]] = []
    if n > 0:
        def window_mean(v):
            c = np.concatenate([[0.0], np.cumsum(v[ids])])
            return (c[window_size:] - c[:-window_size]) / window_size
        actual, mx, mn, avg = (window_mean(v) for v in (usage.freq, usage.max, usage.min, usage.avg))
        with np.errstate(divide="ignore", invalid="ignore"):
            pmax = (actual - avg) / (mx - avg) * 100
            pmin = (avg - actual) / (avg - mn) * 100
        out = [float(v) for v in np.where(pmax >= 0, pmax, -pmin)]
    return out + [None] * int(window_size / 2)


def device_metrics(bases, codon_off, ref_bases=None, usage: Optional[CodonUsage] = None, threshold: float = 0.3):
    """GC content, codon recovery and CFD of many decoded sequences at once, on the GPU, straight
    from tc_codon_decode's output: ``bases`` uint8 [3 * n_codons] (ASCII, CUDA tensor), ``codon_off``
    int32 [n_seq+1].  Integer counts come from one kernel (tc_codon_stats); the ratios formed from
    them equal gc_content / codon_similarity / cfd of the corresponding strings exactly.
    Returns {"gc": [n_seq], "recovery": [n_seq] or None, "cfd": [n_seq] or None} as numpy arrays."""
    import torch
    from . import _lib
    if not bases.is_cuda:
        raise _lib.HipLibraryError("device_metrics needs the bases on the GPU (use the string functions on the host)")
    off = codon_off.to(device=bases.device, dtype=torch.int32).contiguous()
    n_seq = off.numel() - 1
    flags = None
    if usage is not None:
        rel = np.nan_to_num(usage.relative, nan=np.inf)
        flags = torch.from_numpy((rel < threshold).astype(np.uint8)).to(bases.device)
    counts = torch.empty(n_seq, 4, dtype=torch.int32, device=bases.device)
    ref = None if ref_bases is None else ref_bases.contiguous()
    _lib.check(_lib.load().tc_codon_stats(bases.contiguous().data_ptr(), None if ref is None else ref.data_ptr(),
                                          off.data_ptr(), None if flags is None else flags.data_ptr(),
                                          counts.data_ptr(), n_seq, torch.cuda.current_stream().cuda_stream),
               "tc_codon_stats")
    c = counts.cpu().numpy().astype(np.float64)
    if c[:, 3].any():
        raise KeyError("sequence contains a letter outside ACGTU")
    n = np.diff(off.cpu().numpy()).astype(np.float64)
    return {"gc": c[:, 0] / (3 * n) * 100,
            "recovery": None if ref is None else c[:, 1] / n,
            "cfd": None if flags is None else c[:, 2] / n * 100}


def device_min_max(bases, codon_off, usage: CodonUsage, window_size: int = 18):
    """%MinMax profiles of many decoded sequences at once on the GPU (tc_codon_minmax): ``bases`` uint8
    [3 * n_codons] (ASCII, CUDA tensor), ``codon_off`` int32 [n_seq+1].  Returns one list per sequence in the
    reference's form (minmax.py:31-79): len - window_size + 1 floats, then window_size // 2 None entries; the
    floats equal get_min_max_percentage's bit for bit (same f64 operations in the same order)."""
    import torch
    from . import _lib
    if not bases.is_cuda:
        raise _lib.HipLibraryError("device_min_max needs the bases on the GPU (use min_max_percentage on the host)")
    off = codon_off.to(device=bases.device, dtype=torch.int32).contiguous()
    n_seq = off.numel() - 1
    off_h = off.cpu().numpy()
    n_codons = int(off_h[-1])
    tables = torch.from_numpy(np.stack([usage.freq, usage.max, usage.min, usage.avg]).astype(np.float64)).to(bases.device)
    out = torch.empty(max(n_codons, 1), dtype=torch.float64, device=bases.device)
    status = torch.zeros(1, dtype=torch.int32, device=bases.device)
    if n_codons:
        _lib.check(_lib.load().tc_codon_minmax(bases.contiguous().data_ptr(), off.data_ptr(), tables.data_ptr(),
                                               int(window_size), out.data_ptr(), status.data_ptr(), n_seq, n_codons,
                                               torch.cuda.current_stream().cuda_stream), "tc_codon_minmax")
    vals = out.cpu().numpy()
    if int(status.item()):
        raise KeyError("sequence contains a letter outside ACGTU or a codon missing from the usage table")
    res = []
    for s in range(n_seq):
        a, b = int(off_h[s]), int(off_h[s + 1])
        n = max(b - a - window_size + 1, 0)
        res.append([float(x) for x in vals[a:a + n]] + [None] * int(window_size / 2))
    return res
