"""CSV sequence dataset for pretraining, as integer token arrays.

Mirrors ``SequenceDataset`` of the reference's transcodon/dataset.py:41-165 as shipped: every row
gives ``<cls> cds <eos>`` (T read as U, one nucleotide per token), the species id of its
``species_name`` column and a structure label of all ``*`` (the 5'UTR and the dot-bracket labels are
hard-disabled there, :65-67, :80-83).  Where the reference builds token STRINGS and leaves the
integer conversion to four collate workers, rows are encoded to int32 arrays once, at load time.

The reference needs a ``species_name`` column (:145) while its bundled test.csv carries ``organism``;
either is accepted here.  ``use_utr=True`` re-enables the ``5utr_sequence`` column
(``<cls> utr <sep> cds <eos>``), ``use_struct=True`` the ``rna_2d`` dot-bracket column, right-aligned
over the sequence as dataset.py:78-96 does.
"""

import csv
import sys
from typing import List, Optional

import numpy as np

from .alphabet import ORGANISM2ID

_BASE = {"A": 5, "U": 6, "T": 6, "C": 7, "G": 8}
_STRUCT = {".": 0, "(": 1, ")": 2}                  # rna_2d_toks, alphabet.py:29-32; '*' -> -100
CLS, PAD, EOS, UNK, SEP = 0, 1, 2, 3, 4


def encode_nucleotides(seq: str) -> np.ndarray:
    """A/U(T)/C/G -> 5..8 (case-insensitive); any other letter raises KeyError, as the reference's
    alphabet lookup does."""
    return np.fromiter((_BASE[c] for c in seq.upper()), dtype=np.int32, count=len(seq))


class SequenceDataset:
    def __init__(self, csv_file: Optional[str] = None, use_utr: bool = False, use_struct: bool = False):
        self.tokens: List[np.ndarray] = []
        self.organism: List[int] = []
        self.struct_label: List[np.ndarray] = []
        if csv_file is None:
            return
        csv.field_size_limit(sys.maxsize)
        with open(csv_file, "r") as fh:
            for row in csv.DictReader(fh):
                name = row["species_name"] if "species_name" in row else row["organism"]
                utr = row.get("5utr_sequence", "") if use_utr else ""
                self.add(row["cds_sequence"], ORGANISM2ID[name], utr,
                         row["rna_2d"].split("|")[0] if use_struct and row.get("rna_2d") else "")

    def add(self, cds: str, organism: int, utr: str = "", dot_bracket: str = ""):
        body = encode_nucleotides(cds)
        if utr:
            body = np.concatenate([encode_nucleotides(utr), [SEP], body]).astype(np.int32)
        st = np.full(len(body), -100, np.int32)
        if dot_bracket:                                   # right-aligned over utr + sep + cds (dataset.py:89-93)
            lab = np.fromiter((_STRUCT.get(c, -100) for c in dot_bracket), dtype=np.int32, count=len(dot_bracket))
            st[len(st) - len(lab):] = lab[-len(st):]
        self.tokens.append(np.concatenate([[CLS], body, [EOS]]).astype(np.int32))
        # RNA2DSequence wraps the label string in <cls> ... <eos> as well: both map to -100
        self.struct_label.append(np.concatenate([[-100], st, [-100]]).astype(np.int32))
        self.organism.append(int(organism))

    def __len__(self) -> int:
        return len(self.tokens)

    def __getitem__(self, idx) -> dict:
        return {"sequence": self.tokens[idx], "organism": self.organism[idx], "struct_label": self.struct_label[idx]}

    @classmethod
    def synthetic(cls, n: int, seed: int = 0, min_codons: int = 30, max_codons: int = 400, n_species: int = 1438):
        """Random coding sequences (uniform codons, no structure labels): stands in for the 5.4 M-row
        pretraining table, which is not available offline."""
        rng = np.random.default_rng(seed)
        ds = cls()
        for _ in range(n):
            m = int(rng.integers(min_codons, max_codons + 1))
            ds.tokens.append(np.concatenate([[CLS], rng.integers(5, 9, 3 * m), [EOS]]).astype(np.int32))
            ds.struct_label.append(np.full(3 * m + 2, -100, np.int32))
            ds.organism.append(int(rng.integers(0, n_species)))
        return ds


def train_val_split(n: int, test_size: float = 0.003, seed: int = 42):
    """Index split in the proportions of data_module.py:42-45 (a seeded permutation; not sklearn's
    draw, so the membership of the two parts differs from the reference's while their sizes agree)."""
    perm = np.random.default_rng(seed).permutation(n)
    n_val = int(np.ceil(n * test_size)) if n * test_size > 0 else 0
    n_val = min(max(n_val, 0), max(n - 1, 0))
    return perm[n_val:], perm[:n_val]
