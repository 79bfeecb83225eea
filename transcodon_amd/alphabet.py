"""Nucleotide alphabet, batch converter and species table.

Mirrors the API of the reference's transcodon/alphabet.py (Alphabet :129-269,
BatchConverter :272-313, ORGANISM2ID :122-124) for the 'CodonModel' architecture: ten
symbols, one nucleotide per token, no automatic <cls>/<eos> (the sequence classes put
them into the string).
"""

import os
from typing import List, Sequence, Tuple

import torch

_SPECIALS_FRONT = ("<cls>", "<pad>", "<eos>", "<unk>", "<sep>")
_BASES = ("A", "U", "C", "G")
_SPECIALS_BACK = ("<mask>",)

rna_2d_toks = {".": 0, "(": 1, ")": 2, "*": -100}   # structure labels, alphabet.py:29-32


def _load_organisms():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "organisms.txt")
    with open(path, "r", encoding="utf-8") as fh:
        return {line.rstrip("\n"): i for i, line in enumerate(fh) if line.strip()}


# species name -> row of species_embedding (1438 entries; ids follow the released checkpoint)
ORGANISM2ID = _load_organisms()


class Alphabet:
    def __init__(self, standard_toks: Sequence[str], prepend_toks=_SPECIALS_FRONT,
                 append_toks=_SPECIALS_BACK, prepend_bos=False, append_eos=False, use_codons=True):
        self.standard_toks = list(standard_toks)
        self.prepend_toks = list(prepend_toks)
        self.append_toks = list(append_toks)
        self.prepend_bos, self.append_eos, self.use_codons = prepend_bos, append_eos, use_codons
        self.all_toks = self.prepend_toks + self.standard_toks + self.append_toks
        self.tok_to_idx = {t: i for i, t in enumerate(self.all_toks)}
        self.unk_idx = self.tok_to_idx["<unk>"]
        self.padding_idx = self.get_idx("<pad>")
        self.cls_idx = self.get_idx("<cls>")
        self.mask_idx = self.get_idx("<mask>")
        self.eos_idx = self.get_idx("<eos>")
        self.all_special_tokens = ["<eos>", "<unk>", "<pad>", "<cls>", "<mask>"]

    def __len__(self):
        return len(self.all_toks)

    def get_idx(self, tok):
        return self.tok_to_idx.get(tok, self.unk_idx)

    def get_tok(self, ind):
        return self.all_toks[ind]

    def to_dict(self):
        return dict(self.tok_to_idx)

    def get_batch_converter(self):
        return BatchConverter(self)

    @classmethod
    def from_architecture(cls, name: str) -> "Alphabet":
        if name != "CodonModel":
            raise ValueError("Unknown architecture selected")
        return cls(_BASES)

    def tokenize(self, text: str) -> List[str]:
        """Whitespace-separated symbols; symbols glued together (e.g. 'AUG<mask>') are
        split greedily on the alphabet's own tokens, anything else stays one token."""
        out: List[str] = []
        by_len = sorted(self.all_toks, key=len, reverse=True)
        for word in text.split():
            i = 0
            while i < len(word):
                for t in by_len:
                    if word.startswith(t, i):
                        out.append(t)
                        i += len(t)
                        break
                else:                      # unknown run up to the next known token
                    j = i + 1
                    while j < len(word) and not any(word.startswith(t, j) for t in by_len):
                        j += 1
                    out.append(word[i:j])
                    i = j
        return out

    def encode(self, text: str) -> List[int]:
        return [self.tok_to_idx[t] for t in self.tokenize(text)]   # KeyError on foreign symbols


class BatchConverter:
    """[(label, token string)] -> (labels, strings, int64 tokens right-padded with <pad>)."""

    def __init__(self, alphabet: Alphabet):
        self.alphabet = alphabet

    def __call__(self, raw_batch: Sequence[Tuple[str, str]]):
        a = self.alphabet
        labels = [lab for lab, _ in raw_batch]
        strs = [s for _, s in raw_batch]
        ids = [a.encode(s) for s in strs]
        bos, eos = int(a.prepend_bos), int(a.append_eos)
        width = max(len(x) for x in ids) + bos + eos
        tokens = torch.full((len(ids), width), a.padding_idx, dtype=torch.int64)
        for r, x in enumerate(ids):
            if bos:
                tokens[r, 0] = a.cls_idx
            tokens[r, bos:bos + len(x)] = torch.tensor(x, dtype=torch.int64)
            if eos:
                tokens[r, bos + len(x)] = a.eos_idx
        return labels, strs, tokens
