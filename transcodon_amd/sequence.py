"""Sequence wrappers that turn nucleotide strings into space-separated token strings.

API of the reference's transcodon/sequence.py: every class exposes ``.seq`` (the token
string handed to the BatchConverter) and ``.tokens``.  One token per nucleotide; '*' in an
inference template becomes <mask>; the 5'UTR and CDS are joined by <sep>
(sequence.py:96-142,168-212).
"""

from typing import Iterable, List

CODON_TO_AA = {}
for _aa, _codons in {
    "F": "UUU UUC", "L": "UUA UUG CUU CUC CUA CUG", "I": "AUU AUC AUA", "M": "AUG",
    "V": "GUU GUC GUA GUG", "P": "CCU CCC CCA CCG", "T": "ACU ACC ACA ACG",
    "A": "GCU GCC GCA GCG", "Y": "UAU UAC", "#": "UAA UAG UGA", "H": "CAU CAC",
    "Q": "CAA CAG", "N": "AAU AAC", "K": "AAA AAG", "D": "GAU GAC", "E": "GAA GAG",
    "C": "UGU UGC", "W": "UGG", "R": "CGU CGC CGA CGG AGA AGG",
    "S": "AGU AGC UCU UCC UCA UCG", "G": "GGU GGC GGA GGG",
}.items():
    for _c in _codons.split():
        CODON_TO_AA[_c] = _aa
codon_to_amino_acid = CODON_TO_AA


def _symbols(s: str, mask_star: bool) -> List[str]:
    s = s.replace(" ", "")
    return [("<mask>" if (mask_star and ch == "*") else ch) for ch in s]


class Sequence:
    _seq = ""

    @property
    def seq(self) -> str:
        return self._seq

    @property
    def tokens(self) -> List[str]:
        return self._seq.split()

    def _set(self, parts: Iterable[str]):
        self._seq = " ".join(p.strip() for p in parts if p.strip() != "")


class CodonSequence(Sequence):
    """<cls> cds <eos>, T->U (training form; the UTR argument is ignored like upstream)."""

    def __init__(self, seq_utr_, seq_):
        self._set(["<cls>"] + _symbols(str(seq_).replace("T", "U"), False) + ["<eos>"])


class WithUtr_CodonSequence(Sequence):
    """<cls> utr <sep> cds <eos>, T->U."""

    def __init__(self, utr_seq_, cds_seq_):
        self._set(["<cls>"] + _symbols(str(utr_seq_).replace("T", "U"), False) + ["<sep>"]
                  + _symbols(str(cds_seq_).replace("T", "U"), False) + ["<eos>"])


class RNA2DSequence(Sequence):
    """<cls> dot-bracket symbols <eos>."""

    def __init__(self, seq_):
        self._set(["<cls>"] + _symbols(str(seq_), False) + ["<eos>"])


class CodonSequence_infer(Sequence):
    """<cls> template <eos>; '*' -> <mask>; no T->U (the template is already RNA)."""

    def __init__(self, seq_):
        self._set(["<cls>"] + _symbols(str(seq_), True) + ["<eos>"])


class WithUtr_CodonSequence_infer(Sequence):
    """<cls> utr <sep> template <eos>; T->U on both parts; '*' -> <mask>."""

    def __init__(self, utr_seq_, cds_seq_):
        self._set(["<cls>"] + _symbols(str(utr_seq_).replace("T", "U"), True) + ["<sep>"]
                  + _symbols(str(cds_seq_).replace("T", "U"), True) + ["<eos>"])


class AminoAcidSequence(Sequence):
    """<cls> (aa + codon) ... <eos> for a coding sequence."""

    def __init__(self, seq_):
        s = str(seq_).replace("T", "U").replace(" ", "")
        self._set(["<cls>"] + [CODON_TO_AA[s[i:i + 3]] + s[i:i + 3] for i in range(0, len(s), 3)]
                  + ["<eos>"])
