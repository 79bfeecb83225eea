"""Codon optimisation driver: tokenise -> forward -> greedy synonymous-codon decode.

``embed_sequence`` keeps the call contract of the reference's
transcodon/infer_utr.py:17-159; ``AA_tokenize`` / ``back_translation_ok`` those of
infer.py:76-88,113-129.  The decode itself (infer_utr.py:98-133, a triple Python loop with
one softmax launch and one device->host copy per base) is a single kernel here
(tc_codon_decode).  ``optimize_codons`` is the batched form: many proteins per forward,
ragged CDS offsets, one decode launch.
"""

from typing import List, Optional, Sequence, Union

import torch

from . import ops
from .alphabet import ORGANISM2ID, Alphabet
from .sequence import CODON_TO_AA, CodonSequence_infer, WithUtr_CodonSequence_infer

_ALPHABET = Alphabet.from_architecture("CodonModel")
_CONVERTER = _ALPHABET.get_batch_converter()

# first synonymous codon of every amino acid in the decode table's order
_FIRST_CODON = {"F": "UUU", "I": "AUU", "M": "AUG", "V": "GUU", "P": "CCU", "T": "ACU", "A": "GCU",
                "Y": "UAU", "_": "UAA", "H": "CAU", "Q": "CAA", "N": "AAU", "K": "AAA", "D": "GAU",
                "E": "GAA", "C": "UGU", "W": "UGG", "G": "GGU"}


def AA_tokenize(seq: str) -> str:
    """CDS template of a protein whose last character stands for the stop codon: the two
    bases shared by an amino acid's codons are given, the rest is '*' (masked)."""
    parts = []
    for ch in seq[:-1]:
        if ch == "L":
            parts.append("*U*")
        elif ch == "R":
            parts.append("*G*")
        elif ch == "S":
            parts.append("***")
        else:
            parts.append(_FIRST_CODON[ch][:2] + "*")      # KeyError for non-standard letters
    parts.append("U**")
    return "".join(parts)


def back_translation_ok(rna: str, aa_seq: str) -> bool:
    res = "".join(CODON_TO_AA[rna[i:i + 3]] for i in range(0, len(rna), 3))
    return res[:-1] == aa_seq[:len(res)][:-1]


def tokenize(seq) -> torch.Tensor:
    _, _, tokens = _CONVERTER([("", seq.seq)])
    return tokens


def _species_id(species) -> int:
    if isinstance(species, str):
        return ORGANISM2ID[species]
    if isinstance(species, int):
        return species
    raise ValueError("species_type_ids must be str or int")


def _model_device(model):
    return next(model.parameters()).device


class _HostCheckedIds:
    """Inside the batched drivers the token ids come from this module's tokeniser and the species ids
    are checked here on the host, so the per-forward device-side range check (one host
    synchronisation per forward) is switched off for the duration."""

    def __init__(self, model, species):
        self.model = model
        n = getattr(getattr(model, "args", None), "species_vocab_size", None)
        if n is not None and any(not (0 <= int(s) < n) for s in species):
            raise IndexError("species id out of range")

    def __enter__(self):
        self.prev = getattr(self.model, "check_indices", None)
        if self.prev is not None:
            self.model.check_indices = False

    def __exit__(self, *exc):
        if self.prev is not None:
            self.model.check_indices = self.prev


def embed_sequence(model, utr_seq, sequence: Union[str, WithUtr_CodonSequence_infer],
                   species_type_ids: Union[int, str] = 0, AA_seq: str = "", average: bool = True):
    """Returns (representations, logits sliced to the CDS [1, 3m(+), vocab], predicted RNA)."""
    if isinstance(sequence, str):
        seq = CodonSequence_infer(sequence) if utr_seq == "false" else \
            WithUtr_CodonSequence_infer(utr_seq, sequence)
    elif isinstance(sequence, (WithUtr_CodonSequence_infer, CodonSequence_infer)):
        seq = sequence
    else:
        raise ValueError("Input sequence must be string or CodonSequence.")
    dev = _model_device(model)
    species = torch.tensor([_species_id(species_type_ids)], device=dev)
    tokens = tokenize(seq).to(dev)
    with torch.no_grad():
        output = model(tokens, species)
    logits = output["logits"]
    start = 1 + (0 if utr_seq == "false" else len(utr_seq) + 1)
    cds_logits = logits[:, start:-1, :]
    T = tokens.shape[1]
    m = len(AA_seq)
    if 3 * m > T - 1 - start:
        raise IndexError("AA_seq is longer than the CDS part of the sequence")
    aa = torch.tensor(list(ops.encode_amino_acids(AA_seq)), dtype=torch.uint8, device=dev)
    off = torch.tensor([0, m], dtype=torch.int32, device=dev)
    row0 = torch.tensor([start], dtype=torch.int32, device=dev)
    bases = ops.codon_decode(logits.view(T, -1), aa, off, row0)
    rna = bytes(bases.cpu().numpy()).decode("ascii")
    return output["representations"], cds_logits, rna


def extract_embeddings(model, cds_seqs: Sequence[str], organisms: Sequence[Union[int, str]],
                       max_batch_tokens: int = 1 << 17, max_positions: int = 6146,
                       pooled: bool = True, packed: bool = True):
    """Embedding extraction (reference transcodon/infer_embedding.py: representations[12] of
    `<cls> cds <eos>`, one sequence per call) for many coding sequences at once.

    Sequences (DNA or RNA letters, T is read as U) go through the model in batches of at most
    ``max_batch_tokens`` tokens.  packed=True (default): the sequences of a batch are laid one
    after the other without padding (ProteinBertModel.forward_packed), so the work is the sum of
    the lengths; packed=False: sorted by length into right-padded batches with key-padding-aware
    attention.  Returns a [N, E] float tensor on the CPU with the mean of the final-layer states
    over the CDS positions (the pooling hinted at in infer_embedding.py:145), or, with
    pooled=False, a list of [len_i, E] tensors of the CDS positions."""
    from .sequence import CodonSequence
    dev = _model_device(model)
    L = model.num_layers
    enc = [_ALPHABET.encode(CodonSequence("", s).seq)[:max_positions] for s in cds_seqs]
    species = [_species_id(o) for o in organisms]
    out = [None] * len(enc)
    with _HostCheckedIds(model, species):
        return _extract(model, enc, species, out, dev, L, max_batch_tokens, pooled, packed)


def _extract(model, enc, species, out, dev, L, max_batch_tokens, pooled, packed):
    if packed:
        import numpy as np
        pos = 0
        while pos < len(enc):
            chunk, ntok = [], 0
            while pos < len(enc) and (not chunk or ntok + len(enc[pos]) <= max_batch_tokens):
                chunk.append(pos)
                ntok += len(enc[pos])
                pos += 1
            cu = np.concatenate([[0], np.cumsum([len(enc[k]) for k in chunk])])
            flat = torch.from_numpy(np.concatenate([np.asarray(enc[k], np.int32) for k in chunk]))
            with torch.no_grad():
                rep = model.forward_packed(flat.to(dev), [species[k] for k in chunk], cu,
                                           want_logits=False)["representations"][L]
            lens = np.maximum(np.diff(cu) - 2, 0)
            if pooled:
                p = ops.mean_pool_packed(rep, torch.from_numpy(cu[:-1] + 1).to(dev), torch.from_numpy(lens).to(dev)).cpu()
                for r, k in enumerate(chunk):
                    out[k] = p[r]
            else:
                rep = rep.cpu()
                for r, k in enumerate(chunk):
                    out[k] = rep[int(cu[r]) + 1:int(cu[r]) + 1 + int(lens[r])].clone()
        return torch.stack(out) if pooled else out
    order = sorted(range(len(enc)), key=lambda k: len(enc[k]))
    pos = 0
    while pos < len(order):
        width, chunk = 0, []
        while pos < len(order):
            w = max(width, len(enc[order[pos]]))
            if chunk and w * (len(chunk) + 1) > max_batch_tokens:
                break
            chunk.append(order[pos])
            width = w
            pos += 1
        tok = torch.full((len(chunk), width), _ALPHABET.padding_idx, dtype=torch.int32)
        for r, k in enumerate(chunk):
            tok[r, :len(enc[k])] = torch.tensor(enc[k], dtype=torch.int32)
        with torch.no_grad():
            rep = model(tok.to(dev), torch.tensor([species[k] for k in chunk], dtype=torch.int32, device=dev),
                        repr_layers=[L])["representations"][L]
        lens = torch.tensor([max(len(enc[k]) - 2, 0) for k in chunk], dtype=torch.int32, device=dev)
        if pooled:
            p = ops.mean_pool(rep, torch.ones_like(lens), lens).cpu()
            for r, k in enumerate(chunk):
                out[k] = p[r]
        else:
            rep = rep.cpu()
            for r, k in enumerate(chunk):
                out[k] = rep[r, 1:1 + int(lens[r])].clone()
    return torch.stack(out) if pooled else out


def optimize_codons(model, proteins: Sequence[str], organisms: Sequence[Union[int, str]],
                    utrs: Optional[Sequence[str]] = None, max_batch_tokens: int = 1 << 17,
                    max_aa: int = 2048, packed: bool = True) -> List[str]:
    """Batched codon optimisation: RNA string per protein (stop codon included).

    Proteins are given without the stop marker; like infer.py:262-269 a '_' is appended and
    templates longer than 3*max_aa bases are truncated.  Sequences are sorted by length and run in
    batches of at most ``max_batch_tokens`` tokens: packed=True lays a batch's sequences one after
    the other without padding rows (ProteinBertModel.forward_packed), packed=False right-pads them
    to the longest of the batch."""
    import numpy as np
    dev = _model_device(model)
    items = []
    for i, prot in enumerate(proteins):
        aa = prot + "_"
        tmpl = AA_tokenize(aa)
        if len(tmpl) > 3 * max_aa:
            aa, tmpl = aa[:max_aa], tmpl[:3 * max_aa]
        utr = None if utrs is None else utrs[i]
        seq = CodonSequence_infer(tmpl) if utr is None else WithUtr_CodonSequence_infer(utr, tmpl)
        ids = _ALPHABET.encode(seq.seq)
        start = 1 + (0 if utr is None else len(utr.replace(" ", "")) + 1)
        items.append((i, ids, start, aa, _species_id(organisms[i])))
    with _HostCheckedIds(model, [it[4] for it in items]):
        return _optimize(model, items, dev, max_batch_tokens, packed)


def _optimize(model, items, dev, max_batch_tokens, packed):
    import numpy as np
    order = sorted(range(len(items)), key=lambda k: len(items[k][1]))
    out: List[Optional[str]] = [None] * len(items)
    pos = 0
    while pos < len(order):
        width, ntok = 0, 0
        chunk = []
        while pos < len(order):
            cand = items[order[pos]]
            w = max(width, len(cand[1]))
            cost = ntok + len(cand[1]) if packed else w * (len(chunk) + 1)
            if chunk and cost > max_batch_tokens:
                break
            chunk.append(cand)
            width = w
            ntok += len(cand[1])
            pos += 1
        B = len(chunk)
        aa_all = b"".join(ops.encode_amino_acids(c[3]) for c in chunk)
        offs = [0]
        for c in chunk:
            offs.append(offs[-1] + len(c[3]))
        if packed:
            cu = np.concatenate([[0], np.cumsum([len(c[1]) for c in chunk])])
            flat = torch.from_numpy(np.concatenate([np.asarray(c[1], np.int32) for c in chunk]))
            row0 = [int(cu[r]) + c[2] for r, c in enumerate(chunk)]
            with torch.no_grad():
                logits = model.forward_packed(flat.to(dev), [c[4] for c in chunk], cu, want_struct=False,
                                              want_representation=False)["logits"]
        else:
            tok = torch.full((B, width), _ALPHABET.padding_idx, dtype=torch.int32)
            for r, (_, ids, _, _, _) in enumerate(chunk):
                tok[r, :len(ids)] = torch.tensor(ids, dtype=torch.int32)
            species = torch.tensor([c[4] for c in chunk], dtype=torch.int32)
            row0 = [r * width + c[2] for r, c in enumerate(chunk)]
            with torch.no_grad():
                logits = model(tok.to(dev), species.to(dev))["logits"].view(B * width, -1)
        bases = ops.codon_decode(
            logits,
            torch.frombuffer(bytearray(aa_all), dtype=torch.uint8).to(dev),
            torch.tensor(offs, dtype=torch.int32, device=dev),
            torch.tensor(row0, dtype=torch.int32, device=dev))
        flat_s = bytes(bases.cpu().numpy()).decode("ascii")
        for r, c in enumerate(chunk):
            rna = flat_s[3 * offs[r]:3 * offs[r + 1]]
            if not back_translation_ok(rna, c[3]):
                raise RuntimeError(f"prediction for sequence {c[0]} does not translate back to its protein")
            out[c[0]] = rna
    return out
