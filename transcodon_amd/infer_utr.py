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


# ---- the same tokenisation as arrays ---------------------------------------------------------
# AA_tokenize -> CodonSequence_infer / WithUtr_CodonSequence_infer -> Alphabet.encode walks Python
# strings symbol by symbol (0.7 M tokens/s on one core, a seventh of what one GPU consumes); the
# batched drivers build the same token ids from byte lookup tables instead.
import numpy as _np

_TOK = _ALPHABET.tok_to_idx
_BASE_LUT = _np.full(256, -1, _np.int16)
for _ch in "AUCG":
    _BASE_LUT[ord(_ch)] = _TOK[_ch]
_BASE_LUT[ord("T")] = _TOK["U"]                       # T is read as U (sequence.py:60-62,78-79)
_TMPL_LUT = _BASE_LUT.copy()
_TMPL_LUT[ord("*")] = _TOK["<mask>"]                  # '*' of a template becomes <mask> (sequence.py:67-68)
_AA_LUT = _np.full((256, 3), -1, _np.int16)
for _aa in "FLIMVPTAYHQNKDECWRSG_":
    _AA_LUT[ord(_aa)] = [_TMPL_LUT[ord(_c)] for _c in AA_tokenize(_aa + "_")[:3]]
_STOP_IDS = _np.array([_TMPL_LUT[ord(_c)] for _c in "U**"], _np.int16)


def _lut_ids(text: str, lut) -> "_np.ndarray":
    try:
        raw = _np.frombuffer(text.replace(" ", "").encode("ascii"), _np.uint8)
    except UnicodeEncodeError as e:
        raise KeyError(text[e.start]) from None
    ids = lut[raw]
    if ids.size and int(ids.min()) < 0:
        raise KeyError(chr(int(raw[int(_np.argmax(ids < 0))])))      # the first foreign symbol, as Alphabet.encode
    return ids


def template_token_ids(aa_seq: str, utr: Optional[str] = None, max_aa: Optional[int] = None):
    """Token ids of `<cls> [utr <sep>] template <eos>` for a protein whose last character stands for
    the stop codon: what Alphabet.encode(CodonSequence_infer(AA_tokenize(aa_seq)).seq) returns (or the
    WithUtr form), as an int32 array.  Returns (ids, first CDS row, the amino acids actually covered):
    like infer.py:262-269, a template longer than 3 * max_aa bases is cut there."""
    raw = _np.frombuffer(aa_seq.encode("ascii", "replace"), _np.uint8)
    body = _AA_LUT[raw[:-1]]
    if body.size and int(body.min()) < 0:
        raise KeyError(aa_seq[int(_np.argmax(body.min(axis=1) < 0))])
    tmpl = _np.concatenate([body.reshape(-1), _STOP_IDS])
    if max_aa is not None and tmpl.size > 3 * max_aa:
        aa_seq, tmpl = aa_seq[:max_aa], tmpl[:3 * max_aa]
    parts = [_np.array([_TOK["<cls>"]], _np.int16)]
    start = 1
    if utr is not None:
        u = _lut_ids(utr, _TMPL_LUT)
        parts += [u, _np.array([_TOK["<sep>"]], _np.int16)]
        start += u.size + 1
    parts += [tmpl, _np.array([_TOK["<eos>"]], _np.int16)]
    return _np.concatenate(parts).astype(_np.int32), start, aa_seq


def cds_token_ids(cds: str) -> "_np.ndarray":
    """Token ids of `<cls> cds <eos>` (T read as U): Alphabet.encode(CodonSequence("", cds).seq)."""
    return _np.concatenate([[_TOK["<cls>"]], _lut_ids(cds, _BASE_LUT), [_TOK["<eos>"]]]).astype(_np.int32)


# codon (16 a + 4 b + c over A, U, C, G = 0..3) -> amino-acid letter as a byte, '#' for the stop codons
_B4 = _np.full(256, -1, _np.int16)
for _i, _ch in enumerate("AUCG"):
    _B4[ord(_ch)] = _i
_CODON_AA = _np.zeros(64, _np.uint8)
for _a in range(4):
    for _b in range(4):
        for _c in range(4):
            _CODON_AA[16 * _a + 4 * _b + _c] = ord(CODON_TO_AA["AUCG"[_a] + "AUCG"[_b] + "AUCG"[_c]])


def back_translation_mismatch(bases: "_np.ndarray", aa_letters: "_np.ndarray", offsets: Sequence[int]) -> int:
    """The check of back_translation_ok for a whole decoded batch at once: bases uint8 [3 * n_aa] (ASCII
    A/U/C/G), aa_letters uint8 [n_aa] (the proteins with their stop markers, concatenated), offsets
    [n_seq + 1] into the amino acids.  Every codon but each sequence's last must translate to its amino
    acid (infer.py:76-88 compares res[:-1]).  Returns the index of the first sequence that does not, or -1."""
    b = _B4[bases].astype(_np.int64).reshape(-1, 3)
    aa = _CODON_AA[16 * b[:, 0] + 4 * b[:, 1] + b[:, 2]]
    bad = aa != aa_letters
    off = _np.asarray(offsets, _np.int64)
    bad[off[1:] - 1] = False                                  # the last codon of a sequence is not compared
    if not bad.any():
        return -1
    return int(_np.searchsorted(off, int(_np.argmax(bad)), side="right") - 1)


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
    enc = [cds_token_ids(s)[:max_positions] for s in cds_seqs]
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
            tok[r, :len(enc[k])] = torch.from_numpy(_np.asarray(enc[k], _np.int32))
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
        ids, start, aa = template_token_ids(prot + "_", None if utrs is None else utrs[i], max_aa)
        items.append((i, ids, start, aa, _species_id(organisms[i])))
    with _HostCheckedIds(model, [it[4] for it in items]):
        return _optimize(model, items, dev, max_batch_tokens, packed)


def _optimize(model, items, dev, max_batch_tokens, packed):
    import numpy as np
    order = sorted(range(len(items)), key=lambda k: len(items[k][1]))
    out: List[Optional[str]] = [None] * len(items)

    def finish(host_buf, done, chunk, offs):
        """Host side of one batch: wait for its bases, check the back-translation, cut the strings."""
        done.synchronize()
        host = host_buf.numpy()
        wrong = back_translation_mismatch(host, np.frombuffer("".join(c[3] for c in chunk).encode("ascii"), np.uint8), offs)
        if wrong >= 0:
            raise RuntimeError(f"prediction for sequence {chunk[wrong][0]} does not translate back to its protein")
        flat_s = host.tobytes().decode("ascii")
        for r, c in enumerate(chunk):
            out[c[0]] = flat_s[3 * offs[r]:3 * offs[r + 1]]

    pending = None
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
                tok[r, :len(ids)] = torch.from_numpy(np.asarray(ids, np.int32))
            species = torch.tensor([c[4] for c in chunk], dtype=torch.int32)
            row0 = [r * width + c[2] for r, c in enumerate(chunk)]
            with torch.no_grad():
                # (no representation asked for: the decode reads logits only, and a large folded batch then runs
                # without any LayerNorm kernel)
                logits = model(tok.to(dev), species.to(dev), repr_layers=[])["logits"].view(B * width, -1)
        bases = ops.codon_decode(
            logits,
            torch.frombuffer(bytearray(aa_all), dtype=torch.uint8).to(dev),
            torch.tensor(offs, dtype=torch.int32, device=dev),
            torch.tensor(row0, dtype=torch.int32, device=dev))
        # The decoded bases travel to a pinned host buffer asynchronously; the host part of THIS batch (check,
        # strings) runs after the NEXT batch's kernels have been queued, so the GPU never waits for it.
        host_buf = torch.empty(bases.numel(), dtype=torch.uint8).pin_memory()
        host_buf.copy_(bases, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        if pending is not None:
            finish(*pending)
        pending = (host_buf, done, chunk, offs)
    if pending is not None:
        finish(*pending)
    return out
