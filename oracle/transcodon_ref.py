"""CPU (torch fp32) restatement of the TransCodon forward + greedy codon decode.

TEST INFRASTRUCTURE ONLY: used as the checker in tests/, in
__graft_entry__.smoke() and as bench.py's cpu_baseline.  Never imported by the
transcodon_amd package.

The restatement is functional: it works on a plain ``dict`` of tensors keyed
like the reference's ``ProteinBertModel.state_dict()`` instead of nn.Modules.
Every function cites the reference file:line it follows (paths relative to the
reference repository root).

Third-party arithmetic restated from its published algorithm:
rotary-embedding-torch==0.8.6 (environment.yml:205) -> ``rope_rotate``.
No reference test pins it: parity is unpinned at the RoPE boundary.
"""

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# Alphabet / tokeniser facts (transcodon/alphabet.py:78-81, 175-193)
# --------------------------------------------------------------------------
ALL_TOKS = ["<cls>", "<pad>", "<eos>", "<unk>", "<sep>", "A", "U", "C", "G", "<mask>"]
TOK2IDX = {t: i for i, t in enumerate(ALL_TOKS)}
CLS, PAD, EOS, UNK, SEP, MASK = 0, 1, 2, 3, 4, 9
BASE2IDX = {"A": 5, "U": 6, "C": 7, "G": 8}          # transcodon/infer_utr.py:98

# Synonymous codons in the reference's dict order (transcodon/infer_utr.py:66-88;
# identical table with '_' as the stop key in infer.py:90-112).
AA2CODONS = {
    "F": ["UUU", "UUC"],
    "L": ["UUA", "UUG", "CUU", "CUC", "CUA", "CUG"],
    "I": ["AUU", "AUC", "AUA"],
    "M": ["AUG"],
    "V": ["GUU", "GUC", "GUA", "GUG"],
    "P": ["CCU", "CCC", "CCA", "CCG"],
    "T": ["ACU", "ACC", "ACA", "ACG"],
    "A": ["GCU", "GCC", "GCA", "GCG"],
    "Y": ["UAU", "UAC"],
    "_": ["UAA", "UAG", "UGA"],
    "H": ["CAU", "CAC"],
    "Q": ["CAA", "CAG"],
    "N": ["AAU", "AAC"],
    "K": ["AAA", "AAG"],
    "D": ["GAU", "GAC"],
    "E": ["GAA", "GAG"],
    "C": ["UGU", "UGC"],
    "W": ["UGG"],
    "R": ["CGU", "CGC", "CGA", "CGG", "AGA", "AGG"],
    "S": ["AGU", "AGC", "UCU", "UCC", "UCA", "UCG"],
    "G": ["GGU", "GGC", "GGA", "GGG"],
}
CODON2AA = {c: ("#" if aa == "_" else aa) for aa, cs in AA2CODONS.items() for c in cs}


@dataclass
class Config:
    """Model hyper-parameters (transcodon/model.py:32-84 defaults)."""
    num_layers: int = 12
    embed_dim: int = 768
    ffn_embed_dim: int = 3072
    attention_heads: int = 12
    species_vocab_size: int = 1438
    alphabet_size: int = 10
    struct_classes: int = 3

    @property
    def head_dim(self):
        return self.embed_dim // self.attention_heads


TINY = Config(num_layers=2, embed_dim=128, ffn_embed_dim=512, attention_heads=2,
              species_vocab_size=16)
FULL = Config()


# --------------------------------------------------------------------------
# Host-side string handling
# --------------------------------------------------------------------------
def aa_tokenize(seq: str) -> str:
    """CDS template for a protein string whose last char is the stop marker.

    infer.py:113-129: every residue but the last gives the first two bases of
    its first synonymous codon plus '*'; L -> '*U*', R -> '*G*', S -> '***';
    the final position is always 'U**'.  Unknown letters raise KeyError.
    """
    out = []
    for ch in seq[:-1]:
        if ch == "L":
            out.append("*U*")
        elif ch == "R":
            out.append("*G*")
        elif ch == "S":
            out.append("***")
        else:
            out.append(AA2CODONS[ch][0][:2] + "*")
    out.append("U**")
    return "".join(out)


def token_strings(utr: Optional[str], template: str) -> List[str]:
    """Token list of (WithUtr_)CodonSequence_infer.

    transcodon/sequence.py:168-212 with _split_infer :60-70: '*' -> <mask>,
    spaces dropped; with a UTR both parts get T->U and are joined by <sep>;
    without a UTR the template is used as is (no T->U).  utr None or the
    string "false" selects the no-UTR form (transcodon/infer_utr.py:25-28).
    """
    def split(s):
        return ["<mask>" if c == "*" else c for c in s.replace(" ", "")]
    if utr is None or utr == "false":
        return ["<cls>"] + split(template) + ["<eos>"]
    return (["<cls>"] + split(utr.replace("T", "U")) + ["<sep>"]
            + split(template.replace("T", "U")) + ["<eos>"])


def encode(tokens: Sequence[str]) -> List[int]:
    """Alphabet.encode for already split tokens (transcodon/alphabet.py:268-269);
    a symbol outside the alphabet raises KeyError like the reference."""
    return [TOK2IDX[t] for t in tokens]


def batch_tokens(seqs: Sequence[Sequence[int]]) -> torch.Tensor:
    """BatchConverter: right-pad with <pad> to the batch maximum, int64
    (transcodon/alphabet.py:280-313 with prepend_bos=append_eos=False)."""
    L = max(len(s) for s in seqs)
    out = torch.full((len(seqs), L), PAD, dtype=torch.int64)
    for i, s in enumerate(seqs):
        out[i, : len(s)] = torch.tensor(s, dtype=torch.int64)
    return out


def back_translate_ok(rna: str, aa_seq: str) -> bool:
    """infer.py:76-88: translate the prediction and compare all but the last
    residue with the (length-clipped) input protein."""
    res = "".join(CODON2AA[rna[i:i + 3]] for i in range(0, len(rna), 3))
    return res[:-1] == aa_seq[: len(res)][:-1]


# --------------------------------------------------------------------------
# Model forward
# --------------------------------------------------------------------------
def gelu(x):
    """Exact erf GELU, transcodon/modules.py:17-24."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def rope_freqs(head_dim: int) -> torch.Tensor:
    """rotary-embedding-torch 0.8.6 'lang' frequencies, theta=10000:
    1 / theta ** (arange(0, dim, 2) / dim)  (stored as rope.freqs[dim/2])."""
    return 1.0 / (10000.0 ** (torch.arange(0, head_dim, 2).float() / head_dim))


def rope_rotate(t: torch.Tensor, freqs: torch.Tensor) -> torch.Tensor:
    """RotaryEmbedding.rotate_queries_or_keys on [..., T, D]
    (call sites transcodon/multihead_attention.py:358-360).

    Positions are 0..T-1 along dim -2; angles pos*freqs are repeated pairwise
    '(n r)', r=2; rotate_half acts on interleaved pairs:
    out[2i] = x[2i] cos - x[2i+1] sin ; out[2i+1] = x[2i+1] cos + x[2i] sin.
    """
    T = t.shape[-2]
    pos = torch.arange(T, dtype=freqs.dtype)
    ang = torch.einsum("i,j->ij", pos, freqs)               # [T, D/2]
    ang = ang.repeat_interleave(2, dim=-1)                   # [T, D]
    x1 = t[..., 0::2]
    x2 = t[..., 1::2]
    rot = torch.stack((-x2, x1), dim=-1).reshape(t.shape)
    return t * ang.cos() + rot * ang.sin()


def mha(sd: Dict[str, torch.Tensor], p: str, x: torch.Tensor, H: int,
        key_padding_mask: Optional[torch.Tensor], want_scores: bool = False):
    """MultiheadAttention.forward, RoPE branch, on x[T,B,E]
    (transcodon/multihead_attention.py:262-265, 284-288, 358-408).
    Returns (attn[T,B,E], pre-softmax scores [H,B,T,T] fp32 or None)."""
    T, B, E = x.shape
    D = E // H
    q = F.linear(x, sd[p + "q_proj.weight"], sd[p + "q_proj.bias"])
    k = F.linear(x, sd[p + "k_proj.weight"], sd[p + "k_proj.bias"])
    v = F.linear(x, sd[p + "v_proj.weight"], sd[p + "v_proj.bias"])
    q = q * (D ** -0.5)                                       # :265
    q = q.contiguous().view(T, B * H, D).transpose(0, 1)      # :284
    k = k.contiguous().view(T, B * H, D).transpose(0, 1)
    v = v.contiguous().view(T, B * H, D).transpose(0, 1)
    freqs = sd.get(p + "rope.freqs", rope_freqs(D))
    q = rope_rotate(q, freqs)                                 # :359
    k = rope_rotate(k, freqs)                                 # :360
    s = torch.bmm(q, k.transpose(1, 2))                       # :362
    if key_padding_mask is not None:                          # :373-379
        s = s.view(B, H, T, T).masked_fill(
            key_padding_mask[:, None, None, :].to(torch.bool), float("-inf")
        ).view(B * H, T, T)
    pr = F.softmax(s, dim=-1, dtype=torch.float32).type_as(s)  # :384-389
    a = torch.bmm(pr, v)                                      # :391
    a = a.transpose(0, 1).contiguous().view(T, B, E)          # :398
    a = F.linear(a, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])
    scores = None
    if want_scores:                                           # :400-403
        scores = s.view(B, H, T, T).to(torch.float32).transpose(1, 0)
    return a, scores


def layer_norm(sd, p, x, eps=1e-5):
    """ESM1bLayerNorm == torch.nn.LayerNorm(E, eps=1e-5) when apex is absent
    (transcodon/modules.py:27-41)."""
    return F.layer_norm(x, (x.shape[-1],), sd[p + "weight"], sd[p + "bias"], eps)


def transformer_layer(sd, p, x, H, key_padding_mask, want_scores=False):
    """Pre-LN block, transcodon/modules.py:83-105."""
    r = x
    x = layer_norm(sd, p + "self_attn_layer_norm.", x)
    x, sc = mha(sd, p + "self_attn.", x, H, key_padding_mask, want_scores)
    x = r + x
    r = x
    x = layer_norm(sd, p + "final_layer_norm.", x)
    x = gelu(F.linear(x, sd[p + "fc1.weight"], sd[p + "fc1.bias"]))
    x = F.linear(x, sd[p + "fc2.weight"], sd[p + "fc2.bias"])
    return r + x, sc


def lm_head(sd, x):
    """RobertaLMHead.forward, transcodon/modules.py:154-160 (weight tied to
    embed_tokens.weight, model.py:164-168)."""
    x = F.linear(x, sd["lm_head.dense.weight"], sd["lm_head.dense.bias"])
    x = layer_norm(sd, "lm_head.layer_norm.", gelu(x))
    return F.linear(x, sd["embed_tokens.weight"]) + sd["lm_head.bias"]


def struct_head(sd, x):
    """RNA_2D_RobertaLMHead.forward, transcodon/modules.py:172-178."""
    x = F.linear(x, sd["structure_linear.dense.weight"], sd["structure_linear.dense.bias"])
    x = layer_norm(sd, "structure_linear.layer_norm.", gelu(x))
    x = F.linear(x, sd["structure_linear.out_proj.weight"], sd["structure_linear.out_proj.bias"])
    return x + sd["structure_linear.bias"]


def forward(sd: Dict[str, torch.Tensor], cfg: Config, tokens: torch.Tensor,
            species: torch.Tensor, repr_layers: Sequence[int] = (12,),
            need_head_weights: bool = False) -> Dict[str, object]:
    """ProteinBertModel.forward, transcodon/model.py:175-259."""
    assert tokens.ndim == 2
    tokens = tokens.long()
    pad = tokens.eq(PAD)                                              # :178
    x = F.embedding(tokens, sd["embed_tokens.weight"])                # :180
    x = x + F.embedding(species.long(), sd["species_embedding.weight"]).unsqueeze(1)  # :201
    x = x * (1 - pad.unsqueeze(-1).type_as(x))                        # :205-206
    repr_layers = set(repr_layers)
    reps = {}
    if 0 in repr_layers:
        reps[0] = x
    x = x.transpose(0, 1)                                             # :217
    kpm = pad if bool(pad.any()) else None                            # :219-220
    attn = []
    L = cfg.num_layers
    for i in range(L):                                                # :222-230
        x, sc = transformer_layer(sd, f"layers.{i}.", x, cfg.attention_heads, kpm,
                                  want_scores=need_head_weights)
        if (i + 1) in repr_layers:
            reps[i + 1] = x.transpose(0, 1)
        if need_head_weights:
            attn.append(sc.transpose(1, 0))
    x = layer_norm(sd, "emb_layer_norm_after.", x).transpose(0, 1)    # :232-233
    if L in repr_layers:
        reps[L] = x                                                   # :236-237
    out = {"logits": lm_head(sd, x), "struct_label_output": struct_head(sd, x),
           "representations": reps}
    if need_head_weights:                                             # :246-256
        a = torch.stack(attn, 1)
        if kpm is not None:
            m = 1 - kpm.type_as(a)
            m = m.unsqueeze(1) * m.unsqueeze(2)
            a = a * m[:, None, None, :, :]
        out["attentions"] = a
    return out


# --------------------------------------------------------------------------
# Greedy synonymous-codon decode
# --------------------------------------------------------------------------
def decode(cds_logits: torch.Tensor, aa_seq: str, return_margins: bool = False):
    """transcodon/infer_utr.py:98-133 on cds_logits[3m(+),10] (already sliced
    to the CDS, batch dim removed).

    Per residue i: rows 3i..3i+2; score(codon) = python-float product, started
    at 1.0, of the fp32 softmax (over all 10 symbols) probability of each of
    its bases; first codon (dict order) with a strictly larger score wins.
    Margins (optional) = log(best) - log(second best), inf if a single codon.
    """
    probs = torch.softmax(cds_logits.float(), dim=-1).numpy().astype(np.float64)
    out = []
    margins = []
    for i, aa in enumerate(aa_seq):
        best, best_p, scores = None, float("-inf"), []
        for codon in AA2CODONS[aa]:
            p = 1.0
            for j, base in enumerate(codon):
                p *= float(probs[3 * i + j, BASE2IDX[base]])
            scores.append(p)
            if p > best_p:
                best_p, best = p, codon
        out.append(best)
        if return_margins:
            s = sorted(scores, reverse=True)
            margins.append(float("inf") if len(s) < 2 else
                           (math.log(s[0]) - math.log(s[1]) if s[1] > 0 else float("inf")))
    rna = "".join(out)
    return (rna, margins) if return_margins else rna


def slice_cds_logits(logits: torch.Tensor, utr: Optional[str]) -> torch.Tensor:
    """transcodon/infer_utr.py:60-63 for a B=1 result: drop <cls> and the last
    position, then the UTR and <sep>."""
    l = logits[:, 1:-1, :]
    if not (utr is None or utr == "false"):
        l = l[:, len(utr) + 1:, :]
    return l


def embed_sequence(sd, cfg: Config, utr: Optional[str], template: str, species: int,
                   aa_seq: str):
    """transcodon/infer_utr.py:17-159: tokenise -> forward -> slice -> decode.
    Returns (representations, sliced logits [1,3m,10], predicted RNA)."""
    toks = batch_tokens([encode(token_strings(utr, template))])
    out = forward(sd, cfg, toks, torch.tensor([species]))
    l = slice_cds_logits(out["logits"], utr)
    return out["representations"], l, decode(l[0], aa_seq)


def infer_rows(sd, cfg: Config, rows, organism2id, with_utr: bool):
    """infer.py:261-286 row loop (B=1): returns the prediction_dna column."""
    preds = []
    for row in rows:
        aa = row["protein_seq"] + "_"
        tmpl = aa_tokenize(aa)
        if len(tmpl) > 2048 * 3:
            aa = aa[:2048]
            tmpl = tmpl[: 2048 * 3]
        utr = row["utr"] if with_utr else "false"
        _, _, rna = embed_sequence(sd, cfg, utr, tmpl, organism2id[row["organism"]], aa)
        assert back_translate_ok(rna, aa)
        preds.append(rna.replace("U", "T"))
    return preds


# --------------------------------------------------------------------------
# Training objective and schedule
# --------------------------------------------------------------------------
def mlm_loss(logits, labels, struct_logits=None, struct_labels=None):
    """pretraining.py:109-130: mean CE (ignore -100) + 0.4 * struct CE when any
    struct label is set."""
    loss = F.cross_entropy(logits.reshape(-1, logits.shape[-1]).float(),
                           labels.reshape(-1).long(), ignore_index=-100)
    if struct_labels is not None and bool((struct_labels != -100).any()):
        loss = loss + 0.4 * F.cross_entropy(
            struct_logits.reshape(-1, struct_logits.shape[-1]).float(),
            struct_labels.reshape(-1).long(), ignore_index=-100)
    return loss


def warmup_cosine(step: int, warmup_steps: int, num_steps: int) -> float:
    """pretraining.py:76-82 (progress is divided by num_steps, not by
    num_steps - warmup_steps)."""
    if step < warmup_steps:
        return (step + 1) / warmup_steps
    progress = (step - warmup_steps) / num_steps
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * progress)))
