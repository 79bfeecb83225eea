"""Whole-model parity on the GPU: the HIP path (through tc_encoder_forward and
tc_codon_decode) against the golden fixtures made from the reference and against the
CPU oracle on the same seeded weights.

Stated tolerances
  f32 path : |dlogit| <= 2e-4 absolute (accumulation order differs from the CPU BLAS);
             decoded codons identical wherever the oracle's decision margin
             (log p_best - log p_second) exceeds 1e-3.
  bf16 path: |dlogit| <= 0.06 absolute and <= 0.015 RMS on logits of O(1) spread
             (bf16 operands, fp32 accumulation / softmax / LayerNorm / residual): three times the
             largest values measured over the tests of this file (2.2e-2 and 5.4e-3, printed by
             every check), so that a regression of the kernels' accuracy shows.
"""
import json
import os

import numpy as np
import pytest
import torch

import transcodon_amd as ta
from oracle import transcodon_ref as O
from oracle.weights import seeded_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda"
F32_ATOL = 2e-4
BF16_ATOL, BF16_RMS = 0.06, 0.015
MARGIN = 1e-3


def _model(cfg, sd, dtype):
    args = ta.default_args(num_layers=cfg.num_layers, embed_dim=cfg.embed_dim,
                           ffn_embed_dim=cfg.ffn_embed_dim, attention_heads=cfg.attention_heads,
                           species_vocab_size=cfg.species_vocab_size)
    m = ta.ProteinBertModel(args, ta.Alphabet.from_architecture("CodonModel"))
    m.load_state_dict(sd, strict=True)
    return m.to(DEV).eval().set_compute_dtype(dtype)


@pytest.fixture(scope="module")
def full_sd():
    return seeded_state_dict(O.FULL, 7)


@pytest.fixture(scope="module")
def full_f32(full_sd):
    return _model(O.FULL, full_sd, torch.float32)


@pytest.fixture(scope="module")
def full_bf16(full_sd):
    return _model(O.FULL, full_sd, torch.bfloat16)


def _check(out, want, keep, dtype):
    d = (out - want)[keep]
    if dtype == torch.float32:
        assert float(np.abs(d).max()) <= F32_ATOL, float(np.abs(d).max())
    else:
        print(f"  [bf16 logit error: max {float(np.abs(d).max()):.3e} rms {float(np.sqrt((d ** 2).mean())):.3e}]")
        assert float(np.abs(d).max()) <= BF16_ATOL, float(np.abs(d).max())
        assert float(np.sqrt((d ** 2).mean())) <= BF16_RMS


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_tiny_model_against_golden(golden_dir, dtype):
    g = np.load(os.path.join(golden_dir, "tiny_forward.npz"))
    sd = seeded_state_dict(O.TINY, int(g["seed"]), perturb=True)
    m = _model(O.TINY, sd, dtype)
    tok, sp = torch.from_numpy(g["tokens"]).to(DEV), torch.from_numpy(g["species"]).to(DEV)
    with torch.no_grad():
        out = m(tok, sp, repr_layers=[0, 1, 2])
    keep = g["tokens"] != O.PAD
    _check(out["logits"].cpu().numpy(), g["logits"], keep, dtype)
    _check(out["struct_label_output"].cpu().numpy(), g["struct"], keep, dtype)
    assert np.array_equal(out["representations"][0].cpu().numpy(), g["rep0"])
    tol = 1e-4 if dtype == torch.float32 else 0.1
    for i in (1, 2):
        d = np.abs(out["representations"][i].cpu().numpy() - g[f"rep{i}"])[keep]
        assert float(d.max()) <= tol, (i, float(d.max()))
    # loss through the CE kernel
    from transcodon_amd import ops
    loss, cnt, _ = ops.ce_loss(out["logits"].view(-1, 10), torch.from_numpy(g["labels"]).to(DEV).view(-1))
    got = float(loss.item()) / int(cnt.item())
    assert abs(got - float(g["loss_mlm"])) <= (1e-4 if dtype == torch.float32 else 2e-2)


def test_tiny_model_attentions_slow_path(golden_dir):
    g = np.load(os.path.join(golden_dir, "tiny_forward.npz"))
    sd = seeded_state_dict(O.TINY, int(g["seed"]), perturb=True)
    m = _model(O.TINY, sd, torch.float32)
    with torch.no_grad():
        out = m(torch.from_numpy(g["tokens"]).to(DEV), torch.from_numpy(g["species"]).to(DEV),
                repr_layers=[2], need_head_weights=True)
    a, b = out["attentions"].cpu().numpy(), g["attentions"]
    assert a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b))
    fin = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), fin)
    np.testing.assert_allclose(a[fin], b[fin], rtol=1e-3, atol=2e-4)
    keep = g["tokens"] != O.PAD
    _check(out["logits"].cpu().numpy(), g["logits"], keep, torch.float32)


def test_tiny_model_nopad_and_bad_index(golden_dir):
    g = np.load(os.path.join(golden_dir, "tiny_forward_nopad.npz"))
    sd = seeded_state_dict(O.TINY, int(g["seed"]), perturb=True)
    m = _model(O.TINY, sd, torch.float32)
    tok, sp = torch.from_numpy(g["tokens"]).to(DEV), torch.from_numpy(g["species"]).to(DEV)
    with torch.no_grad():
        out = m(tok, sp, repr_layers=[2])
    _check(out["logits"].cpu().numpy(), g["logits"], np.ones(g["tokens"].shape, bool), torch.float32)
    np.testing.assert_allclose(out["representations"][2].cpu().numpy(), g["rep2"], atol=1e-4)
    with pytest.raises(IndexError):
        with torch.no_grad():
            m(tok, torch.tensor([99], device=DEV))
    with pytest.raises(IndexError):
        with torch.no_grad():
            m(torch.full_like(tok, 11), sp)


@pytest.mark.parametrize("key", ["noutr", "utr"])
def test_full_row2_f32_logits_and_codons(golden_dir, full_f32, key):
    g = np.load(os.path.join(golden_dir, f"full_row2_{key}.npz"))
    meta = json.load(open(os.path.join(golden_dir, "full_testcsv_decode.json")))["rows"][2]
    rna_ref = meta[key]["rna"]
    aa = "".join(O.CODON2AA[rna_ref[i:i + 3]] for i in range(0, len(rna_ref), 3)).replace("#", "_")
    n_utr = len(g["tokens"]) - 2 - len(meta["template"])
    utr = "false" if key == "noutr" else "A" * (n_utr - 1)   # only its length matters below
    # feed the exact golden tokens by building the sequence object from them
    from transcodon_amd.sequence import Sequence
    seq = Sequence()
    seq._seq = " ".join(O.ALL_TOKS[t] for t in g["tokens"])
    from transcodon_amd import infer_utr
    orig = infer_utr.CodonSequence_infer
    reps, logits, rna = _embed_tokens(full_f32, g["tokens"], int(g["species"]), n_utr, aa)
    d = np.abs(logits.cpu().numpy() - g["logits"])
    assert float(d.max()) <= F32_ATOL, float(d.max())
    np.testing.assert_allclose(reps[12][0, :4].cpu().numpy(), g["rep12_head"], atol=2e-4)
    margins = g["margins"]
    for i in range(len(aa)):
        if margins[i] > MARGIN:
            assert rna[3 * i:3 * i + 3] == rna_ref[3 * i:3 * i + 3], (i, margins[i])
    assert sum(rna[3 * i:3 * i + 3] != rna_ref[3 * i:3 * i + 3] for i in range(len(aa))) <= 1


def _embed_tokens(model, tokens, species, n_utr, aa):
    """embed_sequence's steps on explicit token ids (slice after <cls>, utr and <sep>)."""
    from transcodon_amd import ops
    tok = torch.from_numpy(np.asarray(tokens))[None].to(DEV)
    with torch.no_grad():
        out = model(tok, torch.tensor([species], device=DEV))
    T = tok.shape[1]
    start = 1 + n_utr
    bases = ops.codon_decode(out["logits"].view(T, -1),
                             torch.tensor(list(ops.encode_amino_acids(aa)), dtype=torch.uint8, device=DEV),
                             torch.tensor([0, len(aa)], dtype=torch.int32, device=DEV),
                             torch.tensor([start], dtype=torch.int32, device=DEV))
    return out["representations"], out["logits"][:, start:-1], bytes(bases.cpu().numpy()).decode()


def test_full_row2_embed_sequence_api(golden_dir, full_f32):
    """The reference's call contract: embed_sequence(model, utr|'false', template, species, AA)."""
    meta = json.load(open(os.path.join(golden_dir, "full_testcsv_decode.json")))["rows"][2]
    rna_ref = meta["noutr"]["rna"]
    aa = "".join(O.CODON2AA[rna_ref[i:i + 3]] for i in range(0, len(rna_ref), 3)).replace("#", "_")
    reps, logits, rna = ta.embed_sequence(full_f32, "false", meta["template"], meta["organism"], aa)
    g = np.load(os.path.join(golden_dir, "full_row2_noutr.npz"))
    assert logits.shape == g["logits"].shape
    assert float(np.abs(logits.cpu().numpy() - g["logits"]).max()) <= F32_ATOL
    assert sum(a != b for a, b in zip(rna, rna_ref)) <= 1 and len(rna) == len(rna_ref)
    assert 12 in reps and reps[12].shape == (1, meta["noutr"]["T"], 768)


def test_full_row5_bf16_logits(golden_dir, full_bf16):
    g = np.load(os.path.join(golden_dir, "full_row5_noutr.npz"))
    tok = torch.from_numpy(g["tokens"])[None].to(DEV)
    with torch.no_grad():
        out = full_bf16(tok, torch.tensor([int(g["species"])], device=DEV))
    got = out["logits"][:, 1:-1].cpu().numpy()
    _check(got, g["logits"], np.ones(got.shape[:2], bool), torch.bfloat16)
    # token agreement of the greedy decode at bf16: reported, and bounded loosely
    meta = json.load(open(os.path.join(golden_dir, "full_testcsv_decode.json")))["rows"][5]
    rna_ref = meta["noutr"]["rna"]
    aa = "".join(O.CODON2AA[rna_ref[i:i + 3]] for i in range(0, len(rna_ref), 3)).replace("#", "_")
    _, _, rna = _embed_tokens(full_bf16, g["tokens"], int(g["species"]), 0, aa)
    agree = np.mean([rna[3 * i:3 * i + 3] == rna_ref[3 * i:3 * i + 3] for i in range(len(aa))])
    confident = [i for i in range(len(aa)) if g["margins"][i] > 0.1]
    assert all(rna[3 * i:3 * i + 3] == rna_ref[3 * i:3 * i + 3] for i in confident)
    print(f"  [bf16 codon agreement {agree:.4f}]")
    assert agree >= 0.98                  # measured: 1.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_full_padded_batch(golden_dir, full_f32, full_bf16, dtype):
    g = np.load(os.path.join(golden_dir, "full_padded_batch.npz"))
    m = full_f32 if dtype == torch.float32 else full_bf16
    with torch.no_grad():
        out = m(torch.from_numpy(g["tokens"]).to(DEV), torch.from_numpy(g["species"]).to(DEV))
    keep = g["tokens"] != O.PAD
    _check(out["logits"].cpu().numpy(), g["logits"], keep, dtype)
    _check(out["struct_label_output"].cpu().numpy(), g["struct"], keep, dtype)


@pytest.mark.parametrize("packed", [True, False])
@pytest.mark.parametrize("key", ["noutr", "utr"])
def test_testcsv_f32_codon_agreement(golden_dir, full_f32, key, packed):
    """cfg1: every bundled test.csv protein, without and with its 5'UTR, through the batched driver
    on the f32 path, against the RNA the reference's embed_sequence produced on the same weights.
    "Bit-exact except at ties": every codon that differs from the reference's must sit at a
    position where the reference's own decision margin (log p_best - log p_second, stored per
    amino acid in full_testcsv_margins.npz) is below MARGIN; the count is printed."""
    meta = json.load(open(os.path.join(golden_dir, "full_testcsv_decode.json")))["rows"]
    margins = np.load(os.path.join(golden_dir, "full_testcsv_margins.npz"))
    prots, orgs, refs, utrs = [], [], [], []
    for r in meta:
        rna = r[key]["rna"]
        aa = "".join(O.CODON2AA[rna[i:i + 3]] for i in range(0, len(rna), 3))
        prots.append(aa[:-1])
        orgs.append(r["organism"])
        refs.append(rna)
        utrs.append(r["utr_seq"])
    got = ta.optimize_codons(full_f32, prots, orgs, utrs if key == "utr" else None, max_batch_tokens=4096, packed=packed)
    total = diff = 0
    worst = 0.0
    for ri, (a, b) in enumerate(zip(got, refs)):
        assert len(a) == len(b)
        mg = margins[f"r{ri}_{key}"]
        assert len(mg) == len(a) // 3
        for i in range(0, len(a), 3):
            total += 1
            if a[i:i + 3] != b[i:i + 3]:
                diff += 1
                worst = max(worst, float(mg[i // 3]))
                assert mg[i // 3] < MARGIN, (ri, i // 3, float(mg[i // 3]), a[i:i + 3], b[i:i + 3])
    print(f"test.csv {key} packed={packed}: {diff} of {total} codons differ from the reference, all at margins <= {worst:.2e}")
    assert total == sum(r["protein_len"] + 1 for r in meta)
    assert diff <= max(2, total // 1000), (diff, total)


@pytest.mark.parametrize("B,T,rows", [(64, 512, [0, 17, 63]), (32, 1024, [3, 31])])
def test_headline_kernel_mix_bf16_against_oracle(full_sd, full_bf16, B, T, rows):
    """64 x 512 tokens in bf16: the shapes that take the persistent 256 x 256 GEMM kernels, the
    LDS-staged rotary epilogue (T % 256 == 0), the 8-wave attention kernel and head_rows_kernel, i.e.
    the kernel mix of the 256 x 512 headline batch; 32 x 1024 is the row length of the 4096 x 1024
    batch-sharded configuration (four rotary slices and sixteen key tiles per sequence).  Sequences
    are independent, so the CPU oracle only runs the rows that are compared."""
    rng = np.random.default_rng(4321 + T)
    tok = np.empty((B, T), np.int64)
    for b in range(B):
        u = int(rng.integers(20, 101))
        body = rng.choice([5, 6, 7, 8, 9], T - 3 - u, p=[.17, .17, .17, .16, .33])
        tok[b] = [O.CLS] + list(rng.integers(5, 9, u)) + [O.SEP] + list(body) + [O.EOS]
    sp = rng.integers(0, 1438, B)
    with torch.no_grad():
        out = full_bf16(torch.from_numpy(tok).to(DEV), torch.from_numpy(sp).to(DEV))
    ref = O.forward(full_sd, O.FULL, torch.from_numpy(tok[rows]), torch.from_numpy(sp[rows]))
    keep = np.ones((len(rows), T), bool)
    _check(out["logits"][rows].cpu().numpy(), ref["logits"].numpy(), keep, torch.bfloat16)
    _check(out["struct_label_output"][rows].cpu().numpy(), ref["struct_label_output"].numpy(), keep, torch.bfloat16)
    d = np.abs(out["representations"][12][rows].cpu().numpy() - ref["representations"][12].numpy())
    assert float(d.max()) < 0.15


def test_folded_layernorm_forward_against_separate_kernels_and_oracle(full_sd, full_bf16):
    """The same 64 x 512 bf16 batch with the layers' LayerNorms folded into the products (default for
    batches this large) and as separate kernels: both within the bf16 tolerance of the oracle, the folded
    run no further from it than the separate one by more than a quarter, 23 row-statistics launches
    where 24 LayerNorm launches were, and padded rows handled (one sequence is half padding)."""
    from transcodon_amd import _lib
    rng = np.random.default_rng(99)
    B, T = 64, 512
    tok = rng.integers(5, 10, (B, T))
    tok[:, 0] = O.CLS
    tok[:, -1] = O.EOS
    tok[3, 256:] = 1                                   # padding tail
    tok[3, 255] = O.EOS
    sp = rng.integers(0, 1438, B)
    t_d, s_d = torch.from_numpy(tok).to(DEV), torch.from_numpy(sp).to(DEV)
    lib = _lib.load()
    assert lib.tc_ln_fold_supported(_lib.TC_BF16, B, T, 768, 3072) == 1
    runs = {}
    for fold in (True, False):
        full_bf16.fold_layernorm = fold
        lib.tc_prof_reset()
        lib.tc_prof_enable(1)
        with torch.no_grad():
            out = full_bf16(t_d, s_d, repr_layers=[0, 5, 12])
        torch.cuda.synchronize()
        lib.tc_prof_enable(0)
        runs[fold] = (out["logits"].float().cpu().numpy(), _lib.prof_collect(_lib.PROF_LN_ROWS)[0],
                      _lib.prof_collect(_lib.PROF_LAYERNORM)[0],
                      {k: v[[0, 3, 41]].float().cpu().numpy() for k, v in out["representations"].items()})
    full_bf16.fold_layernorm = True
    assert runs[True][1] == 23 and runs[True][2] == 1
    assert runs[False][1] == 0 and runs[False][2] == 25
    rows = [0, 3, 41]
    ref = O.forward(full_sd, O.FULL, torch.from_numpy(tok[rows]), torch.from_numpy(sp[rows]))["logits"].numpy()
    keep = tok[rows] != 1
    err = {}
    for fold in (True, False):
        got = runs[fold][0][rows]
        _check(got, ref, keep, torch.bfloat16)
        err[fold] = float(np.sqrt(np.mean((got - ref)[keep] ** 2)))
    print(f"\nrms logit error against the oracle: folded {err[True]:.4e}, separate kernels {err[False]:.4e}")
    assert err[True] <= 1.25 * err[False] + 1e-4
    # representations: the folded forward has no f32 residual stream (bf16 hi + lo planes); layer 0 is the
    # embedding sum (exact to 16 bits of mantissa), the inner layer and the final states stay within
    # the bf16 path's tolerance of the oracle's
    ref_rep = O.forward(full_sd, O.FULL, torch.from_numpy(tok[rows]), torch.from_numpy(sp[rows]), repr_layers=(0, 5, 12))["representations"]
    for fold in (True, False):
        rep = runs[fold][3]
        assert set(rep) == {0, 5, 12}
        np.testing.assert_allclose(rep[0][keep], ref_rep[0].numpy()[keep], rtol=2.0 ** -15, atol=1e-6)
        for layer in (5, 12):
            d = np.abs(rep[layer] - ref_rep[layer].numpy())[keep]
            assert float(d.max()) < 0.15, (fold, layer, float(d.max()))


def test_final_layernorm_folded_into_the_head_products(full_sd, full_bf16):
    """A folded forward that is not asked for representations[12] (what the batched codon driver calls) applies
    emb_layer_norm_after inside the two head dense layers (model.py:232-240): no LayerNorm kernel at all, 24
    row-statistics launches, logits and structure logits within the bf16 tolerance of the oracle and as close to it
    as the forward that runs the final LayerNorm as a kernel."""
    from transcodon_amd import _lib
    rng = np.random.default_rng(101)
    B, T = 64, 512
    tok = rng.integers(5, 10, (B, T))
    tok[:, 0] = O.CLS
    tok[:, -1] = O.EOS
    tok[7, 300:] = 1
    tok[7, 299] = O.EOS
    sp = rng.integers(0, 1438, B)
    t_d, s_d = torch.from_numpy(tok).to(DEV), torch.from_numpy(sp).to(DEV)
    lib = _lib.load()
    runs = {}
    for reps in ((), (12,)):
        lib.tc_prof_reset()
        lib.tc_prof_enable(1)
        with torch.no_grad():
            out = full_bf16(t_d, s_d, repr_layers=list(reps))
        torch.cuda.synchronize()
        lib.tc_prof_enable(0)
        runs[reps] = (out["logits"].float().cpu().numpy(), out["struct_label_output"].float().cpu().numpy(),
                      _lib.prof_collect(_lib.PROF_LN_ROWS)[0], _lib.prof_collect(_lib.PROF_LAYERNORM)[0])
    assert runs[()][2] == 24 and runs[()][3] == 0
    assert runs[(12,)][2] == 23 and runs[(12,)][3] == 1
    rows = [0, 7, 33]
    ref = O.forward(full_sd, O.FULL, torch.from_numpy(tok[rows]), torch.from_numpy(sp[rows]))
    keep = tok[rows] != 1
    err = {}
    for reps in ((), (12,)):
        got = runs[reps][0][rows]
        _check(got, ref["logits"].numpy(), keep, torch.bfloat16)
        err[reps] = float(np.sqrt(np.mean((got - ref["logits"].numpy())[keep] ** 2)))
    print(f"\nrms logit error against the oracle: final LayerNorm folded {err[()]:.4e}, as a kernel {err[(12,)]:.4e}")
    assert err[()] <= 1.25 * err[(12,)] + 1e-4
    ds = np.abs(runs[()][1][rows] - ref["struct_label_output"].numpy())[keep]
    assert float(ds.max()) <= BF16_ATOL, float(ds.max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_multihead_attention_module_against_oracle(dtype):
    """MultiheadAttention.forward on its own (the sub-module API of multihead_attention.py:162-408):
    output, and with need_head_weights=True the pre-softmax per-head scores with -inf at padded keys
    (:400-403), against the oracle's restatement of that forward."""
    from transcodon_amd.multihead_attention import MultiheadAttention
    torch.manual_seed(5)
    E, H, T, B = 128, 2, 37, 3
    mod = MultiheadAttention(E, H, rope_embedding=True)
    with torch.no_grad():
        for lin in (mod.q_proj, mod.k_proj, mod.v_proj, mod.out_proj):
            lin.bias.copy_(torch.randn(E) * 0.1)
    sd = {"a." + k: v.detach().clone() for k, v in mod.state_dict().items()}
    x = torch.randn(T, B, E)
    kpm = torch.zeros(B, T, dtype=torch.bool)
    kpm[1, 30:] = True
    kpm[2, 11:] = True
    mod = mod.to(DEV)
    mod.compute_dtype = dtype
    for mask in (None, kpm):
        ref_out, ref_sc = O.mha(sd, "a.", x, H, mask, want_scores=True)
        with torch.no_grad():
            got, w = mod(x.to(DEV), x.to(DEV) if False else None, None,
                         key_padding_mask=None if mask is None else mask.to(DEV), need_head_weights=True)
            got2, w2 = mod(x.to(DEV), key_padding_mask=None if mask is None else mask.to(DEV))
        assert w2 is None and torch.equal(got, got2)
        assert got.shape == (T, B, E) and w.shape == (H, B, T, T) and w.dtype == torch.float32
        live = np.ones((T, B), bool) if mask is None else ~mask.numpy().T       # pad queries: unspecified
        tol = 2e-5 if dtype == torch.float32 else 3e-2
        d = np.abs(got.float().cpu().numpy() - ref_out.numpy())[live]
        assert float(d.max()) <= tol, float(d.max())
        a, b = w.cpu().numpy(), ref_sc.numpy()
        assert np.array_equal(np.isneginf(a), np.isneginf(b))
        fin = np.isfinite(b)
        np.testing.assert_allclose(a[fin], b[fin], atol=2e-5 if dtype == torch.float32 else 5e-2)
    with pytest.raises(NotImplementedError):
        mod(x.to(DEV), attn_mask=torch.zeros(T, T, device=DEV))


def test_synthetic_batch_oracle_vs_hip_bf16(full_sd, full_bf16):
    """Same seeded synthetic UTR+CDS batch through the CPU oracle and the bf16 HIP path."""
    rng = np.random.default_rng(1234)
    B, T = 4, 256
    tok = np.full((B, T), O.PAD, np.int64)
    for b in range(B):
        L = T - 17 * b
        u = int(rng.integers(20, 60))
        row = [O.CLS] + list(rng.integers(5, 9, u)) + [O.SEP]
        row += list(rng.choice([5, 6, 7, 8, 9], L - len(row) - 1, p=[.17, .17, .17, .16, .33])) + [O.EOS]
        tok[b, :L] = row
    sp = rng.integers(0, 1438, B)
    ref = O.forward(full_sd, O.FULL, torch.from_numpy(tok), torch.from_numpy(sp))
    with torch.no_grad():
        out = full_bf16(torch.from_numpy(tok).to(DEV), torch.from_numpy(sp).to(DEV))
    keep = tok != O.PAD
    _check(out["logits"].cpu().numpy(), ref["logits"].numpy(), keep, torch.bfloat16)
    d = np.abs(out["representations"][12].cpu().numpy() - ref["representations"][12].numpy())[keep]
    assert float(d.max()) < 0.15


@pytest.mark.parametrize("packed", [True, False])
def test_embedding_extraction_varlen(full_sd, full_f32, packed):
    """cfg5 in miniature: variable-length CDS batch, packed (no padding rows) or padded + key-masked;
    mean-pooled final states against the oracle run one sequence at a time (as infer_embedding.py does)."""
    rng = np.random.default_rng(1237)
    seqs = ["".join(rng.choice(list("ACGT"), size=3 * int(n))) for n in (30, 77, 31, 120, 64)]
    orgs = [int(x) for x in rng.integers(0, 1438, len(seqs))]
    got = ta.extract_embeddings(full_f32, seqs, orgs, max_batch_tokens=400, packed=packed)
    assert got.shape == (len(seqs), 768)
    for i, s in enumerate(seqs):
        toks = O.batch_tokens([O.encode(["<cls>"] + list(s.replace("T", "U")) + ["<eos>"])])
        ref = O.forward(full_sd, O.FULL, toks, torch.tensor([orgs[i]]))["representations"][12][0, 1:-1].mean(0)
        np.testing.assert_allclose(got[i].numpy(), ref.numpy(), atol=2e-4)
    full = ta.extract_embeddings(full_f32, seqs[:2], orgs[:2], pooled=False, packed=packed)
    assert [t.shape for t in full] == [(90, 768), (231, 768)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_packed_forward_equals_padded_and_oracle(full_sd, full_f32, full_bf16, dtype):
    """Packed variable-length forward (rows = sum of lengths, rotary positions restarting per sequence,
    attention per (sequence, query block)) against the padded forward of the same sequences row for
    row, and against the oracle one sequence at a time (infer_embedding.py:17-148 runs B = 1)."""
    rng = np.random.default_rng(99)
    lens = [92, 233, 95, 362, 600, 194, 257]             # below, at and above the query-block sizes (64 / 256)
    m = full_f32 if dtype == torch.float32 else full_bf16
    T = max(lens)
    tok = np.full((len(lens), T), O.PAD, np.int64)
    for b, n in enumerate(lens):
        tok[b, :n] = [O.CLS] + list(rng.choice([5, 6, 7, 8, 9], n - 2)) + [O.EOS]
    sp = rng.integers(0, 1438, len(lens))
    cu = np.concatenate([[0], np.cumsum(lens)])
    flat = np.concatenate([tok[b, :n] for b, n in enumerate(lens)])
    with torch.no_grad():
        pad = m(torch.from_numpy(tok).to(DEV), torch.from_numpy(sp).to(DEV))
        pk = m.forward_packed(torch.from_numpy(flat).to(DEV), sp, cu)
        only_rep = m.forward_packed(torch.from_numpy(flat).to(DEV), sp, cu, want_logits=False)
    assert set(only_rep) == {"representations"}
    assert torch.equal(only_rep["representations"][12], pk["representations"][12])
    tol = 2e-5 if dtype == torch.float32 else 0.03
    for b, n in enumerate(lens):
        rows = slice(int(cu[b]), int(cu[b + 1]))
        for key in ("logits", "struct_label_output"):
            d = (pk[key][rows] - pad[key][b, :n]).abs().max()
            assert float(d) <= tol, (key, b, float(d))
        d = (pk["representations"][12][rows] - pad["representations"][12][b, :n]).abs().max()
        assert float(d) <= (1e-4 if dtype == torch.float32 else 0.15), (b, float(d))
    for b in (0, 4, 6):
        n = lens[b]
        ref = O.forward(full_sd, O.FULL, torch.from_numpy(tok[b:b + 1, :n]), torch.from_numpy(sp[b:b + 1]))
        rows = slice(int(cu[b]), int(cu[b + 1]))
        _check(pk["logits"][rows].cpu().numpy()[None], ref["logits"].numpy(), np.ones((1, n), bool), dtype)
    with pytest.raises(ValueError):
        m.forward_packed(torch.from_numpy(flat).to(DEV), sp[:-1], cu)


def test_packed_forward_large_batch_bf16(full_sd, full_bf16):
    """A packed batch big enough for the persistent 256 x 256 GEMM kernels (row count not a multiple
    of the tile): a few of its sequences against the oracle."""
    rng = np.random.default_rng(1237)
    lens = np.clip(3 * np.exp(rng.normal(5.0, 0.6, 300)).astype(int) + 2, 92, 2000)
    lens = lens[np.cumsum(lens) <= 70000]
    toks = [np.array([O.CLS] + list(rng.integers(5, 9, n - 2)) + [O.EOS]) for n in lens]
    sp = rng.integers(0, 1438, len(lens))
    cu = np.concatenate([[0], np.cumsum(lens)])
    assert cu[-1] > 65536 and cu[-1] % 256 != 0
    with torch.no_grad():
        pk = full_bf16.forward_packed(torch.from_numpy(np.concatenate(toks)).to(DEV), sp, cu)
    for b in (0, len(lens) // 2, len(lens) - 1):
        ref = O.forward(full_sd, O.FULL, torch.from_numpy(toks[b])[None], torch.from_numpy(sp[b:b + 1]))
        rows = slice(int(cu[b]), int(cu[b + 1]))
        _check(pk["logits"][rows].cpu().numpy()[None], ref["logits"].numpy(), np.ones((1, lens[b]), bool), torch.bfloat16)
        d = np.abs(pk["representations"][12][rows].cpu().numpy() - ref["representations"][12][0].numpy())
        assert float(d.max()) < 0.15


def test_packed_forward_with_and_without_row_panels(full_bf16):
    """The packed bf16 forward takes its q/k/v product over the layout's row panels (both LayerNorms of a layer
    folded into the products); a layout without the panel list keeps the first LayerNorm a kernel and gathers
    the rotary factors per row.  Same model, same rows: the two agree to the bf16 tolerance of the forward, and
    the decoded codon choice is the same wherever the winning margin exceeds it."""
    rng = np.random.default_rng(4242)
    lens = np.clip(3 * np.exp(rng.normal(5.0, 0.7, 400)).astype(int) + 2, 3, 2000)
    lens[:6] = [3, 255, 256, 257, 513, 1999]
    lens = lens[np.cumsum(lens) <= 70000]
    toks = np.concatenate([np.array([O.CLS] + list(rng.integers(5, 9, n - 2)) + [O.EOS]) for n in lens])
    sp = rng.integers(0, 1438, len(lens))
    cu = np.concatenate([[0], np.cumsum(lens)])
    lay = full_bf16.packed_layout(sp, cu, DEV)
    assert lay["n_panels"] == int(((lens + 255) // 256).sum())
    pan = lay["panels"].cpu().numpy()
    assert (pan[:, 1] - pan[:, 0] <= 256).all() and (pan[:, 2] % 256 == 0).all()
    assert int((pan[:, 1] - pan[:, 0]).sum()) == int(cu[-1])
    bare = {k: v for k, v in lay.items() if k not in ("panels", "n_panels")}
    flat = torch.from_numpy(toks).to(DEV)
    with torch.no_grad():
        a = full_bf16.forward_packed(flat, sp, cu, layout=lay)
        b = full_bf16.forward_packed(flat, sp, cu, layout=bare)
    d = (a["logits"] - b["logits"]).abs()
    assert float(d.max()) <= 0.06, float(d.max())
    assert float(d.pow(2).mean().sqrt()) <= 0.01
    top = a["logits"].topk(2, dim=-1).values
    sure = (top[:, 0] - top[:, 1]) > 0.12
    assert bool((a["logits"].argmax(-1)[sure] == b["logits"].argmax(-1)[sure]).all())
    dr = (a["representations"][12] - b["representations"][12]).abs().max()
    assert float(dr) <= 0.2, float(dr)


def test_forwards_on_two_streams_with_their_own_workspaces(full_bf16):
    """Two captured forwards with private workspaces replayed on two streams, and two packed forwards in two
    workspace slots, run side by side (nothing orders one stream against the other): every result equals the
    one-at-a-time result bit for bit, also when the same graph is replayed again right behind."""
    m = full_bf16
    rng = np.random.default_rng(77)
    B, T = 64, 512
    toks = [torch.from_numpy(rng.integers(5, 9, (B, T))).to(DEV) for _ in range(2)]
    for t in toks:
        t[:, 0], t[:, -1] = O.CLS, O.EOS
    sps = [torch.from_numpy(rng.integers(0, 1438, B)).to(DEV) for _ in range(2)]
    with torch.no_grad():
        want = [m(t, s)["logits"].clone() for t, s in zip(toks, sps)]
    graphs = [m.capture_forward(B, T, private_workspace=True) for _ in range(2)]
    assert graphs[0]._workspace.data_ptr() != graphs[1]._workspace.data_ptr()
    lanes = [torch.cuda.Stream() for _ in range(2)]
    torch.cuda.synchronize()
    got = [[], []]
    for rnd in range(3):
        for k in range(2):
            with torch.cuda.stream(lanes[k]):
                got[k].append(graphs[k](toks[k], sps[k], check_indices=False)["logits"].clone())
    torch.cuda.synchronize()
    for k in range(2):
        for g in got[k]:
            assert torch.equal(g, want[k])
    # packed batches in two workspace slots
    lens = [np.clip(3 * np.exp(rng.normal(5.0, 0.6, 200)).astype(int) + 2, 30, 1500) for _ in range(2)]
    lens = [l[np.cumsum(l) <= 40000] for l in lens]
    flat = [torch.from_numpy(np.concatenate([np.array([O.CLS] + list(rng.integers(5, 9, n - 2)) + [O.EOS]) for n in l])).to(DEV)
            for l in lens]
    sp = [rng.integers(0, 1438, len(l)) for l in lens]
    cu = [np.concatenate([[0], np.cumsum(l)]) for l in lens]
    lay = [m.packed_layout(sp[k], cu[k], DEV) for k in range(2)]
    old_check, m.check_indices = m.check_indices, False
    try:
        with torch.no_grad():
            want_p = [m.forward_packed(flat[k], None, None, want_logits=False, layout=lay[k])["representations"][12].clone()
                      for k in range(2)]
            torch.cuda.synchronize()
            got_p = [[], []]
            for rnd in range(3):
                for k in range(2):
                    with torch.cuda.stream(lanes[k]), m.workspace_slot(k):
                        got_p[k].append(m.forward_packed(flat[k], None, None, want_logits=False,
                                                         layout=lay[k])["representations"][12])
            torch.cuda.synchronize()
    finally:
        m.check_indices = old_check
    for k in range(2):
        for g in got_p[k]:
            assert torch.equal(g, want_p[k])


def test_cli_csv_round_trip(tmp_path):
    """python -m transcodon_amd.infer on a small CSV: same flags and columns as infer.py:229-302,
    predictions equal to the batched API and to the row-by-row embed_sequence path."""
    import pandas as pd
    from transcodon_amd import checkpoint, infer
    cfg = O.TINY
    sd = seeded_state_dict(cfg, 5, perturb=True)
    m = _model(cfg, sd, torch.float32)
    ck = str(tmp_path / "tiny.safetensors")
    checkpoint.save(m, ck)
    names = list(ta.ORGANISM2ID)[:cfg.species_vocab_size]
    rows = {"protein_seq": ["MKT", "MALWMRLLPLLALLALWGPDPAAA", "MSRL", "M" * 40 + "LSR", "MGG"],
            "organism": [names[1], names[0], names[7], names[3], names[1]],
            "utr": ["ACGU", "GGGACU", "A", "CUCUCUGG", "UUAC"]}
    src = str(tmp_path / "in.csv")
    pd.DataFrame(rows).to_csv(src, index=False)
    common = ["--model_checkpoint", ck, "--input_data", src, "--num_layers", str(cfg.num_layers), "--embed_dim",
              str(cfg.embed_dim), "--ffn_embed_dim", str(cfg.ffn_embed_dim), "--attention_heads",
              str(cfg.attention_heads), "--species_vocab_size", str(cfg.species_vocab_size)]
    for with_utr in (False, True):
        out = str(tmp_path / f"out{int(with_utr)}.csv")          # bare directory-less names also work
        infer.main(common + ["--output_file", out] + (["--with_utr"] if with_utr else []) +
                   (["--batch_size", "2"] if with_utr else []))
        got = pd.read_csv(out)
        assert list(got.columns) == ["organism", "protein_seq", "prediction_dna"]
        assert list(got["protein_seq"]) == rows["protein_seq"] and list(got["organism"]) == rows["organism"]
        want = ta.optimize_codons(m, rows["protein_seq"], rows["organism"], rows["utr"] if with_utr else None)
        assert list(got["prediction_dna"]) == [w.replace("U", "T") for w in want]
        for i in (0, 3):                                           # the reference's own row-by-row call
            aa = rows["protein_seq"][i] + "_"
            _, _, rna = ta.embed_sequence(m, rows["utr"][i] if with_utr else "false", ta.AA_tokenize(aa),
                                          rows["organism"][i], aa)
            assert rna.replace("U", "T") == got["prediction_dna"][i]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_long_ragged_sequences_against_oracle(full_sd, full_f32, full_bf16, dtype):
    """T = 1024 (the long-sequence configuration), one full row and one padded row."""
    rng = np.random.default_rng(77)
    B, T = 2, 1024
    tok = np.full((B, T), O.PAD, np.int64)
    for b, L in enumerate((T, 777)):
        tok[b, :L] = [O.CLS] + list(rng.choice([5, 6, 7, 8, 9], L - 2)) + [O.EOS]
    sp = rng.integers(0, 1438, B)
    ref = O.forward(full_sd, O.FULL, torch.from_numpy(tok), torch.from_numpy(sp))
    m = full_f32 if dtype == torch.float32 else full_bf16
    with torch.no_grad():
        out = m(torch.from_numpy(tok).to(DEV), torch.from_numpy(sp).to(DEV))
    _check(out["logits"].cpu().numpy(), ref["logits"].numpy(), tok != O.PAD, dtype)


def test_large_batch_matches_small_batch_bf16(full_bf16):
    """At 64 x 512 tokens the persistent large-tile GEMM kernels run; a 4-sequence batch uses the
    small-tile ones.  The same sequences must give the same logits either way (bf16 operands,
    fp32 accumulation in both; only the summation order differs)."""
    rng = np.random.default_rng(5)
    T = 512
    base = np.stack([[O.CLS] + list(rng.choice([5, 6, 7, 8, 9], T - 2)) + [O.EOS] for _ in range(4)])
    sp4 = rng.integers(0, 1438, 4)
    big, sp = np.tile(base, (16, 1)), np.tile(sp4, 16)
    with torch.no_grad():
        small = full_bf16(torch.from_numpy(base).to(DEV), torch.from_numpy(sp4).to(DEV))["logits"].cpu().numpy()
        large = full_bf16(torch.from_numpy(big).to(DEV), torch.from_numpy(sp).to(DEV))["logits"].cpu().numpy()
    for rep in (0, 7, 15):
        d = np.abs(large[4 * rep:4 * rep + 4] - small)
        assert float(d.max()) <= 0.03 and float(np.sqrt((d ** 2).mean())) <= 0.005, (rep, float(d.max()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_graph_captured_forward_equals_eager(full_f32, full_bf16, dtype):
    """The forward of a fixed batch shape replayed from one HIP graph gives bit for bit what the eager call
    gives, for inputs other than the ones it was captured with (padding included), twice in a row; the
    out-of-range check still fires after a replay; another shape is refused."""
    model = full_f32 if dtype == torch.float32 else full_bf16
    rng = np.random.default_rng(11)
    B, T = 3, 96
    graphed = model.capture_forward(B, T, repr_layers=[0, 6, 12])
    for trial in range(2):
        tok = np.full((B, T), O.PAD, np.int64)
        for b, n in enumerate((T, 70 - 9 * trial, 33)):
            tok[b, :n] = [O.CLS] + list(rng.choice([5, 6, 7, 8, 9], n - 2)) + [O.EOS]
        sp = rng.integers(0, 1438, B)
        tok_d, sp_d = torch.from_numpy(tok).to(DEV), torch.from_numpy(sp).to(DEV)
        with torch.no_grad():
            eager = model(tok_d, sp_d, repr_layers=[0, 6, 12])
        got = graphed(tok_d, sp_d)
        assert torch.equal(got["logits"], eager["logits"])
        assert torch.equal(got["struct_label_output"], eager["struct_label_output"])
        for layer in (0, 6, 12):
            assert torch.equal(got["representations"][layer], eager["representations"][layer]), layer
    bad = tok_d.clone()
    bad[1, 3] = 77
    with pytest.raises(IndexError):
        graphed(bad, sp_d)
    graphed(tok_d, sp_d)                                   # and the graph is usable again afterwards
    with pytest.raises(ValueError):
        graphed(tok_d[:, :50], sp_d)


def test_graph_replays_back_to_back_with_folded_layernorms(full_bf16):
    """Replays launched one after the other with no host synchronisation in between, at a shape that takes
    the folded-LayerNorm path (its row accumulators are cleared at the start of every forward, by a kernel:
    a memset node was not ordered against the previous launch of the graph and cleared sums still in use),
    and with inner representations (copied by a kernel for the same reason): every replay equals eager."""
    rng = np.random.default_rng(17)
    B, T = 64, 512
    from transcodon_amd import _lib
    assert _lib.load().tc_ln_fold_supported(_lib.TC_BF16, B, T, 768, 3072) != 0
    tok = np.stack([[O.CLS] + list(rng.choice([5, 6, 7, 8, 9], T - 2)) + [O.EOS] for _ in range(B)])
    sp = rng.integers(0, 1438, B)
    tok_d, sp_d = torch.from_numpy(tok).to(DEV), torch.from_numpy(sp).to(DEV)
    with torch.no_grad():
        eager = full_bf16(tok_d, sp_d, repr_layers=[3, 12])
    graphed = full_bf16.capture_forward(B, T, repr_layers=[3, 12])
    for n in (1, 3, 2):
        for _ in range(n):
            got = graphed(tok_d, sp_d, check_indices=False)
        logits, r3 = got["logits"].clone(), got["representations"][3].clone()
        torch.cuda.synchronize()
        assert torch.equal(logits, eager["logits"]), n
        assert torch.equal(r3, eager["representations"][3]), n


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_graph_follows_weight_changes_and_workspace_regrowth(dtype):
    """A captured forward addresses the model's packed weights and workspace by raw pointers.  After
    load_state_dict with other weights, after an eager forward that makes the model's workspace grow,
    and after an AdamW step on a flat parameter store, the next replay must give what the eager forward
    gives on the weights as they are now (the graph keeps what it addresses alive and is captured again
    when the weights changed), and must not be captured again when nothing changed."""
    from transcodon_amd.train import AdamW, FlatParameters
    cfg = O.TINY
    m = _model(cfg, seeded_state_dict(cfg, 3, perturb=True), dtype)
    rng = np.random.default_rng(5)
    B, T = 3, 64
    tok = np.full((B, T), O.PAD, np.int64)
    for b, n in enumerate((T, 40, 17)):
        tok[b, :n] = [O.CLS] + list(rng.choice([5, 6, 7, 8, 9], n - 2)) + [O.EOS]
    tok_d = torch.from_numpy(tok).to(DEV)
    sp_d = torch.from_numpy(rng.integers(0, cfg.species_vocab_size, B)).to(DEV)
    keep = tok_d != O.PAD

    def same(graphed):
        got = {k: v.clone() for k, v in graphed(tok_d, sp_d).items() if k != "representations"}
        with torch.no_grad():
            eager = m(tok_d, sp_d)
        for k in got:
            assert torch.equal(got[k][keep], eager[k][keep]), k
        return got["logits"]

    graphed = m.capture_forward(B, T)
    first = same(graphed)
    same(graphed)
    assert graphed.captures == 1
    # other weights into the same parameters
    m.load_state_dict(seeded_state_dict(cfg, 4, perturb=True), strict=True)
    second = same(graphed)
    assert graphed.captures == 2
    assert not torch.equal(first[keep], second[keep])
    # the model's shared workspace is replaced by a larger one; junk goes through the freed blocks
    with torch.no_grad():
        m(torch.from_numpy(rng.integers(5, 9, (8, 512))).to(DEV), torch.zeros(8, dtype=torch.int64, device=DEV))
    junk = [torch.full((1 << 20,), float("nan"), device=DEV) for _ in range(8)]
    third = same(graphed)
    assert torch.equal(second[keep], third[keep]) and graphed.captures == 2
    del junk
    # parameters re-homed into a flat store, then changed by the optimiser kernel behind torch's back
    for p in m.parameters():
        p.requires_grad_(True)
    store = FlatParameters(m)
    opt = AdamW(store, lr=0.05)
    fourth = same(graphed)
    assert torch.allclose(third[keep], fourth[keep], atol=1e-6)
    n_cap = graphed.captures
    store.grad.normal_(generator=torch.Generator(device=DEV).manual_seed(1))
    opt.step()
    fifth = same(graphed)
    assert graphed.captures == n_cap + 1
    assert not torch.allclose(fourth[keep], fifth[keep], atol=1e-4)
    # a frozen graph (check_weights off) keeps replaying and is the caller's promise
    graphed.check_weights = False
    graphed(tok_d, sp_d)
    assert graphed.captures == n_cap + 1


def test_headline_size_determinism_and_batch_permutation(full_bf16):
    """At the headline size (256 sequences x 512 tokens, the folded-LayerNorm kernels): two forwards of the same
    batch are bit-identical (the fold's row sums are integer atomics, nothing else accumulates across
    workgroups), and a permutation of the sequences permutes the outputs bit for bit (no result depends on a
    row's place in a tile or on its neighbours); padded rows stay out of it."""
    rng = np.random.default_rng(23)
    B, T = 256, 512
    tok = np.full((B, T), O.PAD, np.int64)
    for b in range(B):
        n = T if b % 3 else int(rng.integers(40, T))
        tok[b, :n] = [O.CLS] + list(rng.choice([5, 6, 7, 8, 9], n - 2)) + [O.EOS]
    sp = rng.integers(0, 1438, B)
    perm = rng.permutation(B)
    tok_d, sp_d = torch.from_numpy(tok).to(DEV), torch.from_numpy(sp).to(DEV)
    perm_d = torch.from_numpy(perm).to(DEV)
    with torch.no_grad():
        a = full_bf16(tok_d, sp_d)
        logits_a, rep_a = a["logits"].clone(), a["representations"][12].clone()
        b = full_bf16(tok_d, sp_d)
        assert torch.equal(b["logits"], logits_a) and torch.equal(b["representations"][12], rep_a)
        c = full_bf16(tok_d[perm_d], sp_d[perm_d])
    keep = (tok_d[perm_d] != O.PAD)
    assert torch.equal(c["logits"][keep], logits_a[perm_d][keep])
    assert torch.equal(c["representations"][12][keep], rep_a[perm_d][keep])
    assert torch.isfinite(logits_a[tok_d != O.PAD]).all()
