#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own modules.

Build-container only (needs /root/reference; never runs on the GPU box).
The reference's modules are imported unmodified from /root/reference; nothing
of them is copied.  Packages the reference imports that are not installed
offline get in-memory stand-ins registered in sys.modules BEFORE the import:

  * Bio.Seq.Seq / Bio.Data.CodonTable.standard_dna_table -- only used in type
    annotations / an unused import (sequence.py:4, pipeline.py:11).
  * rotary_embedding_torch.RotaryEmbedding -- restated from the published
    algorithm of rotary-embedding-torch==0.8.6 (environment.yml:205).  This is
    why parity is "unpinned" at the RoPE boundary.
  * pytorch_lightning / dtw -- only so that the top-level infer.py can be
    imported for its AA_tokenize and _split_into_codons helpers.

Outputs are numeric / string fixtures only.

    python tools/gen_golden.py            # writes tests/golden/
"""

import argparse
import csv
import json
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True


def install_shims():
    bio = types.ModuleType("Bio")
    bio_seq = types.ModuleType("Bio.Seq")
    bio_seq.Seq = str
    bio_data = types.ModuleType("Bio.Data")
    bio_ct = types.ModuleType("Bio.Data.CodonTable")
    bio_ct.standard_dna_table = None
    bio_io = types.ModuleType("Bio.SeqIO")
    bio_rec = types.ModuleType("Bio.SeqRecord")
    bio_rec.SeqRecord = object
    sys.modules["Bio.SeqRecord"] = bio_rec
    bio.Seq, bio.Data, bio_data.CodonTable, bio.SeqIO = bio_seq, bio_data, bio_ct, bio_io
    sys.modules.update({"Bio": bio, "Bio.Seq": bio_seq, "Bio.Data": bio_data,
                        "Bio.Data.CodonTable": bio_ct, "Bio.SeqIO": bio_io})

    rot = types.ModuleType("rotary_embedding_torch")

    class RotaryEmbedding(torch.nn.Module):
        def __init__(self, dim, theta=10000):
            super().__init__()
            freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
            self.freqs = torch.nn.Parameter(freqs, requires_grad=False)

        def rotate_queries_or_keys(self, t, seq_dim=-2):
            n = t.shape[seq_dim]
            pos = torch.arange(n, device=t.device, dtype=t.dtype)
            f = torch.einsum("..., f -> ... f", pos.type(self.freqs.dtype), self.freqs)
            f = torch.repeat_interleave(f, 2, dim=-1)
            x1, x2 = t[..., 0::2], t[..., 1::2]
            rh = torch.stack((-x2, x1), dim=-1).reshape(t.shape)
            return (t * f.cos() + rh * f.sin()).type(t.dtype)

    rot.RotaryEmbedding = RotaryEmbedding
    sys.modules["rotary_embedding_torch"] = rot


def install_cli_shims():
    class _AnyAttr(types.ModuleType):
        def __getattr__(self, name):          # Trainer, Callback, ... -> placeholder type
            if name.startswith("__"):
                raise AttributeError(name)
            return object

    pl = _AnyAttr("pytorch_lightning")
    pl.LightningModule = torch.nn.Module
    for sub, names in {"loggers": ["WandbLogger", "TensorBoardLogger"],
                       "callbacks": ["LearningRateMonitor", "Callback", "ModelCheckpoint"],
                       "strategies": ["DDPStrategy"]}.items():
        m = types.ModuleType("pytorch_lightning." + sub)
        for n in names:
            setattr(m, n, object)
        setattr(pl, sub, m)
        sys.modules["pytorch_lightning." + sub] = m
    sys.modules["pytorch_lightning"] = pl
    dtw = types.ModuleType("dtw")
    dtw.accelerated_dtw = None
    sys.modules["dtw"] = dtw


def ref_model(cfg, sd):
    from transcodon.alphabet import Alphabet
    from transcodon.model import ProteinBertModel
    args = argparse.Namespace(
        num_layers=cfg.num_layers, embed_dim=cfg.embed_dim, attention_dropout=0.0,
        logit_bias=False, rope_embedding=True, ffn_embed_dim=cfg.ffn_embed_dim,
        attention_heads=cfg.attention_heads, species_embedding=True, struct_label=True,
        species_vocab_size=cfg.species_vocab_size, max_positions=2048)
    alphabet = Alphabet.from_architecture("CodonModel")
    m = ProteinBertModel(args, alphabet)
    missing = m.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m.eval()


def main():
    os.makedirs(OUT, exist_ok=True)
    os.chdir(REF)                       # alphabet.py:123 opens a cwd-relative json
    sys.path.insert(0, REF)
    install_shims()
    torch.manual_seed(0)
    torch.set_grad_enabled(False)

    from oracle import transcodon_ref as O
    from oracle.weights import seeded_state_dict, state_dict_spec
    from transcodon.alphabet import Alphabet, ORGANISM2ID
    from transcodon.sequence import CodonSequence_infer, WithUtr_CodonSequence_infer
    from transcodon import infer_utr

    # ---- G0: state_dict key/shape inventory of the reference at full size ----
    full_sd = seeded_state_dict(O.FULL, seed=7)
    m_full = ref_model(O.FULL, full_sd)
    ref_keys = {k: tuple(v.shape) for k, v in m_full.state_dict().items()}
    mine = {n: s for n, s, _ in state_dict_spec(O.FULL)}
    assert ref_keys == mine, "state_dict spec differs from the reference"
    with open(os.path.join(OUT, "state_dict_keys.json"), "w") as f:
        json.dump({k: list(v) for k, v in ref_keys.items()}, f, indent=0)
    n_params = sum(p.numel() for p in m_full.parameters())
    print("full model params", n_params)

    # ---- G1: tiny config, perturbed weights, padded batch with UTR/sep/mask ----
    tiny_sd = seeded_state_dict(O.TINY, seed=11, perturb=True)
    m_tiny = ref_model(O.TINY, tiny_sd)
    rng = np.random.default_rng(123)
    B, T = 4, 48
    toks = np.full((B, T), O.PAD, np.int64)
    lens = [48, 33, 17, 40]
    for b, L in enumerate(lens):
        body = rng.choice([5, 6, 7, 8, 9], size=L - 2, p=[.2, .2, .2, .2, .2])
        row = np.concatenate([[O.CLS], body, [O.EOS]])
        if L > 12:
            row[6] = O.SEP
        toks[b, :L] = row
    species = np.array([0, 5, 15, 9], np.int64)
    out = m_tiny(torch.from_numpy(toks), torch.from_numpy(species), repr_layers=[0, 1, 2],
                 need_head_weights=True)
    labels = np.where(rng.random((B, T)) < 0.25, toks, -100)
    labels[toks == O.PAD] = -100
    slab = np.where(rng.random((B, T)) < 0.5, rng.integers(0, 3, (B, T)), -100)
    slab[toks == O.PAD] = -100
    ce = torch.nn.CrossEntropyLoss(reduction="mean", ignore_index=-100)
    loss_mlm = ce(out["logits"].view(-1, 10), torch.from_numpy(labels).view(-1))
    loss_2d = ce(out["struct_label_output"].view(-1, 3), torch.from_numpy(slab).view(-1))
    np.savez_compressed(
        os.path.join(OUT, "tiny_forward.npz"),
        seed=11, tokens=toks, species=species, lens=np.array(lens),
        logits=out["logits"].numpy(), struct=out["struct_label_output"].numpy(),
        rep0=out["representations"][0].numpy(), rep1=out["representations"][1].numpy(),
        rep2=out["representations"][2].numpy(), attentions=out["attentions"].numpy(),
        labels=labels, struct_labels=slab, loss_mlm=loss_mlm.numpy(),
        loss_total=(loss_mlm + 0.4 * loss_2d).numpy())
    # oracle cross-check while we are here
    o = O.forward(tiny_sd, O.TINY, torch.from_numpy(toks), torch.from_numpy(species),
                  repr_layers=[0, 1, 2], need_head_weights=True)
    print("tiny oracle-vs-ref max|dlogit|",
          float((o["logits"] - out["logits"]).abs().max()))

    # ---- G1b: tiny, no padding (padding_mask=None branch) ----
    toks2 = toks[:1].copy()
    out2 = m_tiny(torch.from_numpy(toks2), torch.from_numpy(species[:1]), repr_layers=[2])
    np.savez_compressed(os.path.join(OUT, "tiny_forward_nopad.npz"), seed=11, tokens=toks2,
                        species=species[:1], logits=out2["logits"].numpy(),
                        struct=out2["struct_label_output"].numpy(),
                        rep2=out2["representations"][2].numpy())

    # ---- G2: full config, test.csv rows through the reference's embed_sequence ----
    install_cli_shims()
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_infer_cli", os.path.join(REF, "infer.py"))
    cli = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(cli)
        AA_tokenize, backtr = cli.AA_tokenize, cli._split_into_codons
    except Exception as e:                                   # noqa: BLE001
        print("top-level infer.py not importable:", repr(e))
        AA_tokenize, backtr = None, None

    rows = list(csv.DictReader(open(os.path.join(REF, "test.csv"))))
    g2 = {"rows": []}
    all_margins = {}
    import contextlib
    import io
    for ri, row in enumerate(rows):
        aa = row["protein_seq"] + "_"
        tmpl_ref = AA_tokenize(aa) if AA_tokenize else None
        tmpl = O.aa_tokenize(aa)
        if tmpl_ref is not None:
            assert tmpl_ref == tmpl, "AA_tokenize restatement differs"
        if len(tmpl) > 2048 * 3:
            aa, tmpl = aa[:2048], tmpl[: 2048 * 3]
        org = ORGANISM2ID[row["organism"]]
        rec = {"row": ri, "organism": row["organism"], "organism_id": org,
               "protein_len": len(row["protein_seq"]), "template": tmpl, "utr_seq": row["utr"]}
        for with_utr in (False, True):
            utr = row["utr"] if with_utr else "false"
            seq = (CodonSequence_infer(tmpl) if not with_utr
                   else WithUtr_CodonSequence_infer(utr, tmpl))
            ids = infer_utr.tokenize(seq)[0].tolist()
            assert ids == O.encode(O.token_strings(utr, tmpl)), "tokeniser restatement differs"
            with contextlib.redirect_stdout(io.StringIO()):
                reps, logits, rna = infer_utr.embed_sequence(m_full, utr, tmpl, org, aa)
                if backtr is not None:
                    assert backtr(rna, aa)
            _, margins = O.decode(logits[0], aa, return_margins=True)
            o_rna = O.decode(logits[0], aa)
            assert o_rna == rna, "decode restatement differs"
            key = "utr" if with_utr else "noutr"
            rec[key] = {"T": len(ids), "rna": rna,
                        "min_margin": float(np.min(margins)),
                        "n_margin_lt_1e-4": int(np.sum(np.array(margins) < 1e-4))}
            all_margins[f"r{ri}_{key}"] = np.array(margins, np.float32)
            if ri in (2, 5):
                np.savez_compressed(
                    os.path.join(OUT, f"full_row{ri}_{key}.npz"), seed=7,
                    tokens=np.array(ids, np.int64), species=org,
                    logits=logits.numpy().astype(np.float32),
                    rep12_head=reps[12][0, :4, :].numpy(),
                    margins=np.array(margins, np.float64))
            print("row", ri, key, "T", len(ids), "min margin", rec[key]["min_margin"])
        g2["rows"].append(rec)
    with open(os.path.join(OUT, "full_testcsv_decode.json"), "w") as f:
        json.dump(g2, f, indent=1)
    # decision margin (log p_best - log p_second) of every amino acid of every row, both modes
    np.savez_compressed(os.path.join(OUT, "full_testcsv_margins.npz"), **all_margins)

    # ---- G2b: full config, padded batch of two rows (key-padding branch) ----
    r2 = O.encode(O.token_strings("false", g2["rows"][2]["template"]))
    r5 = O.encode(O.token_strings(rows[3]["utr"], g2["rows"][3]["template"]))[:200]
    tb = O.batch_tokens([r2, r5])
    sp = torch.tensor([g2["rows"][2]["organism_id"], g2["rows"][3]["organism_id"]])
    ob = m_full(tb, sp)
    np.savez_compressed(os.path.join(OUT, "full_padded_batch.npz"), seed=7, tokens=tb.numpy(),
                        species=sp.numpy(), logits=ob["logits"].numpy(),
                        struct=ob["struct_label_output"].numpy(),
                        rep12_b1_head=ob["representations"][12][1, :4, :].numpy())

    # ---- G3: decode known-answer tests through the reference's decode loop ----
    class Fake(torch.nn.Module):
        def __init__(self, logits):
            super().__init__()
            self.l = logits

        def forward(self, tokens, species):
            return {"logits": self.l, "representations": {}}

    kats = []
    aas = "FLIMVPTAYHQNKDECWRSG"
    rng = np.random.default_rng(5)

    def run_kat(name, aa, cds_logits):
        m = len(aa)
        full = torch.zeros(1, 3 * m + 2, 10)
        full[0, 1:-1] = cds_logits
        with contextlib.redirect_stdout(io.StringIO()):
            _, _, rna = infer_utr.embed_sequence(Fake(full), "false", "A" * (3 * m), 0, aa)
        assert O.decode(cds_logits, aa) == rna, name
        kats.append({"name": name, "aa": aa, "logits": cds_logits.numpy().tolist(), "rna": rna})

    aa = aas + "_"
    run_kat("all_equal_first_wins", aa, torch.zeros(3 * len(aa), 10))
    run_kat("random_normal", aa, torch.from_numpy(rng.standard_normal((3 * len(aa), 10)).astype(np.float32)))
    run_kat("random_peaked", aa, torch.from_numpy((8 * rng.standard_normal((3 * len(aa), 10))).astype(np.float32)))
    big = torch.from_numpy((50 * rng.standard_normal((3 * len(aa), 10))).astype(np.float32))
    run_kat("large_logits_underflow", aa, big)
    ninf = torch.from_numpy(rng.standard_normal((3 * len(aa), 10)).astype(np.float32))
    ninf[:, 5] = float("-inf")                    # P(A) == 0 everywhere
    run_kat("neg_inf_base_A", aa, ninf)
    sp_mass = torch.zeros(3 * len(aa), 10)
    sp_mass[:, :5] = 6.0                          # specials hog the mass, bases tie
    run_kat("special_tokens_mass", aa, sp_mass)
    lrs = "LLLLRRRRSSSS_"
    tie = torch.zeros(3 * len(lrs), 10)
    tie[0::3, 7] = 1.0                            # favour C at pos 1
    tie[2::3, 8] = 1.0                            # favour G at pos 3
    run_kat("six_fold_partial_ties", lrs, tie)
    with open(os.path.join(OUT, "decode_kat.json"), "w") as f:
        json.dump(kats, f)

    # ---- G5: LR schedule and weight initialisation from the reference's own pretraining.py ----
    # (CodonModel.__init__ / init_weights :40-58, configure_optimizers :64-92), imported with the
    # Lightning stand-ins above; LightningModule is torch.nn.Module plus two no-op methods.
    import pytorch_lightning as pl_shim

    class _LM(torch.nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

    pl_shim.LightningModule = _LM
    pl_shim.Callback = object
    spec = importlib.util.spec_from_file_location("ref_pretraining", os.path.join(REF, "pretraining.py"))
    pre = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pre)
    targs = argparse.Namespace(
        num_layers=O.TINY.num_layers, embed_dim=O.TINY.embed_dim, attention_dropout=0.0, logit_bias=False,
        rope_embedding=True, ffn_embed_dim=O.TINY.ffn_embed_dim, attention_heads=O.TINY.attention_heads,
        species_embedding=True, struct_label=True, species_vocab_size=O.TINY.species_vocab_size,
        max_positions=2048, learning_rate=2e-4, weight_decay=0.1, lr_scheduler="warmup_cosine",
        warmup_steps=10, num_steps=100)
    torch.manual_seed(3)
    with torch.enable_grad():
        cm = pre.CodonModel(targs, Alphabet.from_architecture("CodonModel"))
        (optim,), (sch,) = cm.configure_optimizers()
    lam = sch["scheduler"].lr_lambdas[0]
    steps = list(range(0, 120, 7)) + [9, 10, 11, 109, 110, 111, 500]
    sched = {"warmup": 10, "num_steps": 100, "steps": steps, "values": [float(lam(s)) for s in steps],
             "optimizer": {"class": type(optim).__name__, "lr": optim.defaults["lr"],
                           "weight_decay": optim.defaults["weight_decay"], "betas": list(optim.defaults["betas"]),
                           "eps": optim.defaults["eps"],
                           "n_param_groups": len(optim.param_groups),
                           "n_params": sum(p.numel() for g in optim.param_groups for p in g["params"])}}
    with open(os.path.join(OUT, "schedule.json"), "w") as f:
        json.dump(sched, f)
    init = {}
    for k, v in cm.model.state_dict().items():
        v = v.float()
        init[k] = {"shape": list(v.shape), "mean": float(v.mean()), "std": float(v.std()) if v.numel() > 1 else 0.0,
                   "all_zero": bool((v == 0).all()), "all_one": bool((v == 1).all())}
    with open(os.path.join(OUT, "init_weights_stats.json"), "w") as f:
        json.dump({"seed": 3, "config": "tiny", "tensors": init}, f, indent=0)
    print("wrote fixtures to", OUT)


if __name__ == "__main__":
    main()
