#!/usr/bin/env python3
"""Fixtures on TRAINED-LIKE weights, from the reference's own ProteinBertModel.

Build-container only (imports /root/reference through the stand-ins of tools/gen_golden.py; nothing of
the reference is copied).  oracle.weights.trained_like_state_dict gives weights with the features of a
trained checkpoint (LayerNorm gains / shifts away from 1 / 0, peaked attention, outlier channels in the
residual stream, a non-zero row mean, wide pre-GELU range); the reference model
(transcodon/model.py:175-259, modules.py:83-105) runs on them in fp32 and, as a yardstick for what bf16
arithmetic costs on these weights, once more under torch.autocast("cpu", torch.bfloat16).  Stored:
inputs, fp32 logits / structure logits / a channel subsample of representations[12] / decode margins, the
autocast run's logits, and statistics of the regime (residual magnitudes, |mean|/std of the rows that enter
the LayerNorms, score spread, pre-GELU range) gathered with forward hooks.

    python tools/gen_golden_trained.py        # writes tests/golden/trained_like_{tiny,full,full_shifted}.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402

REP_STRIDE = 8


def regime_stats(model, tokens, species):
    """What the LayerNorm inputs, attention scores and GELU inputs look like on these weights."""
    st = {"ln_in_absmax": 0.0, "ln_in_mean_over_std": [], "pre_gelu_absmax": 0.0, "score_std": []}
    hooks = []

    def ln_hook(mod, inp):
        x = inp[0].detach().float()
        st["ln_in_absmax"] = max(st["ln_in_absmax"], float(x.abs().max()))
        live = x.std(-1) > 0                                    # padded rows are all zero
        st["ln_in_mean_over_std"].append(float((x.mean(-1).abs() / x.std(-1))[live].mean()))

    def fc1_hook(mod, inp, out):
        st["pre_gelu_absmax"] = max(st["pre_gelu_absmax"], float(out.detach().abs().max()))

    for l in model.layers:
        hooks.append(l.self_attn_layer_norm.register_forward_pre_hook(ln_hook))
        hooks.append(l.final_layer_norm.register_forward_pre_hook(ln_hook))
        hooks.append(l.fc1.register_forward_hook(fc1_hook))
    out = model(tokens, species, need_head_weights=True)
    att = out["attentions"]                      # pre-softmax scores [B, L, H, T, T], 0 at padding pairs
    for l in range(att.shape[1]):
        a = att[0, l]
        st["score_std"].append(float(a[torch.isfinite(a)].std()))
    for h in hooks:
        h.remove()
    return {"ln_in_absmax": st["ln_in_absmax"], "ln_in_mean_over_std": float(np.mean(st["ln_in_mean_over_std"])),
            "ln_in_mean_over_std_max": float(np.max(st["ln_in_mean_over_std"])),
            "pre_gelu_absmax": st["pre_gelu_absmax"], "score_std_mean": float(np.mean(st["score_std"])),
            "score_std_max": float(np.max(st["score_std"]))}


def make(cfg, name, seed, tokens, species, aa_rows, O, trained_like_state_dict, common_offset=0.12):
    sd = trained_like_state_dict(cfg, seed, common_offset)
    m = G.ref_model(cfg, sd)
    t, s = torch.from_numpy(tokens), torch.from_numpy(species)
    out = m(t, s, repr_layers=[cfg.num_layers])
    with torch.autocast("cpu", dtype=torch.bfloat16):
        low = m(t, s, repr_layers=[cfg.num_layers])
    stats = regime_stats(m, t, s)
    o = O.forward(sd, cfg, t, s)
    keep = tokens != O.PAD
    print(name, "oracle-vs-reference max|dlogit|", float((o["logits"] - out["logits"]).abs()[torch.from_numpy(keep)].max()))
    logits = out["logits"].float().numpy()
    low_logits = low["logits"].float().numpy()
    sig = logits.std(-1, keepdims=True)
    rel = (np.abs(low_logits - logits) / sig)[keep]
    print(name, "autocast yardstick: rms", float(np.sqrt((rel ** 2).mean())), "max", float(rel.max()),
          "| logit std per position", float(sig[keep].mean()), "| regime", stats)
    margins = {}
    for r, (start, aa) in aa_rows.items():
        _, mg = O.decode(out["logits"][r, start:start + 3 * len(aa)], aa, return_margins=True)
        margins[f"margins_r{r}"] = np.array(mg, np.float64)
        margins[f"rna_r{r}"] = np.array(O.decode(out["logits"][r, start:start + 3 * len(aa)], aa))
        margins[f"aa_r{r}"] = np.array(aa)
        margins[f"start_r{r}"] = np.array(start)
    np.savez_compressed(
        os.path.join(G.OUT, f"trained_like_{name}.npz"), seed=seed, common_offset=common_offset, tokens=tokens, species=species,
        logits=logits, struct=out["struct_label_output"].float().numpy(),
        rep_last=out["representations"][cfg.num_layers].float().numpy()[..., ::REP_STRIDE], rep_stride=REP_STRIDE,
        autocast_logits=low_logits, autocast_struct=low["struct_label_output"].float().numpy(),
        **{"stat_" + k: np.array(v) for k, v in stats.items()}, **margins)


def main():
    os.makedirs(G.OUT, exist_ok=True)
    os.chdir(G.REF)
    sys.path.insert(0, G.REF)
    G.install_shims()
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    from oracle import transcodon_ref as O
    from oracle.weights import trained_like_state_dict

    def rows(T, lens, rng):
        tok = np.full((len(lens), T), O.PAD, np.int64)
        aa_rows = {}
        for b, n in enumerate(lens):
            m = (n - 2) // 3
            prot = "".join(rng.choice(list("ACDEFGHIKLMNPQRSTVWY"), m - 1)) + "_"
            tmpl = O.aa_tokenize(prot)
            utr = n - 2 - 3 * m                         # 0..2 leftover positions become a short 5'UTR + <sep>
            body = [O.encode([c])[0] if c != "*" else 9 for c in tmpl]
            if utr >= 1:
                row = [O.CLS] + list(rng.integers(5, 9, utr - 1)) + [O.SEP] + body + [O.EOS]
                start = 1 + utr
            else:
                row = [O.CLS] + body + [O.EOS]
                start = 1
            assert len(row) == n
            tok[b, :n] = row
            aa_rows[b] = (start, prot)
        return tok, aa_rows

    rng = np.random.default_rng(2024)
    tok, aa_rows = rows(48, [48, 35, 20, 41], rng)
    make(O.TINY, "tiny", 21, tok, np.array([0, 5, 15, 9], np.int64), aa_rows, O, trained_like_state_dict)
    tok, aa_rows = rows(256, [256, 256, 181], rng)
    make(O.FULL, "full", 22, tok, np.array([3, 1200, 77], np.int64), aa_rows, O, trained_like_state_dict)
    # the same with a five times larger common offset: rows whose mean is of the order of their spread
    make(O.FULL, "full_shifted", 22, tok, np.array([3, 1200, 77], np.int64), aa_rows, O, trained_like_state_dict, 0.6)


if __name__ == "__main__":
    main()
