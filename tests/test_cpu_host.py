"""Host-side logic and the C-ABI surface, no GPU needed."""
import ctypes
import json
import os
import re

import pytest
import torch

import transcodon_amd as ta
from oracle import transcodon_ref as O
from transcodon_amd import _lib, infer_utr, ops
from transcodon_amd.sequence import (CodonSequence, CodonSequence_infer, WithUtr_CodonSequence,
                                     WithUtr_CodonSequence_infer)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "transcodon_hip.h")).read()
    declared = set(re.findall(r"\b(tc_[a-z0-9_]+)\s*\(", header))
    declared -= {"tc_layer_weights", "tc_encoder_desc"}
    assert len(declared) >= 15
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert set(_lib.EXPORTED_SYMBOLS) == declared
    assert _lib.load().tc_version() == _lib.ABI_VERSION


def test_alphabet_matches_reference_facts():
    a = ta.Alphabet.from_architecture("CodonModel")
    assert a.all_toks == O.ALL_TOKS
    assert (a.padding_idx, a.mask_idx, a.cls_idx, a.eos_idx, a.unk_idx) == (1, 9, 0, 2, 3)
    assert not a.prepend_bos and not a.append_eos
    assert len(ta.ORGANISM2ID) == 1438 and ta.ORGANISM2ID["Arabidopsis thaliana"] == 0
    assert sorted(ta.ORGANISM2ID.values()) == list(range(1438))
    with pytest.raises(ValueError):
        ta.Alphabet.from_architecture("nope")


def test_tokeniser_against_golden_rows(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "full_testcsv_decode.json")))
    import csv
    for r in g["rows"]:
        aa_like = "".join(O.CODON2AA[r["noutr"]["rna"][i:i + 3]]
                          for i in range(0, len(r["noutr"]["rna"]), 3)).replace("#", "_")
        assert infer_utr.AA_tokenize(aa_like) == r["template"] == O.aa_tokenize(aa_like)
        ids = infer_utr.tokenize(CodonSequence_infer(r["template"]))[0].tolist()
        assert ids == O.encode(O.token_strings("false", r["template"]))
        assert len(ids) == r["noutr"]["T"]
    utr = "ACGTTGCA AC"
    ids = infer_utr.tokenize(WithUtr_CodonSequence_infer(utr, "AU*GC*"))[0].tolist()
    assert ids == O.encode(O.token_strings(utr, "AU*GC*"))
    assert ids == [0, 5, 7, 8, 6, 6, 8, 7, 5, 5, 7, 4, 5, 6, 9, 8, 7, 9, 2]
    assert WithUtr_CodonSequence("AT", "GC").seq == "<cls> A U <sep> G C <eos>"
    assert CodonSequence("ignored", "ATG").tokens == ["<cls>", "A", "U", "G", "<eos>"]
    with pytest.raises(KeyError):
        infer_utr.AA_tokenize("MXK_")
    with pytest.raises(KeyError):
        ta.Alphabet.from_architecture("CodonModel").encode("A N G")


def test_batch_converter_pads_right():
    bc = ta.Alphabet.from_architecture("CodonModel").get_batch_converter()
    _, strs, toks = bc([("a", "<cls> A U <eos>"), ("b", "<cls> G <mask> C A <eos>")])
    assert toks.dtype == torch.int64 and toks.shape == (2, 6)
    assert toks[0].tolist() == [0, 5, 6, 2, 1, 1]
    assert toks[1].tolist() == [0, 8, 9, 7, 5, 2]


def test_amino_acid_codes_follow_decode_table():
    assert ops.AA_ORDER == "".join(O.AA2CODONS.keys())
    assert ops.encode_amino_acids("FG_") == bytes([0, 20, 9])


def test_state_dict_keys_match_reference(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "state_dict_keys.json")))
    m = ta.ProteinBertModel(ta.default_args(num_layers=2, embed_dim=128, ffn_embed_dim=512,
                                            attention_heads=2, species_vocab_size=16),
                            ta.Alphabet.from_architecture("CodonModel"))
    keys = set(m.state_dict())
    want = {k for k in ref if not k.startswith("layers.") or k.split(".")[1] in ("0", "1")}
    assert keys == want
    assert m.lm_head.weight is m.embed_tokens.weight


def test_no_cpu_fallback():
    m = ta.ProteinBertModel(ta.default_args(num_layers=1, embed_dim=128, ffn_embed_dim=512,
                                            attention_heads=2, species_vocab_size=4),
                            ta.Alphabet.from_architecture("CodonModel"))
    with pytest.raises(_lib.HipLibraryError):
        with torch.no_grad():
            m(torch.zeros(1, 8, dtype=torch.long), torch.zeros(1, dtype=torch.long))
    with pytest.raises(_lib.HipLibraryError):
        ops.layernorm(torch.zeros(4, 128), torch.ones(128), torch.zeros(128))


def test_argument_checks_without_a_gpu():
    lib = _lib.load()
    rc = lib.tc_gemm(1, 0, 16, 64, 16, 16, None, 16, 100, 4, 100, 64, 0.0, None, None, 0, 0, None)
    assert rc == -1 and b"multiple of 128" in lib.tc_last_error()
    rc = lib.tc_attn_fwd(1, 16, 384, None, 16, 1, 8, 2, 32, None)
    assert rc == -1 and b"head_dim" in lib.tc_last_error()
    rc = lib.tc_gemm(7, 0, 16, 64, 16, 16, None, 16, 128, 4, 128, 64, 0.0, None, None, 0, 0, None)
    assert rc == -1


def _tiny_model(seed=3):
    cfg = O.TINY
    args = ta.default_args(num_layers=cfg.num_layers, embed_dim=cfg.embed_dim, ffn_embed_dim=cfg.ffn_embed_dim,
                           attention_heads=cfg.attention_heads, species_vocab_size=cfg.species_vocab_size)
    m = ta.ProteinBertModel(args, ta.Alphabet.from_architecture("CodonModel"))
    return ta.init_weights(m, seed=seed)


@pytest.mark.parametrize("ext", [".safetensors", ".ckpt"])
def test_checkpoint_round_trip_with_reference_prefix(tmp_path, ext):
    """Files carry the 'model.' prefix of the reference's CodonModel wrapper and every key of its
    state_dict, the tied lm_head.weight included (infer.py:248-250 loads with strict=True)."""
    from transcodon_amd import checkpoint
    a, b = _tiny_model(3), _tiny_model(4)
    path = str(tmp_path / ("w" + ext))
    checkpoint.save(a, path, step=7)
    raw = checkpoint.read_state_dict(path)
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_keys.json")))
    want = {"model." + k for k in ref if not k.startswith("layers.") or k.split(".")[1] in ("0", "1")}
    assert set(raw) == want
    assert torch.equal(raw["model.lm_head.weight"], raw["model.embed_tokens.weight"])
    checkpoint.load_into(b, path)
    for (k, v), (_, w) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(v, w), k
    assert b.lm_head.weight is b.embed_tokens.weight


def test_checkpoint_without_tied_alias(tmp_path):
    """A safetensors file whose writer dropped the tied alias still loads."""
    from safetensors.torch import save_file
    from transcodon_amd import checkpoint
    a, b = _tiny_model(3), _tiny_model(4)
    sd = {"model." + k: v.detach().clone() for k, v in a.state_dict().items() if k != "lm_head.weight"}
    path = str(tmp_path / "noalias.safetensors")
    save_file(sd, path)
    checkpoint.load_into(b, path)
    assert torch.equal(a.embed_tokens.weight, b.lm_head.weight)


def test_checkpoint_without_prefix_and_bad_file(tmp_path):
    from transcodon_amd import checkpoint
    a, b = _tiny_model(5), _tiny_model(6)
    path = str(tmp_path / "plain.pt")
    torch.save(a.state_dict(), path)
    checkpoint.load_into(b, path)
    assert torch.equal(a.layers[1].fc2.weight, b.layers[1].fc2.weight)
    torch.save({"something": 3}, str(tmp_path / "bad.pt"))
    with pytest.raises(ValueError):
        checkpoint.read_state_dict(str(tmp_path / "bad.pt"))


def test_cli_parser_mirrors_reference_flags():
    from transcodon_amd import infer
    a = infer.build_parser().parse_args(["--model_checkpoint", "m.safetensors", "--input_data", "i.csv",
                                         "--output_file", "o.csv", "--with_utr", "--batch_size", "6"])
    assert (a.model_checkpoint, a.input_data, a.output_file, a.with_utr, a.batch_size) == \
        ("m.safetensors", "i.csv", "o.csv", True, 6)
    assert (a.num_layers, a.embed_dim, a.ffn_embed_dim, a.attention_heads, a.species_vocab_size) == \
        (12, 768, 3072, 12, 1438)


def test_evaluation_metrics_against_reference_outputs(golden_dir):
    """tests/golden/evaluation.json holds the outputs of the reference's own evaluation functions
    (tools/gen_golden_eval.py) on random coding sequences and a random codon-usage table."""
    from transcodon_amd import evaluation as EV
    with open(os.path.join(golden_dir, "evaluation.json")) as fh:
        g = json.load(fh)
    usage = EV.CodonUsage({k: (v[0], v[1]) for k, v in g["codon_frequencies"].items()})
    for c in g["cases"]:
        assert EV.gc_content(c["dna"]) == pytest.approx(c["gc"], rel=1e-12)
        assert EV.codon_similarity(c["dna"], c["other"]) == pytest.approx(c["similarity"], rel=1e-12)
        assert EV.cfd(c["dna"], usage, 0.3) == pytest.approx(c["cfd"], rel=1e-12)
        assert EV.cfd(c["dna"], usage, 0.5) == pytest.approx(c["cfd_05"], rel=1e-12)
        for key, w in (("minmax", 18), ("minmax_w5", 5)):
            got = EV.min_max_percentage(c["dna"], usage, w)
            assert len(got) == len(c[key])
            for a, b in zip(got, c[key]):
                assert (a is None) == (b is None)
                if a is not None:
                    assert a == pytest.approx(b, rel=1e-8, abs=1e-8)
    assert EV.gc_content("ACGU".replace("U", "T")) == 50.0
    with pytest.raises(KeyError):
        EV.cfd("ATGNNN", usage)


def test_bench_launcher_starts_one_rank_per_gpu():
    """`bench.py --gpus N` without WORLD_SIZE must itself start N ranks (before any GPU call) with the
    rendezvous variables set; --dry-run makes every rank print its identity and exit."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    for extra in ([], ["--config", "4"], ["--config", "3"]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"] + extra,
                           env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
        assert sorted(l["rank"] for l in lines) == [0, 1]
        assert all(l["world"] == 2 and l["local_rank"] == l["rank"] for l in lines)
        assert len({l["master"] for l in lines}) == 1 and lines[0]["master"].startswith("127.0.0.1:")
    # under torchrun (WORLD_SIZE set) the process is one rank and starts nothing
    env.update(WORLD_SIZE="4", RANK="3", LOCAL_RANK="3", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=300)
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["rank"] == 3 and lines[0]["world"] == 4


def test_array_tokenisation_equals_the_string_path():
    """template_token_ids / cds_token_ids (byte lookup tables, used by the batched drivers) against the
    string path they replace: AA_tokenize -> CodonSequence_infer / WithUtr_CodonSequence_infer ->
    Alphabet.encode (infer.py:113-129, sequence.py:60-79), including truncation, T -> U, blanks and the
    KeyError on foreign letters."""
    import numpy as np
    from transcodon_amd import infer_utr as I
    from transcodon_amd.sequence import CodonSequence
    rng = np.random.default_rng(5)
    aas = "FLIMVPTAYHQNKDECWRSG"
    for trial in range(40):
        prot = "".join(rng.choice(list(aas), int(rng.integers(1, 60))))
        utr = None if trial % 2 else "".join(rng.choice(list("ACGTU "), int(rng.integers(0, 30))))
        max_aa = [None, 2048, 7][trial % 3]
        aa = prot + "_"
        tmpl = I.AA_tokenize(aa)
        want_aa = aa
        if max_aa is not None and len(tmpl) > 3 * max_aa:
            want_aa, tmpl = aa[:max_aa], tmpl[:3 * max_aa]
        seq = I.CodonSequence_infer(tmpl) if utr is None else I.WithUtr_CodonSequence_infer(utr, tmpl)
        want = I._ALPHABET.encode(seq.seq)
        ids, start, got_aa = I.template_token_ids(aa, utr, max_aa)
        assert ids.dtype == np.int32 and ids.tolist() == want
        assert got_aa == want_aa
        assert start == 1 + (0 if utr is None else len(utr.replace(" ", "")) + 1)
        cds = "".join(rng.choice(list("ACGTU"), 3 * int(rng.integers(1, 40))))
        assert I.cds_token_ids(cds).tolist() == I._ALPHABET.encode(CodonSequence("", cds).seq)
    for bad in ("MKX_", "MB_", "Mk_"):
        with pytest.raises(KeyError):
            I.AA_tokenize(bad)
        with pytest.raises(KeyError):
            I.template_token_ids(bad)
    with pytest.raises(KeyError):
        I.template_token_ids("MK_", "ACGN")
    with pytest.raises(KeyError):
        I.cds_token_ids("ACGN")


def test_batched_back_translation_check_equals_the_string_one():
    import numpy as np
    from transcodon_amd import infer_utr as I
    from transcodon_amd.sequence import CODON_TO_AA
    rng = np.random.default_rng(8)
    by_aa = {}
    for codon, aa in CODON_TO_AA.items():
        by_aa.setdefault("_" if aa == "#" else aa, []).append(codon)
    prots = ["".join(rng.choice(list("FLIMVPTAYHQNKDECWRSG"), int(rng.integers(2, 30)))) + "_" for _ in range(12)]
    rnas = ["".join(by_aa[a][int(rng.integers(len(by_aa[a])))] for a in p) for p in prots]
    offs = np.concatenate([[0], np.cumsum([len(p) for p in prots])])
    letters = np.frombuffer("".join(prots).encode(), np.uint8)
    flat = np.frombuffer("".join(rnas).encode(), np.uint8)
    assert all(I.back_translation_ok(r, p) for r, p in zip(rnas, prots))
    assert I.back_translation_mismatch(flat, letters, offs) == -1
    # a wrong last codon is not looked at by either form; a wrong inner codon is found by both
    broken = list(rnas)
    broken[4] = broken[4][:-3] + "AUG"
    assert I.back_translation_ok(broken[4], prots[4])
    assert I.back_translation_mismatch(np.frombuffer("".join(broken).encode(), np.uint8), letters, offs) == -1
    wrong = "UGG" if prots[7][1] != "W" else "AUG"
    broken[7] = broken[7][:3] + wrong + broken[7][6:]
    assert not I.back_translation_ok(broken[7], prots[7])
    assert I.back_translation_mismatch(np.frombuffer("".join(broken).encode(), np.uint8), letters, offs) == 7
