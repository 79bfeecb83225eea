#!/usr/bin/env python3
"""Golden vectors for the evaluation metrics, produced by the REFERENCE's own functions
(evaluation/gc.py, cfd.py, minmax.py, codon_recover_rate.py) imported from /root/reference.

Build-container only.  The modules import the `CAI` package, `tqdm` and `dtw` at the top (absent
offline, unused by the functions called here): import-only stand-ins are registered first.

    python tools/gen_golden_eval.py        # writes tests/golden/evaluation.json
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference/evaluation"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "evaluation.json")


def _load(name):
    spec = importlib.util.spec_from_file_location("ref_eval_" + name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    sys.dont_write_bytecode = True
    cai = types.ModuleType("CAI")
    cai.CAI = cai.relative_adaptiveness = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("CAI stand-in"))
    sys.modules.setdefault("CAI", cai)
    if "tqdm" not in sys.modules:
        try:
            import tqdm  # noqa: F401
        except ImportError:
            t = types.ModuleType("tqdm")
            t.tqdm = lambda x, **k: x
            sys.modules["tqdm"] = t
    gc, cfd, minmax = _load("gc"), _load("cfd"), _load("minmax")
    # codon_recover_rate.py runs a script at import; its calculate_similarity is nested in no function
    # but the module body needs files: take the function's source object by compiling only its def
    src = open(os.path.join(REF, "codon_recover_rate.py")).read()
    start = src.index("def calculate_similarity")
    body = src[start:]
    end = body.index("\n\n\n") if "\n\n\n" in body else len(body)
    ns = {}
    import textwrap
    exec(compile(textwrap.dedent(body[:end]), "codon_recover_rate.calculate_similarity", "exec"), ns)
    similarity = ns["calculate_similarity"]

    rng = np.random.default_rng(2024)
    table = {
        "F": ["TTT", "TTC"], "L": ["TTA", "TTG", "CTT", "CTC", "CTA", "CTG"], "I": ["ATT", "ATC", "ATA"], "M": ["ATG"],
        "V": ["GTT", "GTC", "GTA", "GTG"], "S": ["TCT", "TCC", "TCA", "TCG", "AGT", "AGC"], "P": ["CCT", "CCC", "CCA", "CCG"],
        "T": ["ACT", "ACC", "ACA", "ACG"], "A": ["GCT", "GCC", "GCA", "GCG"], "Y": ["TAT", "TAC"], "_": ["TAA", "TAG", "TGA"],
        "H": ["CAT", "CAC"], "Q": ["CAA", "CAG"], "N": ["AAT", "AAC"], "K": ["AAA", "AAG"], "D": ["GAT", "GAC"],
        "E": ["GAA", "GAG"], "C": ["TGT", "TGC"], "W": ["TGG"], "R": ["CGT", "CGC", "CGA", "CGG", "AGA", "AGG"],
        "G": ["GGT", "GGC", "GGA", "GGG"]}
    freqs = {}
    for aa, codons in table.items():
        f = rng.random(len(codons)) + 0.05
        f = f / f.sum()
        freqs[aa] = (codons, [float(x) for x in f])
    all_codons = [c for cs in table.values() for c in cs]
    cases = []
    for n in (1, 5, 18, 19, 40, 333):
        a = "".join(rng.choice(all_codons, n))
        b = "".join(rng.choice(all_codons, max(1, n - int(rng.integers(0, 3)))))
        b = "".join(a[i:i + 3] if rng.random() < 0.6 and i < len(a) else b[i:i + 3] for i in range(0, len(b), 3))
        mm = minmax.get_min_max_percentage(a, freqs, window_size=18)
        cases.append({"dna": a, "other": b, "gc": gc.get_GC_content(a), "cfd": cfd.get_cfd(a, freqs, threshold=0.3),
                      "cfd_05": cfd.get_cfd(a, freqs, threshold=0.5), "minmax": mm,
                      "minmax_w5": minmax.get_min_max_percentage(a, freqs, window_size=5),
                      "similarity": similarity(a, b)})
    with open(OUT, "w") as fh:
        json.dump({"codon_frequencies": freqs, "cases": cases}, fh, indent=0)
    print("wrote", OUT, len(cases), "cases")


if __name__ == "__main__":
    main()
