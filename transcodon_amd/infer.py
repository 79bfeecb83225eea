"""Command-line codon optimisation: the driver of the reference's infer.py:229-302.

    python -m transcodon_amd.infer --model_checkpoint model.safetensors --input_data in.csv \\
        --output_file predictions.csv [--with_utr] [--batch_size N]

Same flags, same CSV contract (input columns ``protein_seq``, ``organism`` and, with
--with_utr, ``utr``; output columns ``organism, protein_seq, prediction_dna`` with U written
as T).  Instead of one forward per row (infer.py:261-286) the rows are length-bucketed into
padded batches; ``--batch_size`` bounds the sequences per batch (the reference accepts the
flag and ignores it).
"""

import argparse
import os
import sys

import pandas as pd
import torch

from . import checkpoint
from .alphabet import Alphabet
from .infer_utr import optimize_codons
from .model import ProteinBertModel


def build_parser():
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--model_checkpoint", type=str, default="./Codon-AA-run/latest-120000.ckpt",
                   help="Path to the trained model checkpoint (.safetensors or .ckpt)")
    p.add_argument("--input_data", type=str, default="input.csv", help="Path to the input data for inference")
    p.add_argument("--batch_size", type=int, default=0,
                   help="Most sequences per padded batch (0: bounded by --max_batch_tokens only)")
    p.add_argument("--output_file", type=str, default="predictions.csv", help="File to save the predictions")
    p.add_argument("--with_utr", action="store_true", help="add utr sequence for infer")
    p.add_argument("--max_batch_tokens", type=int, default=1 << 17, help="Padded tokens per forward")
    p.add_argument("--precision", choices=("fp32", "bf16"), default="fp32",
                   help="GEMM/attention operand type (the reference runs fp32)")
    p.add_argument("--device", type=str, default="cuda")
    ProteinBertModel.add_args(p)
    return p


def run(args):
    alphabet = Alphabet.from_architecture("CodonModel")
    model = ProteinBertModel(args, alphabet)
    checkpoint.load_into(model, args.model_checkpoint)
    model = model.to(args.device).eval()
    model.set_compute_dtype(torch.bfloat16 if args.precision == "bf16" else torch.float32)
    data = pd.read_csv(args.input_data)
    for col in ("protein_seq", "organism") + (("utr",) if args.with_utr else ()):
        if col not in data.columns:
            raise KeyError(f"{args.input_data}: column '{col}' is missing")
    proteins = [str(s) for s in data["protein_seq"]]
    organisms = list(data["organism"])
    utrs = [str(u) for u in data["utr"]] if args.with_utr else None
    if args.batch_size and args.batch_size > 0:
        rnas = []
        # keep the length bucketing inside each group of batch_size rows of the sorted order
        order = sorted(range(len(proteins)), key=lambda i: len(proteins[i]) + (len(utrs[i]) if utrs else 0))
        out = [None] * len(proteins)
        for b in range(0, len(order), args.batch_size):
            idx = order[b:b + args.batch_size]
            res = optimize_codons(model, [proteins[i] for i in idx], [organisms[i] for i in idx],
                                  None if utrs is None else [utrs[i] for i in idx],
                                  max_batch_tokens=args.max_batch_tokens)
            for i, r in zip(idx, res):
                out[i] = r
        rnas = out
    else:
        rnas = optimize_codons(model, proteins, organisms, utrs, max_batch_tokens=args.max_batch_tokens)
    predict_data = pd.DataFrame({
        "organism": data["organism"],
        "protein_seq": data["protein_seq"],
        "prediction_dna": [r.replace("U", "T") for r in rnas],
    })
    directory = os.path.dirname(args.output_file)
    if directory:                      # a bare file name is fine (the reference fails on it)
        os.makedirs(directory, exist_ok=True)
    predict_data.to_csv(args.output_file, index=False)
    return predict_data


def main(argv=None):
    run(build_parser().parse_args(argv))


if __name__ == "__main__":
    main(sys.argv[1:])
