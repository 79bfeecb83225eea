#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSVs (gpurun_out/pmc_<tag>/*/…/*counter_collection.csv) per kernel.

    python tools/pmc_summary.py gpurun_out/pmc_r1 > profiles/r01_pmc_summary.txt

FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; on gfx950 FETCH_SIZE counts
128-byte requests as 64 bytes for wide coalesced reads, so fetched bytes = 2 * FETCH_SIZE KiB.
"""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(gemm_kernel\w*|attn_bf16_kernel|attn_f32_kernel|layernorm_kernel|head_kernel|embed_kernel|decode_kernel|key_mask_kernel)", name)
    if not m:
        return None
    base = m.group(1)
    if base.startswith("gemm_kernel"):
        e = re.search(r"Li(\d)E", name)
        base = "gemm_kernel<epi=%s>" % (e.group(1) if e else "?")
        if "IfLi" in name:
            base += "f32"
    return base


def main(root):
    acc = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(set))
    for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                if k is None:
                    continue
                c = row["Counter_Name"]
                acc[k][c] += float(row["Counter_Value"])
                calls[k][c].add(row["Dispatch_Id"])
    for k in sorted(acc):
        print(k)
        for c in sorted(acc[k]):
            n = len(calls[k][c])
            print(f"   {c:28s} total {acc[k][c]:.4e}  dispatches {n}  per-dispatch {acc[k][c] / max(n, 1):.4e}")
        a = acc[k]
        if "SQ_WAVE_CYCLES" in a:
            wc = a["SQ_WAVE_CYCLES"]
            print("   -> share of wave cycles: WAIT_ANY %.3f  WAIT_INST_ANY %.3f  ACTIVE_INST_ANY %.3f" % (
                a.get("SQ_WAIT_ANY", 0) / wc, a.get("SQ_WAIT_INST_ANY", 0) / wc, a.get("SQ_ACTIVE_INST_ANY", 0) / wc))
        if "FETCH_SIZE" in a or "WRITE_SIZE" in a:
            nf = max(len(calls[k].get("FETCH_SIZE", [])), 1)
            nw = max(len(calls[k].get("WRITE_SIZE", [])), 1)
            print("   -> HBM per dispatch: read %.1f MB (2 x FETCH_SIZE), write %.1f MB" % (
                2 * a.get("FETCH_SIZE", 0) * 1024 / nf / 1e6, a.get("WRITE_SIZE", 0) * 1024 / nw / 1e6))


if __name__ == "__main__":
    main(sys.argv[1])
