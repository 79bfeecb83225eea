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
    """Kernel family of a (possibly mangled) kernel name; None for kernels that are not ours."""
    m = re.search(r"(gemm128_kernel|gemm256w_kernel|gemm256x_kernel|gemm256_kernel|gemm_tn256x_kernel|gemm_tn256_kernel|gemm_tn_kernel|mlm_mask_kernel|attn_bwd_dq_bf16_kernel|attn_bwd_dkv_bf16_kernel|"
                  r"attn_bf16_kernel|attn_f32_kernel|layernorm_bwd_kernel|layernorm_kernel|head_bwd_kernel|"
                  r"head_rows_kernel|head_kernel|ln_rows_finalize_kernel|embed_bwd_kernel|embed_kernel|decode_kernel|key_mask_kernel|colsum_kernel|"
                  r"reduce_slabs_kernel|rope_bwd_kernel|adamw_kernel|delta_kernel|transpose_kernel)", name)
    if not m:
        return None
    base = m.group(1)
    if base.startswith("gemm1") or base.startswith("gemm2"):
        # the epilogue is the second template argument
        e = re.search(r"Li(\d)E", name) or re.search(r"<[^,<>]+,\s*(\d)\s*[,>]", name)
        # the profiler's demangler garbles the <__bf16, 1> instantiation into "<bool _Accum, int, E>"
        # the profiler's demangler also garbles the three-argument forms: "<bool _Accum, int, ELb, E>" is
        # <__bf16, 3, true>, "<bool _Accum, int, E, true>" is <__bf16, 1, true>, "<bool _Accum, int, EL, bool, E>"
        # is <__bf16, 2, true>
        # (five template arguments since the streamed residual: <T, EPI, FOLD, PANELS, STREAM>)
        if "_Accum" in name and "EL, bool, E" in name:
            base += "<epi=2>+split"         # <__bf16, 2, true, ...>: residual product on the bf16 hi/lo planes
            m2 = re.search(r"EL, bool, E, (?:false|true), (\d)", name)
            if m2 and m2.group(1) != "0":
                base += "+stream%s" % m2.group(1)
        elif "_Accum" in name and "ELb, E" in name:
            base += "<epi=3>+ln"
            if re.search(r"ELb, E, true", name):
                base += "+panels"
        else:
            base += "<epi=%s>" % (e.group(1) if e else ("1" if "_Accum" in name else "?"))
            if "ELb1E" in name or ", true>" in name or re.search(r"_Accum, int, E, true", name):
                base += "+ln"          # the variant with a LayerNorm folded in
        if "IfLi" in name or "<float" in name:
            base += "f32"
    return base


def mfma_busy(a, c):
    """Fraction of SIMD-cycles with the matrix pipe busy: the SQ counter is summed over the chip's
    1024 SIMDs, GRBM_GUI_ACTIVE over its 8 XCDs."""
    busy = a["SQ_VALU_MFMA_BUSY_CYCLES"] / max(len(c["SQ_VALU_MFMA_BUSY_CYCLES"]), 1)
    active = a["GRBM_GUI_ACTIVE"] / max(len(c["GRBM_GUI_ACTIVE"]), 1) / 8.0
    return busy / (1024.0 * active) if active else 0.0


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
        if "SQ_VALU_MFMA_BUSY_CYCLES" in a and "GRBM_GUI_ACTIVE" in a:
            print("   -> MFMA pipes busy %.3f of the time (MFMA_BUSY summed over 1024 SIMDs / (GUI_ACTIVE summed over 8 XCDs / 8))" %
                  mfma_busy(a, calls[k]))
        if "FETCH_SIZE" in a or "WRITE_SIZE" in a:
            nf = max(len(calls[k].get("FETCH_SIZE", [])), 1)
            nw = max(len(calls[k].get("WRITE_SIZE", [])), 1)
            print("   -> HBM per dispatch: read %.1f MB (2 x FETCH_SIZE), write %.1f MB" % (
                2 * a.get("FETCH_SIZE", 0) * 1024 / nf / 1e6, a.get("WRITE_SIZE", 0) * 1024 / nw / 1e6))


    if len(sys.argv) > 2:      # machine-readable HBM traffic per dispatch for bench.py
        import json
        out = {}
        for k, a in acc.items():
            if "FETCH_SIZE" in a and "WRITE_SIZE" in a:
                nf, nw = len(calls[k]["FETCH_SIZE"]), len(calls[k]["WRITE_SIZE"])
                out[k] = {"read_bytes": 2 * a["FETCH_SIZE"] * 1024 / nf, "write_bytes": a["WRITE_SIZE"] * 1024 / nw,
                          "dispatches": nf}
                if "SQ_VALU_MFMA_BUSY_CYCLES" in a and "GRBM_GUI_ACTIVE" in a:
                    out[k]["mfma_busy"] = round(mfma_busy(a, calls[k]), 4)
        json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bytes per dispatch; "
                             "reads = 2 x FETCH_SIZE KiB (gfx950 correction)", "kernels": out},
                  open(sys.argv[2], "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1])
