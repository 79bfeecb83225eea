#!/usr/bin/env python3
"""Per-kernel totals of the traces tools/trace_ab.sh left under gpurun_out/: one column per build and repeat.
    python tools/trace_ab_report.py tag n_builds"""
import csv, os, sys
tag, n = sys.argv[1], int(sys.argv[2])
here = os.path.dirname(os.path.abspath(__file__))
cols = {}
for i in range(1, n + 1):
    for rep in (1, 2):
        with open(os.path.join(here, "..", "gpurun_out", f"trace_{tag}_{i}_{rep}.csv")) as fh:
            cols[(i, rep)] = {r["kernel"][:72]: float(r["total_ms"]) for r in csv.DictReader(fh)}
allk = set().union(*[set(c) for c in cols.values()])
keys = sorted(allk, key=lambda k: -max(c.get(k, 0.0) for c in cols.values()))[:int(sys.argv[3]) if len(sys.argv) > 3 else 18]
print("%-72s" % "kernel (total ms)" + "".join("  b%d.%d   " % c for c in sorted(cols)))
for k in keys:
    print("%-72s" % k + "".join("%8.2f " % cols[c].get(k, 0.0) for c in sorted(cols)))
print("%-72s" % "all kernels" + "".join("%8.2f " % sum(cols[c].values()) for c in sorted(cols)))
