#!/usr/bin/env python3
"""Per-kernel summary (calls, total / average duration, share) of a rocprofv3 rocpd database.

    python tools/rocpd_stats.py results.db [out.csv]
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) <= 110 else name[:107] + "..."


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, (end - start) from kernels").fetchall()
    agg = {}
    for name, dur in rows:
        a = agg.setdefault(name, [0, 0])
        a[0] += 1
        a[1] += dur
    total = sum(a[1] for a in agg.values()) or 1
    lines = ["kernel,calls,total_ms,avg_us,percent"]
    for name, (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append('"%s",%d,%.3f,%.2f,%.2f' % (short(name), n, ns / 1e6, ns / 1e3 / n, 100.0 * ns / total))
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main()
