#!/usr/bin/env python
"""Summarise a rocprofv3 `--kernel-trace --stats` run (rocpd SQLite .db or *_kernel_stats.csv)
into the per-kernel table committed under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof1/r1_results.db > profiles/r01_kernel_stats.md
"""
import csv
import sqlite3
import sys


def from_db(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else cols[0]
    rows = cur.execute("select %s, (end - start) from kernels" % name_col).fetchall()
    agg = {}
    for name, dur in rows:
        a = agg.setdefault(name, [0, 0.0, 1e30, 0.0])
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
    return agg


def from_csv(path):
    agg = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            agg[r["Name"]] = [int(r["Calls"]), float(r["TotalDurationNs"]), float(r["MinNs"]),
                              float(r["MaxNs"])]
    return agg


def main():
    path = sys.argv[1]
    agg = from_db(path) if path.endswith(".db") else from_csv(path)
    total = sum(a[1] for a in agg.values())
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        short = name if len(name) < 110 else name[:107] + "..."
        print("| `%s` | %d | %.3f | %.2f | %.2f | %.2f | %.1f |"
              % (short, a[0], a[1] / 1e6, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3,
                 100.0 * a[1] / total))
    print("\ntotal kernel time: %.3f ms over %d dispatches" % (total / 1e6, sum(a[0] for a in agg.values())))


if __name__ == "__main__":
    main()
