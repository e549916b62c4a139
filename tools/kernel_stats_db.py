#!/usr/bin/env python3
"""Per-kernel duration statistics from a rocprofv3 --kernel-trace pass stored as a rocpd SQLite
database (rocprofv3 7.x default output).  usage: tools/kernel_stats_db.py <dir> [csv_out]"""
import collections
import glob
import os
import sqlite3
import sys


def main():
    root = sys.argv[1]
    rows = collections.defaultdict(list)
    for db in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
        con = sqlite3.connect(db)
        for name, start, end in con.execute("select name, start, end from kernels order by start"):
            rows[name].append(end - start)
    total = sum(sum(v) for v in rows.values())
    lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage"]
    for name, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        lines.append('"%s",%d,%d,%.1f,%d,%d,%.2f' % (name, len(v), sum(v), sum(v) / len(v), min(v), max(v), 100.0 * sum(v) / total))
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
