#!/usr/bin/env python3
"""Summarise rocprofv3 passes stored as rocpd SQLite databases (the default output format of
rocprofv3 7.x): mean of every counter per AAD kernel, kernel durations from a --kernel-trace
pass, and derived figures (VALU utilisation, HBM bytes per launch).
usage: tools/pmc_db_summary.py <dir holding the pass sub-directories> [skip_first_n_launches]"""
import collections
import glob
import os
import sqlite3
import sys


def main():
    root = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    counters = collections.defaultdict(lambda: collections.defaultdict(list))
    durations = collections.defaultdict(list)
    for db in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
        con = sqlite3.connect(db)
        try:
            for name, counter, value in con.execute("select kernel_name, counter_name, value from counters_collection order by dispatch_id"):
                if "aad::" in name:
                    counters[name][counter].append(value)
        except sqlite3.Error:
            pass
        try:
            for name, start, end in con.execute("select name, start, end from kernels order by start"):
                if "aad::" in name:
                    durations[name].append(end - start)
        except sqlite3.Error:
            pass
    for k in sorted(set(counters) | set(durations)):
        print(k[:100])
        c = {}
        for n, v in sorted(counters[k].items()):
            v = v[skip:] if len(v) > skip else v
            c[n] = sum(v) / len(v)
            print("   %-26s %16.1f   (n=%d)" % (n, c[n], len(v)))
        d = durations.get(k)
        if d:
            d = d[skip:] if len(d) > skip else d
            print("   duration_ns mean=%.0f min=%d n=%d" % (sum(d) / len(d), min(d), len(d)))
        if "GRBM_GUI_ACTIVE" in c and "SQ_ACTIVE_INST_VALU" in c:
            # GRBM_GUI_ACTIVE sums the 8 XCDs; SQ_ACTIVE_INST_VALU counts 4-cycle quads over 1024 SIMDs
            cyc = c["GRBM_GUI_ACTIVE"] / 8.0
            print("   -> VALU busy %.1f %% of the kernel's cycles (per SIMD)" % (100.0 * c["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / cyc))
        if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c:
            print("   -> VALU instructions per wave %.0f" % (c["SQ_INSTS_VALU"] / c["SQ_WAVES"]))
        if "FETCH_SIZE" in c:
            print("   -> HBM read  %.1f MB raw (x2 = %.1f MB if every request was a 128-B line)" % (c["FETCH_SIZE"] * 1024 / 1e6, c["FETCH_SIZE"] * 2048 / 1e6))
        if "WRITE_SIZE" in c:
            print("   -> HBM write %.1f MB" % (c["WRITE_SIZE"] * 1024 / 1e6))


if __name__ == "__main__":
    main()
