#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc / --kernel-trace CSV output per kernel: mean counter values, mean
duration, and per-wave / per-sample derived figures for the AAD kernels.
usage: tools/pmc_summary.py <dir with *_counter_collection.csv> [samples_per_lane]"""
import collections
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    per_lane = float(sys.argv[2]) if len(sys.argv) > 2 else 988.0
    cc = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
    kt = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in cc:
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for f in kt:
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k in sorted(set(agg) | set(dur)):
        if "aad::" not in k:
            continue
        print(k[:90])
        if dur[k]:
            v = dur[k]
            print("   duration_ns mean=%.0f min=%d n=%d" % (sum(v) / len(v), min(v), len(v)))
        c = {n: sum(v) / len(v) for n, v in agg[k].items()}
        for n in sorted(c):
            print("   %-26s %14.0f" % (n, c[n]))
        waves = c.get("SQ_WAVES")
        if "SQ_WAVE_CYCLES" in c:
            # SQ_* cycle counters tick every 4 shader cycles (MI355X_MICROARCH.md)
            wc = c["SQ_WAVE_CYCLES"] * 4
            print("   -> wave-cycles(total) %.0f" % wc)
            for n in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS"):
                if n in c:
                    print("   -> %-20s %5.1f %% of wave cycles" % (n, 100.0 * c[n] / c["SQ_WAVE_CYCLES"]))
        if "SQ_INSTS_VALU" in c:
            lanes_waves = c.get("SQ_WAVES", 32.0)
            print("   -> VALU/wave/sample %.2f (assuming %d waves, %.0f samples per lane)" % (c["SQ_INSTS_VALU"] / lanes_waves / per_lane, lanes_waves, per_lane))


if __name__ == "__main__":
    main()
