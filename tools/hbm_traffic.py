#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE in one, WRITE_SIZE in the other; the TCC block
cannot hold both) of `bench.py` into profiles/rNN_hbm_traffic.json: HBM bytes per launch of the
encode and decode kernels, corrected as /opt/skills/guides/MI355X_MICROARCH.md section HBM says
(FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide
coalesced stream, so the read side is reported both raw and doubled - these kernels read 16-B
pieces per lane, the doubled figure is the upper bound).
usage: tools/hbm_traffic.py <fetch_dir> <write_dir> <streams> <samples_per_channel> <out.json>"""
import collections
import csv
import glob
import json
import os
import sys


def per_kernel(d, counter):
    """mean counter value per launch of the encode / decode kernels; reads rocprofv3's CSV output
    or its default rocpd SQLite database, whichever the pass directory holds"""
    import sqlite3
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "aad::" in r["Kernel_Name"]:
                agg["encode" if "encode" in r["Kernel_Name"] else "decode"].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        con = sqlite3.connect(f)
        for name, cname, value in con.execute("select kernel_name, counter_name, value from counters_collection order by dispatch_id"):
            if cname == counter and "aad::" in name:
                agg["encode" if "encode" in name else "decode"].append(float(value))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    fetch_dir, write_dir, streams, samples, out = sys.argv[1:6]
    fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    kernels = {}
    for k in ("encode", "decode"):
        if k in fetch and k in write:
            rd_raw, wr = fetch[k] * 1024.0, write[k] * 1024.0
            kernels[k] = {"FETCH_SIZE_KiB": round(fetch[k], 1), "WRITE_SIZE_KiB": round(write[k], 1),
                          "read_bytes_raw": int(rd_raw), "read_bytes_doubled": int(2 * rd_raw), "write_bytes": int(wr),
                          "hbm_bytes_per_launch": int(rd_raw + wr), "hbm_bytes_per_launch_upper": int(2 * rd_raw + wr)}
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_digest  # the stamp bench.py checks before it quotes these numbers
    json.dump({"streams": int(streams), "samples_per_channel": int(samples), "kernels": kernels,
               "kernel_source_sha256": kernel_source_digest(),
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on bench.py --no-extras"}, open(out, "w"), indent=1)
    print(json.dumps(kernels, indent=1))


if __name__ == "__main__":
    main()
