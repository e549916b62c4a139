#!/usr/bin/env python3
"""Turn the rocprofv3 passes of tools/collect_profiles.sh into ONE stamped JSON file that bench.py
quotes on its line (profiles/rNN_pmc_stamp.json): per workload ("headline" = bench.py's step batch,
"saturated" = tools/saturated_probe.py's chip-filling batch) and per kernel role (encode / decode) the
mean per launch of every counter collected, the kernel-trace duration, and the derived figures

  hbm traffic     FETCH_SIZE / WRITE_SIZE come in KiB; on gfx950 FETCH_SIZE tallies 64 B per 128-B request
                  of a wide (16 B per lane) streaming read, so the corrected read side is FETCH_SIZE x 2
                  (/opt/skills/guides/MI355X_MICROARCH.md, section HBM): traffic = 2 x FETCH + WRITE, the raw
                  sum is kept beside it.  FETCH_SIZE and WRITE_SIZE are taken in separate passes (the TCC
                  block cannot hold both).
  valu            SQ_INSTS_VALU (wave-instructions) per launch and per wave.

The file carries the SHA-256 over aad_amd/csrc (bench.kernel_source_digest): bench.py refuses to quote
it once the kernel sources have changed.
usage: tools/stamp_pmc.py [--merge] <out.json> <workload>=<dir of pass sub-directories>[:streams:samples_per_channel] ...
--merge: keep the workloads <out.json> already holds (they must carry the same kernel-source digest) and add / replace the named ones."""
import collections
import csv
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def collect(root, skip=2):
    """-> {kernel name: {counter: mean per launch, "duration_ns": mean, "duration_min_ns": min, "launches": n}}"""
    counters = collections.defaultdict(lambda: collections.defaultdict(list))
    durations = collections.defaultdict(list)
    for db in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
        con = sqlite3.connect(db)
        try:
            for name, counter, value in con.execute("select kernel_name, counter_name, value from counters_collection order by dispatch_id"):
                if "aad::" in name:
                    counters[name][counter].append(float(value))
        except sqlite3.Error:
            pass
        try:
            for name, start, end in con.execute("select name, start, end from kernels order by start"):
                if "aad::" in name:
                    durations[name].append(end - start)
        except sqlite3.Error:
            pass
    for f in glob.glob(os.path.join(root, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "aad::" in r["Kernel_Name"]:
                counters[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(root, "**", "*_kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "aad::" in r["Kernel_Name"]:
                durations[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out = {}
    for k in set(counters) | set(durations):
        e = {}
        for n, v in counters[k].items():
            v = v[skip:] if len(v) > skip else v
            e[n] = sum(v) / len(v)
        d = durations.get(k)
        if d:
            d = d[skip:] if len(d) > skip else d
            e["duration_ns"] = sum(d) / len(d)
            e["duration_min_ns"] = min(d)
            e["launches"] = len(d)
        out[k] = e
    return out


def derive(e):
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        rd, wr = e["FETCH_SIZE"] * 1024.0, e["WRITE_SIZE"] * 1024.0
        e["read_bytes_raw"], e["read_bytes_x2"], e["write_bytes"] = int(rd), int(2 * rd), int(wr)
        e["hbm_bytes_per_launch_raw"] = int(rd + wr)
        e["hbm_bytes_per_launch"] = int(2 * rd + wr)  # the guide's correction: FETCH_SIZE x 2
    if "SQ_INSTS_VALU" in e and e.get("SQ_WAVES"):
        e["valu_insts_per_wave"] = e["SQ_INSTS_VALU"] / e["SQ_WAVES"]
    if "SQ_LDS_BANK_CONFLICT" in e and e.get("SQ_INSTS_LDS"):
        e["lds_conflict_cycles_per_lds_inst"] = e["SQ_LDS_BANK_CONFLICT"] / e["SQ_INSTS_LDS"]
    return {k: (round(v, 3) if isinstance(v, float) else v) for k, v in e.items()}


def main():
    argv = sys.argv[1:]
    merge = bool(argv) and argv[0] == "--merge"
    if merge:
        argv = argv[1:]
    out_path = argv[0]
    sys.path.insert(0, ROOT)
    from bench import kernel_source_digest
    doc = {"kernel_source_sha256": kernel_source_digest(),
           "source": "rocprofv3 --kernel-trace and --pmc passes (FETCH_SIZE and WRITE_SIZE in separate passes, no other trace "
                     "domain) collected by tools/collect_profiles.sh / collect_config_profiles.sh; means per launch",
           "workloads": {}}
    if merge and os.path.exists(out_path):
        old = json.load(open(out_path))
        if old.get("kernel_source_sha256") != doc["kernel_source_sha256"]:
            raise SystemExit("stamp_pmc --merge: %s was taken from other kernel sources" % out_path)
        doc["workloads"] = old.get("workloads", {})
    for spec in argv[1:]:
        name, rest = spec.split("=", 1)
        parts = rest.split(":")
        kernels = collect(parts[0])
        roles = {}
        for role in ("encode", "decode"):
            # the role's kernel is the one the passes saw most: the others are one-off launches (pipeline priming etc.)
            cands = [(e.get("launches", 0), len(e), k) for k, e in kernels.items() if role in k.split("<")[0]]
            if cands:
                k = max(cands)[2]
                roles[role] = dict(kernel=k, **derive(dict(kernels[k])))
        w = {"kernels": roles}
        if len(parts) >= 3:
            w["streams"], w["samples_per_channel"] = int(parts[1]), int(parts[2])
        doc["workloads"][name] = w
    json.dump(doc, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps({n: {r: {k: e.get(k) for k in ("kernel", "duration_ns", "hbm_bytes_per_launch", "hbm_bytes_per_launch_raw",
                                                     "SQ_INSTS_VALU", "SQ_WAVES", "lds_conflict_cycles_per_lds_inst")}
                          for r, e in w["kernels"].items()} for n, w in doc["workloads"].items()}, indent=1))


if __name__ == "__main__":
    main()
