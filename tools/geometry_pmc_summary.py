#!/usr/bin/env python3
"""Summary of tools/saturated_geometries_pmc.sh: per geometry and kernel the duration, instruction counts per wave and
per sample, VALU-active share, LDS conflict cycles per LDS instruction, and HBM traffic against the algorithmic bytes
of each side (PCM: 2 B per channel-sample; codes: the images).  FETCH_SIZE is doubled as
/opt/skills/guides/MI355X_MICROARCH.md (HBM) prescribes for 16-B-per-lane streams; the raw figure is printed too.
usage: tools/geometry_pmc_summary.py <dir holding b<bits>c<channels>/ sub-directories>"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stamp_pmc import collect  # noqa: E402


def main():
    root = sys.argv[1]
    print("# dense kernels at saturation, every fast-path geometry; traffic per launch in MB")
    print("# x_pcm / x_codes: ratio to that side's algorithmic bytes (encode reads PCM and writes codes, decode the reverse)")
    for name in sorted(os.listdir(root), key=lambda n: (n[3], -int(n[1]))):
        d = os.path.join(root, name)
        if not os.path.isdir(d) or not name.startswith("b"):
            continue
        bits, ch = int(name[1]), int(name[3])
        plain = json.loads(open(os.path.join(d, "plain.json")).read().strip().splitlines()[-1])
        streams, samples = plain["streams"], plain["samples_per_channel"]
        n = streams * samples * ch
        pcm_bytes = 2.0 * n
        spb = {4: 1984, 3: 2632, 2: 3960}[bits] // ch
        block = {4: 1024, 3: 1020, 2: 1024}[bits]
        code_bytes = streams * (31 + block * (samples // spb))
        kernels = collect(d, skip=1)
        print("\n== %d-bit %s: %d streams x %d samples/ch = %.1f M samples; PCM %.1f MB, images %.1f MB; HIP events: encode %.4f ms (%.0f Gsamples/s), decode %.4f ms (%.0f Gsamples/s)"
              % (bits, "stereo" if ch == 2 else "mono", streams, samples, n / 1e6, pcm_bytes / 1e6, code_bytes / 1e6,
                 plain["encode_ms"], plain["encode_gsps"], plain["decode_ms"], plain["decode_gsps"]))
        for k, e in sorted(kernels.items()):
            role = "encode" if "encode" in k.split("<")[0] else "decode"
            if e.get("launches", 0) < 2 and "SQ_WAVES" not in e:
                continue
            line = "  %s %s" % (role, k.split("(")[0].replace("void aad::", ""))
            print(line)
            if "duration_ns" in e:
                print("     duration %.1f us (min %.1f)" % (e["duration_ns"] / 1e3, e["duration_min_ns"] / 1e3))
            if "SQ_INSTS_VALU" in e and e.get("SQ_WAVES"):
                w = e["SQ_WAVES"]
                per_rec = samples if role == "encode" else spb
                print("     per wave: VALU %.0f (%.2f per sample)  LDS %.0f  VMEM rd %.0f wr %.0f  SALU %.0f   waves %d"
                      % (e["SQ_INSTS_VALU"] / w, e["SQ_INSTS_VALU"] / w / per_rec, e.get("SQ_INSTS_LDS", 0) / w,
                         e.get("SQ_INSTS_VMEM_RD", 0) / w, e.get("SQ_INSTS_VMEM_WR", 0) / w, e.get("SQ_INSTS_SALU", 0) / w, w))
            if "GRBM_GUI_ACTIVE" in e and "SQ_ACTIVE_INST_VALU" in e:
                cyc = e["GRBM_GUI_ACTIVE"] / 8.0
                print("     VALU active %.1f %% of SIMD cycles (SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / kernel cycles)"
                      % (100.0 * e["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / cyc))
            if "SQ_LDS_BANK_CONFLICT" in e and e.get("SQ_ACTIVE_INST_LDS"):
                print("     LDS bank-conflict cycles %.0f = %.1f %% of LDS-active cycles" % (e["SQ_LDS_BANK_CONFLICT"], 100.0 * e["SQ_LDS_BANK_CONFLICT"] / e["SQ_ACTIVE_INST_LDS"]))
            extra = ["%s %.0f" % (n, e[n]) for n in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS",
                                                      "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE") if n in e]
            if extra:
                print("     raw: " + "  ".join(extra))
            if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
                rd, wr = e["FETCH_SIZE"] * 1024.0, e["WRITE_SIZE"] * 1024.0
                rd_side, wr_side = (pcm_bytes, code_bytes) if role == "encode" else (code_bytes, pcm_bytes)
                print("     HBM read %.1f MB raw, x2 = %.1f MB = %.2fx its %s bytes (raw %.2fx);  write %.1f MB = %.2fx its %s bytes;  total (x2) %.2fx algorithmic"
                      % (rd / 1e6, 2 * rd / 1e6, 2 * rd / rd_side, "PCM" if role == "encode" else "code", rd / rd_side,
                         wr / 1e6, wr / wr_side, "code" if role == "encode" else "PCM", (2 * rd + wr) / (pcm_bytes + code_bytes)))


if __name__ == "__main__":
    main()
