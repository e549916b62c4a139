#!/usr/bin/env python3
"""Human-readable summary of the `configs[]` workloads of a PMC stamp file (tools/collect_config_profiles.sh ->
tools/stamp_pmc.py --merge): per BASELINE shape and kernel the counter traffic against the algorithmic bytes, the VMEM
instruction counts and the VALU instructions per sample.  usage: tools/config_pmc_summary.py profiles/r04_pmc_stamp.json"""
import json
import sys

# workload -> (description, channels, block_size, samples_per_block)
SHAPES = {
    "cfg4_3bit": ("BASELINE config 4: 10 000 eight-channel 3-bit one-block segments (80 000 recurrences)", 8, 1008, 292),
    "cfg4_2bit": ("BASELINE config 4: 10 000 eight-channel 2-bit one-block segments (80 000 recurrences)", 8, 1024, 444),
    "cfg2ii": ("BASELINE config 2(ii): 1000 stereo 4-bit streams x 16 blocks (2000 encode recurrences, 32 000 decode)", 2, 1024, 992),
    "cfg5_shard": ("BASELINE config 5, one GPU's shard: 1250 stereo 4-bit files x 10 blocks", 2, 1024, 992),
}


def main():
    doc = json.load(open(sys.argv[1]))
    print("Counter evidence for the BASELINE shapes on bench.py's configs[] rows - %s" % sys.argv[1])
    print("rocprofv3 --pmc, one pass per counter group, FETCH_SIZE and WRITE_SIZE each in a pass of its own (no other trace domain);")
    print("means per launch of tools/saturated_probe.py's kernels.  read x2 = FETCH_SIZE x 2 (gfx950 tallies 64 B per 128-B request of a")
    print("wide streaming read, MI355X_MICROARCH.md section HBM).  kernel sources: sha256 %s" % doc["kernel_source_sha256"][:16])
    for name, (desc, ch, block_size, spb) in SHAPES.items():
        w = doc["workloads"].get(name)
        if not w:
            continue
        streams, samples = w["streams"], w["samples_per_channel"]
        blocks = -(-samples // spb)
        n = streams * samples * ch
        pcm_bytes = 2 * n
        image_bytes = streams * (31 + blocks * block_size)
        print("\n%s\n  %d streams x %d samples/ch x %d ch = %d channel-samples; algorithmic bytes: PCM %d + images %d = %d" %
              (desc, streams, samples, ch, n, pcm_bytes, image_bytes, pcm_bytes + image_bytes))
        for role in ("encode", "decode"):
            k = w["kernels"].get(role)
            if not k:
                continue
            alg_r, alg_w = (pcm_bytes, image_bytes) if role == "encode" else (image_bytes, pcm_bytes)
            waves = k.get("SQ_WAVES") or 1
            per_rec = samples if role == "encode" else min(spb, samples)
            print("  %s  %s" % (role, k["kernel"].replace("void aad::", "").split("(")[0]))
            print("    duration %.1f us (kernel trace mean)   waves %d   VALU %d = %.0f per wave = %.2f per sample of a recurrence" %
                  (k.get("duration_ns", 0) / 1e3, waves, k.get("SQ_INSTS_VALU", 0), k.get("SQ_INSTS_VALU", 0) / waves, k.get("SQ_INSTS_VALU", 0) / waves / per_rec))
            print("    read  x2 %11d B = %.3f x algorithmic read  (%d)   raw FETCH_SIZE %d B   VMEM_RD %d wave-instructions" %
                  (k["read_bytes_x2"], k["read_bytes_x2"] / alg_r, alg_r, k["read_bytes_raw"], k.get("SQ_INSTS_VMEM_RD", 0)))
            print("    write    %11d B = %.3f x algorithmic write (%d)                          VMEM_WR %d wave-instructions" %
                  (k["write_bytes"], k["write_bytes"] / alg_w, alg_w, k.get("SQ_INSTS_VMEM_WR", 0)))
            print("    total    %11d B = %.3f x algorithmic" % (k["hbm_bytes_per_launch"], k["hbm_bytes_per_launch"] / (pcm_bytes + image_bytes)))
            if k.get("duration_ns"):
                print("    algorithmic bytes / duration = %.1f GB/s = %.4f of 8 TB/s" %
                      ((pcm_bytes + image_bytes) / k["duration_ns"], (pcm_bytes + image_bytes) / k["duration_ns"] / 8000.0))


if __name__ == "__main__":
    main()
