#!/usr/bin/env python3
"""Measurement aid: kernel time of every lane mapping vs batch size, per (bits, channels) geometry,
one-block streams at max_block_size 1024 - where the host's thresholds (aad_hip_engine.hip
kMappingTable) come from.  Decode: split quad ("quad"), fused quad ("quad-fused"), dense.  Encode:
quad, dense.  One JSON line per (geometry, batch size), then the suggested crossovers.
usage: python3 tools/mapping_crossover.py [--quick] > profiles/rNN_mapping_crossover.jsonl"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aad_amd.capi import make_parameter
from aad_amd.engine import Engine, parse_header
from aad_amd.synth import synth_pcm

SIZES = (250, 500, 1000, 2000, 3000, 4000, 6000, 8000, 10000, 12000, 16000, 20000, 24000, 32000, 48000)


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 4)


def main():
    quick = "--quick" in sys.argv
    engine = Engine(0)
    torch.cuda.set_stream(engine.stream)
    rows = []
    for bits in (4, 3, 2):
        for ch in (2, 1):
            spb = {4: 1984, 3: 2632, 2: 3960}[bits] // ch if not (bits == 3 and ch == 1) else 2684
            param = make_parameter(ch, bits, 1024, 48000, False, 0)
            base = torch.from_numpy(synth_pcm(1000, spb, ch, seed=1234)).cuda()
            for rec in SIZES:
                if quick and rec not in (1000, 4000, 8000, 16000, 32000):
                    continue
                streams = rec // ch
                pcm = base.repeat((-(-streams // 1000), 1, 1))[:streams].contiguous()
                engine.set_mapping("auto")
                plan = engine.uniform_encode_plan(param, streams, spb)
                img = torch.zeros((streams, plan.stride), dtype=torch.uint8, device="cuda")
                plan.run(pcm, img)
                torch.cuda.synchronize()
                hd = parse_header(bytes(img[0, :31].cpu().numpy()))
                dplan = engine.uniform_decode_plan(hd, streams, plan.stride, plan.image_size)
                out = torch.zeros_like(pcm)
                row = dict(bits=bits, channels=ch, samples_per_block=spb, recurrences=streams * ch)
                for mapping in ("quad", "dense"):
                    engine.set_mapping(mapping)
                    row["encode_%s_ms" % mapping] = timed(lambda: plan.run(pcm, img), 10)
                for mapping in ("quad", "quad-fused", "dense"):
                    engine.set_mapping(mapping)
                    row["decode_%s_ms" % mapping] = timed(lambda: dplan.run(img, out), 10)
                plan.close()
                dplan.close()
                rows.append(row)
                print(json.dumps(row), flush=True)
    # suggested crossovers: the largest measured size up to which a mapping is still the fastest
    table = {}
    for bits in (4, 3, 2):
        for ch in (2, 1):
            g = [r for r in rows if r["bits"] == bits and r["channels"] == ch]
            enc_quad = max([r["recurrences"] for r in g if r["encode_quad_ms"] <= r["encode_dense_ms"]] or [0])
            dec_split = max([r["recurrences"] for r in g if r["decode_quad_ms"] <= min(r["decode_quad-fused_ms"], r["decode_dense_ms"])] or [0])
            dec_fused = max([r["recurrences"] for r in g if r["decode_quad-fused_ms"] <= r["decode_dense_ms"]] or [0])
            table["b%d_c%d" % (bits, ch)] = dict(encode_quad_upto=enc_quad, decode_split_upto=dec_split, decode_fused_upto=max(dec_fused, dec_split))
    print(json.dumps({"suggested": table}))


if __name__ == "__main__":
    main()
