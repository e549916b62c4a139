#!/usr/bin/env python3
"""Encode / decode kernel time against batch size, one-block streams, every fast-path geometry - one process, HIP-event times.
Looks for steps that the hardware does not explain (a launch geometry that leaves SIMDs idle or runs a second round of waves):
time per 1024 waves' worth of lanes should rise smoothly.  usage (gpurun): python3 tools/size_sweep.py [--trials T] > out.txt"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=0)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--mapping", default="auto", help="lane mapping forced for the timed launches (auto | dense | dense-tiled | quad | quad-fused)")
    ap.add_argument("--sizes", default="", help="comma-separated lane counts (default: the built-in ladder)")
    ap.add_argument("--geometries", default="", help="e.g. 1x4,2x2 (channels x bits; default: all six)")
    args = ap.parse_args()
    import torch
    from aad_amd.capi import make_parameter
    from aad_amd.engine import Engine, parse_header
    from aad_amd.synth import synth_pcm
    engine = Engine(0)
    torch.cuda.set_stream(engine.stream)
    sizes = [1000, 4000, 8192, 12000, 16384, 20000, 24576, 32768, 40000, 49152, 57344, 65536, 80000, 98304, 131072, 196608, 262144]
    if args.sizes:
        sizes = [int(v) for v in args.sizes.split(",")]
    geometries = ((2, 4), (1, 4), (2, 3), (1, 3), (2, 2), (1, 2))
    if args.geometries:
        geometries = tuple(tuple(int(v) for v in g.split("x")) for g in args.geometries.split(","))
    for ch, bits in geometries:
        import ctypes as C
        bs_, spb_ = C.c_uint16(0), C.c_uint32(0)
        assert engine.lib.AADEncoder_CalculateBlockSize(1024, ch, bits, C.byref(bs_), C.byref(spb_)) == 0 or ch > 2
        spb = spb_.value if ch <= 2 else {4: 224, 3: 292, 2: 444}[bits] * 8 // ch if ch == 8 else {4: 1984, 3: 2632, 2: 3960}[bits] // ch // 8 * 8
        base = torch.from_numpy(synth_pcm(1000, spb, ch, seed=1234)).cuda()
        param = make_parameter(ch, bits, 1024, 48000, False, args.trials)
        for lanes in sizes:
            streams = lanes // ch
            pcm = base.repeat((-(-streams // 1000), 1, 1))[:streams].contiguous()
            plan = engine.uniform_encode_plan(param, streams, spb)
            images = torch.zeros((streams, plan.stride), dtype=torch.uint8, device="cuda")
            plan.run(pcm, images)
            hd = parse_header(bytes(images[0, :31].cpu().numpy()))
            dplan = engine.uniform_decode_plan(hd, streams, plan.stride, plan.image_size)
            out = torch.zeros_like(pcm)
            dplan.run(images, out)
            torch.cuda.synchronize()
            engine.set_mapping(args.mapping)
            plan.run(pcm, images)
            dplan.run(images, out)  # first launch under the forced mapping (one-time scratch allocation)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            enc = dec = 0.0
            for _ in range(args.reps):
                ev[0].record()
                plan.run(pcm, images)
                ev[1].record()
                dplan.run(images, out)
                ev[2].record()
                torch.cuda.synchronize()
                enc += ev[0].elapsed_time(ev[1])
                dec += ev[1].elapsed_time(ev[2])
            engine.set_mapping("auto")
            print(json.dumps(dict(channels=ch, bits=bits, trials=args.trials, mapping=args.mapping, lanes=streams * ch, waves_dense=-(-streams * ch // 64),
                                  encode_ms=round(enc / args.reps, 4), decode_ms=round(dec / args.reps, 4))), flush=True)
            plan.close()
            dplan.close()
            del pcm, images, out
    engine.close()


if __name__ == "__main__":
    main()
