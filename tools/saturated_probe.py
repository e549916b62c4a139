#!/usr/bin/env python3
"""Run the encode and decode kernels a few times on a batch big enough to fill the chip
(default 262144 stereo one-block streams = 8 dense waves per SIMD), for rocprofv3 passes:
  rocprofv3 --kernel-trace --stats -d DIR -- python3 tools/saturated_probe.py
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU ... -d DIR -- python3 tools/saturated_probe.py
Prints the HIP-event kernel times as one JSON line."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=262144)
    ap.add_argument("--blocks", type=int, default=1)
    ap.add_argument("--bits", type=int, default=4)
    ap.add_argument("--channels", type=int, default=2)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import torch
    from aad_amd.capi import make_parameter
    from aad_amd.engine import Engine
    from aad_amd.synth import synth_pcm
    param = make_parameter(args.channels, args.bits, 1024, 48000, False, 0)
    engine = Engine(0)
    torch.cuda.set_stream(engine.stream)
    # samples per channel of one 1024-byte block (mono / stereo: the reference's geometry; 8 channels: BASELINE config 4's segments)
    spb = {4: 1984, 3: 2632, 2: 3960}[args.bits] // args.channels if args.channels <= 2 else {(8, 3): 292, (8, 2): 444}[(args.channels, args.bits)]
    samples = spb * args.blocks
    base = torch.from_numpy(synth_pcm(1000, samples, args.channels, seed=1234)).cuda()
    pcm = base.repeat((-(-args.streams // 1000), 1, 1))[:args.streams].contiguous()
    plan = engine.uniform_encode_plan(param, args.streams, samples)
    images = torch.zeros((args.streams, plan.stride), dtype=torch.uint8, device="cuda")
    plan.run(pcm, images)
    from aad_amd.engine import parse_header
    hd = parse_header(bytes(images[0, :31].cpu().numpy()))
    dplan = engine.uniform_decode_plan(hd, args.streams, plan.stride, plan.image_size)
    out = torch.zeros_like(pcm)
    dplan.run(images, out)
    torch.cuda.synchronize()
    if not os.environ.get("AAD_PROBE_NOCHECK"):  # destructive measurement builds decode garbage
        assert torch.equal(out[:1000], out[1000:2000]) if args.streams >= 2000 else True
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
    n = args.streams * samples * args.channels
    bps = 2.0 + hd.block_size / (hd.num_samples_per_block * args.channels)
    print(json.dumps(dict(streams=args.streams, samples_per_channel=samples, encode_ms=enc / args.reps, decode_ms=dec / args.reps,
                          encode_gsps=n / (enc / args.reps) / 1e6, decode_gsps=n / (dec / args.reps) / 1e6,
                          encode_tbs=n * bps / (enc / args.reps) / 1e9, decode_tbs=n * bps / (dec / args.reps) / 1e9)))
    engine.close()


if __name__ == "__main__":
    main()
