#!/usr/bin/env python3
"""Experiment: what a headline step costs beyond its encode kernel (64 us by rocprofv3).  Back-to-back encode launches on one
stream, bare / with an event record behind each / with the decode launched on a second stream behind each event.
usage (GPU box): python3 tools/experiments/launch_gap_probe.py"""
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    import torch
    from aad_amd.capi import make_parameter
    from aad_amd.engine import Engine, parse_header
    from aad_amd.synth import synth_pcm
    param = make_parameter(2, 4, 1024, 48000, False, 0)
    e1, e2 = Engine(0, stream=torch.cuda.Stream()), Engine(0, stream=torch.cuda.Stream())
    pcm = torch.from_numpy(synth_pcm(1000, 992, 2, seed=1234)).cuda()
    out = torch.zeros_like(pcm)
    enc = e1.uniform_encode_plan(param, 1000, 992)
    imgs = [torch.zeros((1000, enc.stride), dtype=torch.uint8, device="cuda") for _ in range(32)]
    enc.run(pcm, imgs[0], None)
    torch.cuda.synchronize()
    hd = parse_header(bytes(imgs[0][0, :31].cpu().numpy()))
    dec = e2.uniform_decode_plan(hd, 1000, enc.stride, enc.image_size)
    evs = [torch.cuda.Event() for _ in range(32)]
    s1, s2 = e1.stream, e2.stream

    def bare(k):
        enc.run(pcm, imgs[k % 32], None, ordered=False)

    def with_event(k):
        enc.run(pcm, imgs[k % 32], None, ordered=False)
        evs[k % 32].record(s1)

    def with_decode(k):
        enc.run(pcm, imgs[k % 32], None, ordered=False)
        evs[k % 32].record(s1)
        s2.wait_event(evs[k % 32])
        dec.run(imgs[k % 32], out, ordered=False)

    def decode_every_other_event(k):  # the decode stream waits on one event per TWO encodes and then decodes both
        enc.run(pcm, imgs[k % 32], None, ordered=False)
        if k % 2 == 1:
            evs[k % 32].record(s1)
            s2.wait_event(evs[k % 32])
            dec.run(imgs[(k - 1) % 32], out, ordered=False)
            dec.run(imgs[k % 32], out, ordered=False)

    def grouped(g):
        def fn(k):
            enc.run(pcm, imgs[k % 32], None, ordered=False)
            if k % g == g - 1:
                evs[k % 32].record(s1)
                s2.wait_event(evs[k % 32])
                for j in range(g):
                    dec.run(imgs[(k - g + 1 + j) % 32], out, ordered=False)
        return fn

    for name, fn in (("encode launches back to back", bare), ("+ event record behind each", with_event),
                     ("+ decode on a second stream behind each event (the pipelined step)", with_decode),
                     ("one event per two encodes, two decodes behind it", decode_every_other_event),
                     ("one event per 4 encodes", grouped(4)), ("one event per 8 encodes", grouped(8)), ("one event per 16 encodes", grouped(16)),
                     ("encode launches back to back (again)", bare)):
        for k in range(64):
            fn(k)
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            for k in range(200):
                fn(k)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 200 * 1e6)
        print(json.dumps({"loop": name, "us_per_step": round(statistics.median(ts), 2), "min": round(min(ts), 2)}))


if __name__ == "__main__":
    main()
