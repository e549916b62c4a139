#!/usr/bin/env python3
"""Time the encode / decode kernels on the other BASELINE shapes (not the bench headline):
config 4 (8-channel 2/3-bit, 10 000 one-block segments), config 5's per-GPU shard (1250 stereo
4-bit files x 10 blocks), and the serial worst case (1 stream x 1000 blocks)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aad_amd.capi import make_parameter  # noqa: E402
from aad_amd.engine import Engine, parse_header  # noqa: E402
from aad_amd.synth import synth_pcm  # noqa: E402


def run(eng, name, streams, samples, ch, bits, trials=0, reps=10):
    pcm = torch.from_numpy(synth_pcm(min(streams, 500), samples, ch, seed=7)).cuda()
    pcm = pcm.repeat((-(-streams // pcm.shape[0]), 1, 1))[:streams].contiguous()
    param = make_parameter(ch, bits, 1024, 48000, False, trials)
    enc = eng.uniform_encode_plan(param, streams, samples)
    img = torch.zeros((streams, enc.stride), dtype=torch.uint8, device="cuda")
    enc.run(pcm, img, None)
    torch.cuda.synchronize()
    hd = parse_header(bytes(img[0, :31].cpu().numpy()))
    dec = eng.uniform_decode_plan(hd, streams, enc.stride, enc.image_size)
    out = torch.zeros((streams, samples, ch), dtype=torch.int16, device="cuda")
    dec.run(img, out)  # warm-up: first launch of the kernel, one-time scratch allocation
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    te = td = 0.0
    for _ in range(reps):
        ev[0].record(); enc.run(pcm, img, None); ev[1].record(); dec.run(img, out); ev[2].record()
        torch.cuda.synchronize()
        te += ev[0].elapsed_time(ev[1]); td += ev[1].elapsed_time(ev[2])
    n = streams * samples * ch
    return dict(config=name, streams=streams, samples_per_ch=samples, channels=ch, bits=bits, trials=trials,
                encode_ms=round(te / reps, 4), decode_ms=round(td / reps, 4),
                encode_msps=round(n / (te / reps) / 1e3, 1), decode_msps=round(n / (td / reps) / 1e3, 1))


def host_api(eng, streams, samples, ch, bits, reps=5):
    """PCIe-inclusive rate of the host-memory convenience calls (stage + kernel + copy back)."""
    import time
    import numpy as np
    pcm = synth_pcm(min(streams, 500), samples, ch, seed=7)
    pcm = np.concatenate([pcm] * (-(-streams // pcm.shape[0])))[:streams]
    param = make_parameter(ch, bits, 1024, 48000, False, 0)
    rows = [pcm[i] for i in range(streams)]
    imgs = eng.encode_host(rows, param)
    te = td = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); imgs = eng.encode_host(rows, param); t1 = time.perf_counter()
        eng.decode_host(imgs); t2 = time.perf_counter()
        te, td = min(te, t1 - t0), min(td, t2 - t1)
    n = streams * samples * ch
    return dict(config="host API (pageable memory, incl. H2D/D2H) %d stereo x1 block" % streams, encode_ms=round(te * 1e3, 3),
                decode_ms=round(td * 1e3, 3), encode_msps=round(n / te / 1e6, 1), decode_msps=round(n / td / 1e6, 1))


def pinned_end_to_end(eng, streams, samples, ch, bits, reps=20):
    """Pinned host buffers -> H2D -> kernel -> D2H, all asynchronous on the engine's stream (what a
    caller that owns pinned memory and device plans gets; no staging copies, no plan creation)."""
    param = make_parameter(ch, bits, 1024, 48000, False, 0)
    h_pcm = torch.from_numpy(synth_pcm(min(streams, 500), samples, ch, seed=7)).repeat((-(-streams // 500), 1, 1))[:streams].contiguous().pin_memory()
    enc = eng.uniform_encode_plan(param, streams, samples)
    d_pcm = torch.empty_like(h_pcm, device="cuda")
    d_img = torch.zeros((streams, enc.stride), dtype=torch.uint8, device="cuda")
    h_img = torch.empty((streams, enc.stride), dtype=torch.uint8).pin_memory()
    h_out = torch.empty_like(h_pcm).pin_memory()
    d_out = torch.empty_like(d_pcm)
    d_pcm.copy_(h_pcm, non_blocking=True)
    enc.run(d_pcm, d_img, None)
    torch.cuda.synchronize()
    hd = parse_header(bytes(d_img[0, :31].cpu().numpy()))
    dec = eng.uniform_decode_plan(hd, streams, enc.stride, enc.image_size)
    dec.run(d_img, d_out)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    te = td = 0.0
    for _ in range(reps):
        ev[0].record()
        d_pcm.copy_(h_pcm, non_blocking=True)
        enc.run(d_pcm, d_img, None)
        h_img.copy_(d_img, non_blocking=True)
        ev[1].record()
        d_img.copy_(h_img, non_blocking=True)
        dec.run(d_img, d_out)
        h_out.copy_(d_out, non_blocking=True)
        ev[2].record()
        torch.cuda.synchronize()
        te += ev[0].elapsed_time(ev[1]); td += ev[1].elapsed_time(ev[2])
    n = streams * samples * ch
    return dict(config="end to end from pinned memory (H2D + kernel + D2H) %d stereo x1 block" % streams,
                encode_ms=round(te / reps, 4), decode_ms=round(td / reps, 4),
                encode_msps=round(n / (te / reps) / 1e3, 1), decode_msps=round(n / (td / reps) / 1e3, 1))


def main():
    eng = Engine(0)
    torch.cuda.set_stream(eng.stream)
    rows = [run(eng, "cfg2 1000 stereo x1 block 4-bit", 1000, 992, 2, 4),
            run(eng, "cfg2(ii) 1000 stereo x16 blocks 4-bit", 1000, 992 * 16, 2, 4, reps=3),
            run(eng, "cfg2(iii) 1 stereo stream x1000 blocks", 1, 992 * 1000, 2, 4, reps=2),
            run(eng, "cfg4 10000 x 8ch 3-bit", 10000, 292, 8, 3),
            run(eng, "cfg4 10000 x 8ch 2-bit", 10000, 444, 8, 2),
            run(eng, "cfg4ref 40000 stereo 3-bit", 40000, 1316, 2, 3),
            run(eng, "cfg5 shard 1250 files x10 blocks", 1250, 9920, 2, 4, reps=5),
            run(eng, "cfg5 all 10000 files x10 blocks", 10000, 9920, 2, 4, reps=3),
            run(eng, "cfg2 t=2", 1000, 992, 2, 4, trials=2),
            pinned_end_to_end(eng, 1000, 992, 2, 4),
            host_api(eng, 1000, 992, 2, 4)]
    for r in rows:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
