#!/usr/bin/env python3
"""Experiment: the pipelined headline step (encode of step k + 1 beside the decode of step k) with the two contexts' streams
restricted to complementary halves of the chip (hipExtStreamCreateWithCUMask), against the unrestricted streams.
usage (GPU box): python3 tools/experiments/cumask_probe.py"""
import ctypes as C
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def masked_stream(torch, hip, words):
    s = C.c_void_p()
    arr = (C.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def run(torch, streams_pair, label, steps=200, regions=7, dec_mapping=None, enc_mapping=None):
    from aad_amd.capi import make_parameter
    from aad_amd.engine import Engine, EncodeDecodePipeline
    from aad_amd.synth import synth_pcm
    param = make_parameter(2, 4, 1024, 48000, False, 0)
    e1, e2 = Engine(0, stream=streams_pair[0]), Engine(0, stream=streams_pair[1])
    if dec_mapping:
        e2.set_mapping(dec_mapping)
    if enc_mapping:
        e1.set_mapping(enc_mapping)
    pcm = torch.from_numpy(synth_pcm(1000, 992, 2, seed=1234)).cuda()
    out = torch.zeros_like(pcm)
    pipe = EncodeDecodePipeline(e1, e2, param, 1000, 992, ring=16)
    for _ in range(50):
        pipe.step(pcm, out)
    torch.cuda.synchronize()
    times = []
    from aad_amd.engine import HipEvent
    ev = [HipEvent(timing=True) for _ in range(4)]
    enc_k, dec_k = [], []
    for _ in range(regions):
        import time
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            if k % 50 == 25:
                pipe.step(pcm, out, timing=ev)
                torch.cuda.synchronize()
                enc_k.append(ev[0].elapsed_ms(ev[1]))
                dec_k.append(ev[2].elapsed_ms(ev[3]))
            else:
                pipe.step(pcm, out)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) / steps * 1e3)
    assert torch.equal(out, pcm) is False or True
    print(json.dumps(dict(label=label, ms_per_step=round(statistics.median(times), 5), min=round(min(times), 5),
                          enc_kernel_ms=round(statistics.median(enc_k), 5), dec_kernel_ms=round(statistics.median(dec_k), 5))))
    pipe.close()
    e1.close()
    e2.close()


def main():
    import torch
    torch.cuda.init()
    hip = C.CDLL("libamdhip64.so")
    hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
    n = torch.cuda.get_device_properties(0).multi_processor_count
    words = (n + 31) // 32
    print("CUs", n)
    run(torch, (torch.cuda.Stream(), torch.cuda.Stream()), "unmasked")
    if "--mappings" in sys.argv:
        for dm in ("dense", "quad-fused", "quad"):
            run(torch, (torch.cuda.Stream(), torch.cuda.Stream()), "decode mapping " + dm, dec_mapping=dm)
        run(torch, (torch.cuda.Stream(), torch.cuda.Stream()), "unmasked again")
        return
    full = (1 << n) - 1
    def split(mask):
        return [(mask >> (32 * i)) & 0xFFFFFFFF for i in range(words)]
    low = (1 << (n // 2)) - 1
    even = sum(1 << i for i in range(0, n, 2))
    for label, a in (("low half / high half", low), ("even / odd", even), ("low 5/8 / high 3/8", (1 << (n * 5 // 8)) - 1)):
        sa, sb = masked_stream(torch, hip, split(a)), masked_stream(torch, hip, split(full & ~a))
        run(torch, (sa, sb), label)
    # both kernels on the SAME half of the chip (fewer busy CUs - higher clocks? - but the encode waves share their CUs)
    for label, a in (("both on the low half", low), ("both on the low 5/8", (1 << (n * 5 // 8)) - 1), ("both on the low 3/4", (1 << (n * 3 // 4)) - 1)):
        sa, sb = masked_stream(torch, hip, split(a)), masked_stream(torch, hip, split(a))
        run(torch, (sa, sb), label)
    run(torch, (torch.cuda.Stream(), torch.cuda.Stream()), "unmasked again")


if __name__ == "__main__":
    main()
