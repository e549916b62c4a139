"""Measurement aid: kernel times (HIP events, launches queued back to back, median of 30) of the
latency-bound BASELINE shapes that use the dense kernels: configs 2(ii) and 5 (decode), 4 (encode)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aad_amd.capi import make_parameter  # noqa: E402
from aad_amd.engine import Engine, parse_header  # noqa: E402
from aad_amd.synth import synth_pcm  # noqa: E402

e = Engine(0)
torch.cuda.set_stream(e.stream)


def timed(fn, reps=30):
    fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)
    return round(t[len(t) // 2] * 1e3, 2)


out = {}
for name, streams, samples, ch, bits in (("cfg2ii_1000x16", 1000, 992 * 16, 2, 4), ("cfg5_1250x10", 1250, 9920, 2, 4),
                                         ("cfg4_8ch_3bit", 10000, 292, 8, 3), ("cfg4_8ch_2bit", 10000, 444, 8, 2),
                                         ("mono4_20000x1", 20000, 2016, 1, 4)):
    param = make_parameter(ch, bits, 1024, 48000, False, 0)
    pcm = torch.from_numpy(synth_pcm(streams, samples, ch, seed=1234)).cuda()
    plan = e.uniform_encode_plan(param, streams, samples)
    img = torch.zeros((streams, plan.stride), dtype=torch.uint8, device="cuda")
    plan.run(pcm, img)
    torch.cuda.synchronize()
    hd = parse_header(bytes(img[0, :31].cpu().numpy()))
    dplan = e.uniform_decode_plan(hd, streams, plan.stride, plan.image_size)
    dec = torch.zeros_like(pcm)
    out[name] = {"encode_us": timed(lambda: plan.run(pcm, img)), "decode_us": timed(lambda: dplan.run(img, dec))}
    plan.close()
    dplan.close()
    del pcm, img, dec
print(json.dumps(out))
