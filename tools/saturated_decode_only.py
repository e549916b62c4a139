"""Measurement aid: the saturated dense decode alone (no encode launches in between), HIP events."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aad_amd.capi import make_parameter  # noqa: E402
from aad_amd.engine import Engine, parse_header  # noqa: E402
from aad_amd.synth import synth_pcm  # noqa: E402

streams, samples = 262144, 992
e = Engine(0)
torch.cuda.set_stream(e.stream)
param = make_parameter(2, 4, 1024, 48000, False, 0)
base = torch.from_numpy(synth_pcm(1000, samples, 2, seed=1234)).cuda()
pcm = base.repeat((263, 1, 1))[:streams].contiguous()
plan = e.uniform_encode_plan(param, streams, samples)
images = torch.zeros((streams, plan.stride), dtype=torch.uint8, device="cuda")
plan.run(pcm, images)
hd = parse_header(bytes(images[0, :31].cpu().numpy()))
dplan = e.uniform_decode_plan(hd, streams, plan.stride, plan.image_size)
out = torch.zeros_like(pcm)
res = {}
for what, fn in (("decode", lambda: dplan.run(images, out)), ("encode", lambda: plan.run(pcm, images))):
    fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)
    res[what + "_ms_median"] = round(t[len(t) // 2], 4)
    res[what + "_ms_min"] = round(t[0], 4)
print(json.dumps(res))
