"""Measurement aid: decode time of N stereo one-block streams with the quad decoder in its split
(two-strand) and fused forms and with the dense mapping - where the crossovers are."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aad_amd.capi import make_parameter
from aad_amd.engine import Engine, parse_header
from aad_amd.synth import synth_pcm

engine = Engine(0)
torch.cuda.set_stream(engine.stream)
param = make_parameter(2, 4, 1024, 48000, False, 0)
base = torch.from_numpy(synth_pcm(1000, 992, 2, seed=1234)).cuda()
for streams in (500, 1000, 2000, 3000, 4000, 6000, 8000, 12000, 16000):
    pcm = base.repeat((-(-streams // 1000), 1, 1))[:streams].contiguous()
    engine.set_mapping("auto")
    plan = engine.uniform_encode_plan(param, streams, 992)
    img = torch.zeros((streams, plan.stride), dtype=torch.uint8, device="cuda")
    plan.run(pcm, img)
    torch.cuda.synchronize()
    hd = parse_header(bytes(img[0, :31].cpu().numpy()))
    row = dict(streams=streams, recurrences=2 * streams)
    for mapping in ("quad", "quad-fused", "dense"):
        engine.set_mapping(mapping)
        dplan = engine.uniform_decode_plan(hd, streams, plan.stride, plan.image_size)
        out = torch.zeros_like(pcm)
        for _ in range(3):
            dplan.run(img, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            dplan.run(img, out)
        e1.record()
        torch.cuda.synchronize()
        row[mapping + "_ms"] = round(e0.elapsed_time(e1) / 20, 4)
        dplan.close()
    print(json.dumps(row), flush=True)
engine.close()
