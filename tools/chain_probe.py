"""Measurement aid: encode time of few long streams (the per-block chain), HIP events, device-resident.
usage: python tools/chain_probe.py [streams=1] [blocks=1000] [trials=0]"""
import json
import sys

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from aad_amd.capi import make_parameter
from aad_amd.engine import Engine
from aad_amd.synth import synth_pcm

streams = int(sys.argv[1]) if len(sys.argv) > 1 else 1
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
trials = int(sys.argv[3]) if len(sys.argv) > 3 else 0
e = Engine(0)
param = make_parameter(2, 4, 1024, 48000, False, trials)
pcm = torch.from_numpy(synth_pcm(streams, 992 * blocks, 2, seed=3)).cuda()
plan = e.uniform_encode_plan(param, streams, 992 * blocks)
img = torch.zeros((streams, plan.stride), dtype=torch.uint8, device="cuda")
best = 1e9
for rep in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    plan.run(pcm, img)
    b.record()
    torch.cuda.synchronize()
    best = min(best, a.elapsed_time(b))
print(json.dumps({"streams": streams, "blocks": blocks, "trials": trials, "encode_ms": round(best, 4),
                  "us_per_block": round(best * 1e3 / blocks, 3)}))
