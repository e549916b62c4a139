import sys, os, json
sys.path.insert(0, os.getcwd())
import torch
from aad_amd.capi import make_parameter
from aad_amd.engine import Engine
from aad_amd.synth import synth_pcm
engine = Engine(0)
torch.cuda.set_stream(engine.stream)
for blocks in (1, 2, 4):
    pcm = torch.from_numpy(synth_pcm(1000, 992 * blocks, 2, seed=1234)).cuda()
    for trials in (0, 1, 2, 3):
        param = make_parameter(2, 4, 1024, 48000, False, trials)
        plan = engine.uniform_encode_plan(param, 1000, 992 * blocks)
        img = torch.zeros((1000, plan.stride), dtype=torch.uint8, device="cuda")
        for _ in range(3):
            plan.run(pcm, img)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            plan.run(pcm, img)
        e1.record()
        torch.cuda.synchronize()
        print(json.dumps(dict(blocks=blocks, trials=trials, encode_ms=round(e0.elapsed_time(e1) / 20, 4))), flush=True)
engine.close()
