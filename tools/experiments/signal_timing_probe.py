import sys, os
sys.path.insert(0, os.getcwd())
import torch
from aad_amd.capi import make_parameter
from aad_amd.engine import Engine, HipEvent, parse_header
from aad_amd.synth import synth_pcm
e = Engine(0, stream=torch.cuda.Stream())
param = make_parameter(2, 4, 1024, 48000, False, 0)
for streams in (1000, 40000):
    pcm = torch.from_numpy(synth_pcm(100, 992, 2, seed=1)).cuda().repeat((streams // 100, 1, 1)).contiguous()
    enc = e.uniform_encode_plan(param, streams, 992)
    img = torch.zeros((streams, enc.stride), dtype=torch.uint8, device="cuda")
    enc.run(pcm, img, None)
    torch.cuda.synchronize()
    hd = parse_header(bytes(img[0, :31].cpu().numpy()))
    dec = e.uniform_decode_plan(hd, streams, enc.stride, enc.image_size)
    out = torch.zeros_like(pcm)
    for mapping in ("auto", "dense", "quad", "quad-fused"):
        e.set_mapping(mapping)
        for rep in range(3):
            a, b = HipEvent(timing=True), HipEvent(timing=True)
            ta, tb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(e.stream):
                ta.record()
                e.signal_next(b, start=a)
                dec.run(img, out, ordered=False)
                tb.record()
            torch.cuda.synchronize()
            if rep == 2:
                print(streams, mapping, "attached %.4f ms  torch events around %.4f ms" % (a.elapsed_ms(b), ta.elapsed_time(tb)))
    e.set_mapping("auto")
