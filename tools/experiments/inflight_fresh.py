"""Two steps in flight, the four streams created FIRST in a fresh process (null stream + three new ones): how stable is the placement?
usage (gpurun): for i in 1 2 3 4 5; do python3 tools/experiments/inflight_fresh.py; done"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from aad_amd.capi import make_parameter
from aad_amd.engine import Engine, EncodeDecodePipeline
from aad_amd.synth import synth_pcm
depth = int(os.environ.get("DEPTH", "2"))
own = os.environ.get("OWN_STREAMS", "0") == "1"
param = make_parameter(2, 4, 1024, 48000, False, 0)
order = os.environ.get("ORDER", "")
if order.startswith("bench"):  # bench.py's order: first engine, the PCM upload, set_stream, then the other three
    e0 = Engine(0)
    pcm = torch.from_numpy(synth_pcm(1000, 992, 2, seed=1234)).cuda()
    torch.cuda.synchronize()
    if order != "bench-noset":
        torch.cuda.set_stream(e0.stream)
    engines = [(e0, Engine(0, stream=torch.cuda.Stream(0))), (Engine(0, stream=torch.cuda.Stream(0)), Engine(0, stream=torch.cuda.Stream(0)))]
else:
    engines = []
    for p in range(depth):
        e = Engine(0) if (p == 0 and not own) else Engine(0, stream=torch.cuda.Stream(0))
        engines.append((e, Engine(0, stream=torch.cuda.Stream(0))))
    pcm = torch.from_numpy(synth_pcm(1000, 992, 2, seed=1234)).cuda()
pipes = [EncodeDecodePipeline(e, d, param, 1000, 992, ring=32) for e, d in engines]
outs = [torch.zeros_like(pcm) for _ in pipes]
n = 1000 * 992 * 2
pre = os.environ.get("PRE", "")
if pre == "benchfn":  # bench.py's own measure() with the second pipeline
    import bench
    m = bench.measure(engines[0][0], torch, None, pcm, param, 20, 5, 1, 8, decode_engine=engines[0][1], repeats=5, max_repeats=5, more_pipelines=engines[1:])
    print("bench.measure: %.0f Msamples/s, steps in flight %d, identical %s" % (2.0 * n * 20 / m["wall_s"] / 1e6, m["steps_in_flight"], m["outputs_identical"]))
    sys.exit(0)
if pre == "ring16":
    for p_ in pipes:
        p_.close()
    pipes = [EncodeDecodePipeline(e, d, param, 1000, 992, ring=16) for e, d in engines]
if pre == "events":  # what bench.py's measure() does before: timed steps with events carried by the kernels' dispatches
    from aad_amd.engine import HipEvent
    for k in range(40):
        ev = [HipEvent(timing=True) for _ in range(4)]
        pipes[0].step(pcm, outs[0], ev)
    torch.cuda.synchronize()
if pre == "measure":
    import bench
    bench.measure(engines[0][0], torch, None, pcm, param, 20, 5, 1, 8, decode_engine=engines[0][1], repeats=3, max_repeats=3)
if pre == "serial":
    import bench
    bench.measure(engines[0][0], torch, None, pcm, param, 20, 3, 1, 8, repeats=3, max_repeats=3)
if pre == "setstream":
    torch.cuda.set_stream(engines[0][0].stream)
for k in range(10 * depth):
    pipes[k % depth].step(pcm, outs[k % depth])
res = []
for steps in (20, 20, 20, 200, 200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        pipes[k % depth].step(pcm, outs[k % depth])
    torch.cuda.synchronize()
    res.append(round(2.0 * n * steps / (time.perf_counter() - t0) / 1e6))
print("depth", depth, "own", own, "K=20 x3, K=200 x2:", res, "identical", all(bool(torch.equal(o, outs[0])) for o in outs))
