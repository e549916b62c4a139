"""Experiment: the pipelined step captured in a HIP graph of `n` steps (two streams forked and joined by
events inside the capture) against the eager pipeline.  Prints us per step for both."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aad_amd.capi import make_parameter  # noqa: E402
from aad_amd.engine import Engine, EncodeDecodePipeline  # noqa: E402
from aad_amd.synth import synth_pcm  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
E = Engine(0, stream=torch.cuda.Stream(0))
D = Engine(0, stream=torch.cuda.Stream(0))
torch.cuda.set_stream(E.stream)
param = make_parameter(2, 4, 1024, 48000, False, 0)
pcm = torch.from_numpy(synth_pcm(1000, 992, 2, seed=1234)).cuda()
out = torch.zeros_like(pcm)
pipe = EncodeDecodePipeline(E, D, param, 1000, 992, ring=2 * n)
for _ in range(4 * n):
    pipe.step(pcm, out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(400):
    pipe.step(pcm, out)
torch.cuda.synchronize()
print("eager pipeline  us/step", round((time.perf_counter() - t0) / 400 * 1e6, 1))

# capture n steps: encodes chained on E, each decode on D after its encode, D joined back at the end
g = torch.cuda.CUDAGraph()
s_enc, s_dec = E.stream, D.stream
with torch.cuda.graph(g, stream=s_enc, capture_error_mode="relaxed"):
    ev = [torch.cuda.Event() for _ in range(n)]
    for k in range(n):
        pipe.enc.run(pcm, pipe.images[k], None, ordered=False)
        ev[k].record(s_enc)
        s_dec.wait_event(ev[k])
        pipe.dec.run(pipe.images[k], out, ordered=False)
    done = torch.cuda.Event()
    done.record(s_dec)
    s_enc.wait_event(done)
torch.cuda.synchronize()
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
reps = 400 // n
t0 = time.perf_counter()
for _ in range(reps):
    g.replay()
torch.cuda.synchronize()
print("graph of %d steps us/step" % n, round((time.perf_counter() - t0) / (reps * n) * 1e6, 1))
ref = out.clone()
pipe2_out = torch.zeros_like(pcm)
pipe.step(pcm, pipe2_out)
torch.cuda.synchronize()
print("graph output equals eager output:", bool(torch.equal(ref, pipe2_out)))
