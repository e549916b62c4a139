import ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from aad_amd.capi import make_parameter
from aad_amd.engine import Engine
from aad_amd.synth import synth_pcm
e = Engine(0); lib = e.lib
marks = hasattr(lib, "AADHipDebug_ReadPhaseTimes")
if marks:
    lib.AADHipDebug_ReadPhaseTimes.argtypes = [C.c_void_p, C.c_uint32]; lib.AADHipDebug_ReadPhaseTimes.restype = C.c_uint32
buf = np.zeros(512, dtype=np.uint64)
for ch, bits, samples, streams in ((8, 3, 292, 10000), (8, 3, 292, 8000), (8, 3, 292, 4000), (8, 3, 292, 1000), (8, 3, 292, 100), (8, 3, 584, 1000), (8, 2, 444, 10000), (8, 2, 444, 1000)):
    pcm = torch.from_numpy(synth_pcm(streams, samples, ch, seed=1234)).cuda()
    plan = e.uniform_encode_plan(make_parameter(ch, bits, 1024, 48000, False, 0), streams, samples)
    img = torch.zeros((streams, plan.stride), dtype=torch.uint8, device="cuda")
    for rep in range(3):
        plan.run(pcm, img); torch.cuda.synchronize()
        n = lib.AADHipDebug_ReadPhaseTimes(buf.ctypes.data, 512) if marks else 0
    ts = []
    for rep in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); plan.run(pcm, img); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print(ch, bits, samples, streams, "kernel us", round(sorted(ts)[len(ts) // 2] * 1e3, 1), "cycles between marks:", " ".join(str(int(x)) for x in np.diff(buf[:n].astype(np.int64))))
