"""Measurement aid: print the encode kernel's phase log (cycles between marks of thread 0) from a
library built with -DAAD_PHASE_TIMING=1 (AAD_HIP_LIBRARY=<that .so>)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from aad_amd.capi import make_parameter
from aad_amd.engine import Engine
from aad_amd.synth import synth_pcm

engine = Engine(0)
lib = engine.lib
lib.AADHipDebug_ReadPhaseTimes.argtypes = [C.c_void_p, C.c_uint32]
lib.AADHipDebug_ReadPhaseTimes.restype = C.c_uint32
buf = np.zeros(512, dtype=np.uint64)
for blocks, trials in ((1, 0), (1, 2), (3, 2)):
    pcm = torch.from_numpy(synth_pcm(1000, 992 * blocks, 2, seed=1234)).cuda()
    param = make_parameter(2, 4, 1024, 48000, False, trials)
    plan = engine.uniform_encode_plan(param, 1000, 992 * blocks)
    img = torch.zeros((1000, plan.stride), dtype=torch.uint8, device="cuda")
    for rep in range(3):
        plan.run(pcm, img)
        torch.cuda.synchronize()
        n = lib.AADHipDebug_ReadPhaseTimes(buf.ctypes.data, 512)
    d = np.diff(buf[:n].astype(np.int64))
    print("blocks=%d trials=%d marks=%d  cycles between marks: %s" % (blocks, trials, n, " ".join(str(int(x)) for x in d)), flush=True)
# decode phases: kernel entry | tables staged | header parsed, first four samples out | chunk loop | tail
from aad_amd.engine import parse_header
pcm = torch.from_numpy(synth_pcm(1000, 992, 2, seed=1234)).cuda()
param = make_parameter(2, 4, 1024, 48000, False, 0)
plan = engine.uniform_encode_plan(param, 1000, 992)
img = torch.zeros((1000, plan.stride), dtype=torch.uint8, device="cuda")
plan.run(pcm, img)
torch.cuda.synchronize()
lib.AADHipDebug_ReadPhaseTimes(buf.ctypes.data, 512)
hd = parse_header(bytes(img[0, :31].cpu().numpy()))
dplan = engine.uniform_decode_plan(hd, 1000, plan.stride, plan.image_size)
out = torch.zeros_like(pcm)
for rep in range(3):
    dplan.run(img, out)
    torch.cuda.synchronize()
    n = lib.AADHipDebug_ReadPhaseTimes(buf.ctypes.data, 512)
print("decode phases, cycles between marks:", " ".join(str(int(x)) for x in np.diff(buf[:n].astype(np.int64))), flush=True)
# split decoder (its own translation unit): entry | tables written | barrier | strand 1 | barrier |
# header parsed + first frames | first loads + prime | chunk loop | tail
try:
    lib.AADHipDebug_ReadSplitPhaseTimes.argtypes = [C.c_void_p, C.c_uint32]
    lib.AADHipDebug_ReadSplitPhaseTimes.restype = C.c_uint32
    for rep in range(3):
        dplan.run(img, out)
        torch.cuda.synchronize()
        n = lib.AADHipDebug_ReadSplitPhaseTimes(buf.ctypes.data, 512)
    print("split decode phases, cycles between marks:", " ".join(str(int(x)) for x in np.diff(buf[:n].astype(np.int64))), flush=True)
except AttributeError:
    pass
engine.close()
