import ctypes as C, sys, time, json
import numpy as np
sys.path.insert(0, ".")
import torch
from aad_amd.capi import make_parameter
from aad_amd.engine import Engine
from aad_amd.synth import synth_pcm
e = Engine(0)
lib, ctx = e.lib, e._ctx
for trials in (0, 2):
  param = make_parameter(2, 4, 1024, 48000, False, trials)
  samples = 992
  base = synth_pcm(1000, samples, 2, seed=1234)
  size = e.encoded_size(param, samples)
  import os
  for streams in [int(v) for v in os.environ.get("SWEEP_STREAMS", "500,1000,2000,3000,3300,3400,3600,4000,5000,6700,7000,10000,20000,50000,100000").split(",")]:
    rows = [np.ascontiguousarray(base[i % 1000]) for i in range(streams)]
    imgs = [np.zeros(size, dtype=np.uint8) for _ in range(streams)]
    decs = [np.zeros((samples, 2), dtype=np.int16) for _ in range(streams)]
    nsamp = np.full(streams, samples, dtype=np.uint32); caps = np.full(streams, size, dtype=np.uint64)
    sizes = np.zeros(streams, dtype=np.uint64); got = np.zeros(streams, dtype=np.uint32)
    pp = (C.c_void_p * streams)(*[r.ctypes.data for r in rows]); ip = (C.c_void_p * streams)(*[r.ctypes.data for r in imgs]); dp = (C.c_void_p * streams)(*[r.ctypes.data for r in decs])
    te, td = [], []
    for k in range(7):
        t0 = time.perf_counter()
        rc1 = lib.AADHip_EncodeBatch(ctx, C.byref(param), streams, pp, nsamp.ctypes.data, ip, caps.ctypes.data, sizes.ctypes.data, None)
        t1 = time.perf_counter()
        rc2 = lib.AADHip_DecodeBatch(ctx, streams, ip, sizes.ctypes.data, dp, nsamp.ctypes.data, got.ctypes.data)
        t2 = time.perf_counter()
        assert rc1 == 0 and rc2 == 0
        if k >= 2: te.append(t1 - t0); td.append(t2 - t1)
    me, md = sorted(te)[len(te)//2], sorted(td)[len(td)//2]
    n = streams * samples * 2
    print(json.dumps(dict(trials=trials, streams=streams, mb=round(streams*(samples*4+size)/1e6,1), encode_ms=round(me*1e3,3), decode_ms=round(md*1e3,3), encode_msps=round(n/me/1e6), decode_msps=round(n/md/1e6))), flush=True)
