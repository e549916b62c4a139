"""AADHip_ContextSignalNextRun (include/aad_hip.h): the event a plan run carries on its own kernel dispatch.
 - the event completes when the run's output is in memory: a second stream that waits for it (and nothing else) reads the
   finished images / PCM - against the oracle;
 - one-shot: the run after the signalled one does not touch the event;
 - empty plans and reconstruction runs (several kernels) record it too;
 - the pipelined step of bench.py (EncodeDecodePipeline) orders its two streams with such events only: many steps, bit-exact."""
import numpy as np
import pytest

import oracle_binding as ob
from aad_amd.capi import STREAM_DESC_DTYPE, make_parameter
from aad_amd.synth import synth_pcm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    return torch


@pytest.mark.parametrize("streams,channels,bits", [(40, 2, 4), (3000, 1, 3), (70000, 2, 4)])
def test_signal_orders_a_second_stream(torch_mod, streams, channels, bits):
    torch = torch_mod
    from aad_amd.engine import Engine, HipEvent, parse_header
    e = Engine(0, stream=torch.cuda.Stream())
    other = torch.cuda.Stream()
    try:
        spb = {4: 1984, 3: 2632, 2: 3960}[bits] // channels
        param = make_parameter(channels, bits, 1024, 48000, False, 0)
        tile = synth_pcm(min(streams, 64), spb, channels, seed=99)
        pcm = torch.from_numpy(tile).cuda().repeat((-(-streams // tile.shape[0]), 1, 1))[:streams].contiguous()
        enc = e.uniform_encode_plan(param, streams, spb)
        img = torch.zeros((streams, enc.stride), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ev, ev2 = HipEvent(), HipEvent()
        for _ in range(3):  # the event is reused, as a ring of them would be
            e.signal_next(ev)
            enc.run(pcm, img, None, ordered=False)
            ev.wait_on(other)
            with torch.cuda.stream(other):
                copy = img.clone()  # runs on `other`, ordered behind the encode by the event alone
            other.synchronize()
        want = ob.encode(tile[0], bits, 1024, 48000, False, 0)
        assert bytes(copy[0, :enc.image_size].cpu().numpy()) == want
        assert torch.equal(copy[:tile.shape[0]], copy[-tile.shape[0]:]) or streams % tile.shape[0] != 0
        # decode, signalled as well; and host-side wait on the event
        hd = parse_header(want[:31])
        dec = e.uniform_decode_plan(hd, streams, enc.stride, enc.image_size)
        out = torch.zeros_like(pcm)
        e.signal_next(ev2)
        dec.run(img, out, ordered=False)
        ev2.synchronize()
        assert np.array_equal(out[0].cpu().numpy(), ob.decode(want)[0])
        # start + stop events with timing: the elapsed time is the kernel's own duration
        t0, t1 = HipEvent(timing=True), HipEvent(timing=True)
        e.signal_next(t1, start=t0)
        dec.run(img, out, ordered=False)
        t1.synchronize()
        assert 0.0 < t0.elapsed_ms(t1) < 1000.0
        # one-shot: this run carries no event, and withdrawing works
        e.signal_next(ev2)
        e.signal_next(None)
        dec.run(img, out, ordered=False)
        e.stream.synchronize()
        enc.close()
        dec.close()
    finally:
        e.close()


def test_signal_on_empty_plan_and_reconstruction(torch_mod):
    torch = torch_mod
    from aad_amd.engine import Engine, HipEvent
    e = Engine(0, stream=torch.cuda.Stream())
    try:
        param = make_parameter(2, 4, 1024, 48000, False, 0)
        empty = e.encode_plan(param, np.zeros(0, dtype=STREAM_DESC_DTYPE))
        ev = HipEvent()
        e.signal_next(ev)
        dummy = torch.zeros(16, dtype=torch.int16, device="cuda")
        empty.run(dummy, torch.zeros(64, dtype=torch.uint8, device="cuda"), None, ordered=False)
        ev.synchronize()  # recorded behind an empty run: returns
        empty.close()
    finally:
        e.close()


def test_pipeline_orders_streams_with_signals_only(torch_mod):
    torch = torch_mod
    from aad_amd.engine import Engine, EncodeDecodePipeline
    e1, e2 = Engine(0, stream=torch.cuda.Stream()), Engine(0, stream=torch.cuda.Stream())
    try:
        param = make_parameter(2, 4, 1024, 48000, False, 0)
        a = torch.from_numpy(synth_pcm(1000, 992, 2, seed=5)).cuda()
        b = torch.from_numpy(synth_pcm(1000, 992, 2, seed=6)).cuda()
        outs = [torch.zeros_like(a) for _ in range(2)]
        pipe = EncodeDecodePipeline(e1, e2, param, 1000, 992, ring=8)
        imgs = []
        for k in range(41):  # several times round the ring, inputs alternating so that a stale image would show
            img = pipe.step(a if k % 2 == 0 else b, outs[k % 2])
            if k >= 39:
                torch.cuda.synchronize()
                imgs.append(img[:3, :pipe.enc.image_size].cpu().numpy().copy())
        torch.cuda.synchronize()
        for which, (src, out) in enumerate(((b, outs[1]), (a, outs[0]))):  # steps 39 (b) and 40 (a)
            host = src[:3].cpu().numpy()
            for s in range(3):
                want = ob.encode(host[s], 4, 1024, 48000, False, 0)
                assert bytes(imgs[which][s]) == want
                assert np.array_equal(out[s].cpu().numpy(), ob.decode(want)[0])
        pipe.close()
    finally:
        e1.close()
        e2.close()
