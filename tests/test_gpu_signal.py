"""AADHip_ContextSignalNextRun (include/aad_hip.h): the event a plan run carries on its own kernel dispatch.
 - the event completes when the run's output is in memory: a second stream that waits for it (and nothing else) reads the
   finished images / PCM - against the oracle;
 - one-shot: the run after the signalled one does not touch the event;
 - empty plans and reconstruction runs (several kernels) record it too;
 - a context created under ROC_SYSTEM_SCOPE_SIGNAL=0 refuses the call (the one known way to hang a caller);
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


def test_signal_on_reconstruction_run(torch_mod):
    """AADHip_ReconstructPlanRun is several kernels (encode, decode, compare, finish): `start` is recorded in front of the
    first, `stop` behind the last (aad_hip_engine.hip, ReconstructPlanRun) - a second stream that waits for `stop` alone
    reads the finished output and statistics; the nested encode / decode runs neither consume nor re-record the events,
    and the run after leaves them untouched."""
    import ctypes as C
    torch = torch_mod
    from aad_amd.engine import RECONSTRUCT_DECODED, Engine, HipEvent, _check
    e = Engine(0, stream=torch.cuda.Stream())
    other = torch.cuda.Stream()
    try:
        streams, samples, ch = 300, 2500, 2
        param = make_parameter(ch, 4, 1024, 48000, False, 1)
        host = synth_pcm(streams, samples, ch, seed=77)
        pcm = torch.from_numpy(host).cuda()
        size = e.encoded_size(param, samples)
        stride = -(-size // 16) * 16
        d = np.zeros(streams, dtype=STREAM_DESC_DTYPE)
        i = np.arange(streams, dtype=np.uint64)
        d["pcm_offset"], d["data_offset"], d["data_size"], d["num_samples"] = i * np.uint64(samples * ch), i * np.uint64(stride), stride, samples
        plan = C.c_void_p()
        _check("AADHip_ReconstructPlanCreate", e.lib.AADHip_ReconstructPlanCreate(e._ctx, C.byref(param), streams, d.ctypes.data, C.byref(plan)))
        try:
            images = torch.empty((streams, stride), dtype=torch.uint8, device="cuda")
            out = torch.zeros_like(pcm)
            stats = torch.zeros((streams, 3), dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            start, stop = HipEvent(timing=True), HipEvent(timing=True)
            e.signal_next(stop, start=start)
            _check("AADHip_ReconstructPlanRun", e.lib.AADHip_ReconstructPlanRun(plan, pcm.data_ptr(), images.data_ptr(), out.data_ptr(),
                                                                                RECONSTRUCT_DECODED, stats.data_ptr()))
            stop.wait_on(other)
            with torch.cuda.stream(other):
                out_copy, stats_copy = out.clone(), stats.clone()  # ordered behind the run by `stop` alone
            other.synchronize()
            first = start.elapsed_ms(stop)
            assert 0.0 < first < 1000.0
            for s in (0, 1, streams - 1):
                img = ob.encode(host[s], 4, 1024, 48000, False, 1)
                y = ob.decode(img)[0]
                assert np.array_equal(out_copy[s].cpu().numpy(), y)
                want = ob.error_stats(host[s], y)
                got = stats_copy[s].cpu().numpy()
                assert np.allclose(got, want, rtol=1e-12, atol=0.0), (s, got, want)
            # the following run carries no events: the pair still holds the first run's interval
            out.zero_()
            _check("AADHip_ReconstructPlanRun", e.lib.AADHip_ReconstructPlanRun(plan, pcm.data_ptr(), images.data_ptr(), out.data_ptr(),
                                                                                RECONSTRUCT_DECODED, stats.data_ptr()))
            e.stream.synchronize()
            assert start.elapsed_ms(stop) == first
            assert torch.equal(out, out_copy)
        finally:
            e.lib.AADHip_ReconstructPlanDestroy(plan)
    finally:
        e.close()


def test_signal_refused_under_device_scope_signals(torch_mod, monkeypatch):
    """ROC_SYSTEM_SCOPE_SIGNAL=0: a context created under it refuses AADHip_ContextSignalNextRun (NG + a LastError that names the
    variable) instead of letting a cross-stream wait hang.  The variable is set AFTER this process's HIP runtime came up, so the
    runtime itself is unaffected - only the library's own check sees it; the hang itself is on record (round 3) and is not re-run."""
    torch = torch_mod
    from aad_amd.capi import ApiError
    from aad_amd.engine import Engine, HipEvent
    torch.cuda.synchronize()  # the runtime is up
    monkeypatch.setenv("ROC_SYSTEM_SCOPE_SIGNAL", "0")
    e = Engine(0, stream=torch.cuda.Stream())
    try:
        assert e.lib.AADHip_SignalNextRunSupported() == 0
        ev = HipEvent()
        with pytest.raises(ApiError):
            e.signal_next(ev)
        assert "ROC_SYSTEM_SCOPE_SIGNAL" in e.last_error()
        e.signal_next(None)  # withdrawing is always fine
        # and the context still works without signals
        param = make_parameter(2, 4, 1024, 48000, False, 0)
        host = synth_pcm(8, 992, 2, seed=3)
        img, size = e.encode_uniform(torch.from_numpy(host).cuda(), param)
        e.stream.synchronize()
        assert bytes(img[0, :size].cpu().numpy()) == ob.encode(host[0], 4, 1024, 48000, False, 0)
    finally:
        e.close()
    monkeypatch.delenv("ROC_SYSTEM_SCOPE_SIGNAL")
    e2 = Engine(0, stream=torch.cuda.Stream())
    try:
        assert e2.lib.AADHip_SignalNextRunSupported() == 1
        e2.signal_next(HipEvent())
        e2.signal_next(None)
    finally:
        e2.close()
