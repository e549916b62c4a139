"""Python mirror of the batched C-ABI (include/aad_hip.h).

torch is plumbing only: it owns the device tensors and the stream that are handed to the C
library as raw pointers.  All codec work happens in libaad_hip.so's HIP kernels; if the library
or a GPU is missing every call here raises - there is no fallback path.
"""
import ctypes as C

import numpy as np

from .capi import (AADApiResult, AADHeaderInfo, ApiError, ERROR_STATS_DTYPE, LANE_MAPPINGS, LANE_STATE_DTYPE,
                   OPTION_COMPARE_ORDER, OPTION_LANE_MAPPING, OPTION_STAGING_THREADS, OPTION_TILE_KBYTES, OPTION_TRIAL_LANES, RECONSTRUCT_DECODED, RECONSTRUCT_RESIDUAL,
                   STREAM_DESC_DTYPE, TRIAL_LANES, load_library, make_parameter)


def _check(where, rc):
    if rc != AADApiResult.OK:
        raise ApiError(where, rc)


def _round_up(v, a):
    return (v + a - 1) // a * a


class Engine:
    """One HIP context (device + stream).  By default it rides on torch's current stream so that
    torch.cuda.Event timing and tensor lifetimes line up with the kernels."""

    def __init__(self, device=0, stream="torch", lib=None):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("aad_amd.Engine needs a HIP device (torch.cuda.is_available() is False)")
        self.torch = torch
        self.lib = lib or load_library()
        self.device = int(device)
        torch.cuda.set_device(self.device)
        # The C context launches on an explicit stream.  Use torch's current stream when it is a
        # real one; the legacy null stream cannot be named through a pointer, so in that case the
        # engine owns a torch side stream and orders it against the caller's stream around every
        # run (see _enter/_exit).  `with torch.cuda.stream(engine.stream):` avoids even that.
        if stream == "torch":
            cur = torch.cuda.current_stream(self.device)
            self.stream = cur if cur.cuda_stream != 0 else torch.cuda.Stream(self.device)
        elif stream is None:
            self.stream = torch.cuda.Stream(self.device)
        else:
            self.stream = stream  # a torch.cuda.Stream
        self._ctx = C.c_void_p()
        _check("AADHip_ContextCreate",
               self.lib.AADHip_ContextCreate(self.device, self.stream.cuda_stream, C.byref(self._ctx)))

    def _enter(self):
        cur = self.torch.cuda.current_stream(self.device)
        if cur.cuda_stream != self.stream.cuda_stream:
            self.stream.wait_stream(cur)
        return cur

    def _exit(self, cur):
        if cur.cuda_stream != self.stream.cuda_stream:
            cur.wait_stream(self.stream)

    def close(self):
        if self._ctx:
            self.lib.AADHip_ContextDestroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def signal_next(self, stop, start=None):
        """stop / start: HipEvent (or None) - recorded when the work of the NEXT plan run on this engine is done / starts,
        carried by the run's own kernel dispatch instead of packets around it (AADHip_ContextSignalNextRun)"""
        _check("AADHip_ContextSignalNextRun",
               self.lib.AADHip_ContextSignalNextRun(self._ctx, start.handle if start is not None else None,
                                                    stop.handle if stop is not None else None))

    def last_error(self):
        return (self.lib.AADHip_ContextLastError(self._ctx) or b"").decode()

    def set_mapping(self, mapping="auto", trial_lanes=None):
        """Force a lane mapping ("auto", "dense", "quad", "quad-fused") and, optionally, the trial
        search's lane layout ("dual", "single") for every later run of this context
        (AADHip_ContextSetOption; the environment is only read when the context is created)."""
        _check("AADHip_ContextSetOption",
               self.lib.AADHip_ContextSetOption(self._ctx, OPTION_LANE_MAPPING, LANE_MAPPINGS[mapping or "auto"]))
        if trial_lanes is not None:
            _check("AADHip_ContextSetOption",
                   self.lib.AADHip_ContextSetOption(self._ctx, OPTION_TRIAL_LANES, TRIAL_LANES[trial_lanes]))

    def set_staging_threads(self, threads=0):
        """Threads that copy between caller buffers and the pinned staging blocks in the host-memory
        entry points: 0 = by core count, 1 = the calling thread alone, up to 8."""
        _check("AADHip_ContextSetOption",
               self.lib.AADHip_ContextSetOption(self._ctx, OPTION_STAGING_THREADS, int(threads)))

    def set_tile_kbytes(self, kbytes=0):
        """Tile budget (KiB) of the host-memory entry points; 0 = built in.  Results do not depend on it."""
        _check("AADHip_ContextSetOption",
               self.lib.AADHip_ContextSetOption(self._ctx, OPTION_TILE_KBYTES, int(kbytes)))

    def set_compare_order(self, sequential=False):
        """fp64 summation order behind the reconstruction modes' statistics: False = tree on the device with the
        reference's order taken only next to a rounding boundary of the printed line, True = always the reference's order
        (bit-identical doubles; AAD_HIP_OPTION_COMPARE_ORDER)"""
        _check("AADHip_ContextSetOption",
               self.lib.AADHip_ContextSetOption(self._ctx, OPTION_COMPARE_ORDER, 1 if sequential else 0))

    def synchronize(self):
        _check("AADHip_ContextSynchronize", self.lib.AADHip_ContextSynchronize(self._ctx))

    def encoded_size(self, param, num_samples):
        return int(self.lib.AADHip_CalculateEncodedSize(C.byref(param), num_samples))

    # ---- plans ---------------------------------------------------------------------------
    def encode_plan(self, param, descs):
        descs = np.ascontiguousarray(descs, dtype=STREAM_DESC_DTYPE)
        plan = C.c_void_p()
        _check("AADHip_EncodePlanCreate",
               self.lib.AADHip_EncodePlanCreate(self._ctx, C.byref(param), len(descs), descs.ctypes.data, C.byref(plan)))
        return EncodePlan(self, plan, param, descs)

    def decode_plan(self, header, descs, has_file_header=True):
        descs = np.ascontiguousarray(descs, dtype=STREAM_DESC_DTYPE)
        plan = C.c_void_p()
        _check("AADHip_DecodePlanCreate",
               self.lib.AADHip_DecodePlanCreate(self._ctx, C.byref(header), 1 if has_file_header else 0, len(descs),
                                                descs.ctypes.data, C.byref(plan)))
        return DecodePlan(self, plan, header, descs)

    # ---- uniform batches (every stream the same length) -----------------------------------
    def uniform_encode_plan(self, param, num_streams, num_samples):
        """Stream table for a [streams, samples, channels] int16 tensor and a [streams, stride]
        uint8 output; stride = encoded size rounded up to 64 bytes (images on 64-byte boundaries let the
        dense stereo encoder store whole granules - aad_encode.hip.h run_block)."""
        size = self.encoded_size(param, num_samples)
        if size == 0:
            raise ApiError("AADHip_CalculateEncodedSize", AADApiResult.INVALID_FORMAT)
        stride = _round_up(size, 64)
        d = np.zeros(num_streams, dtype=STREAM_DESC_DTYPE)
        i = np.arange(num_streams, dtype=np.uint64)
        d["pcm_offset"] = i * np.uint64(num_samples * param.num_channels)
        d["data_offset"] = i * np.uint64(stride)
        d["data_size"] = stride
        d["num_samples"] = num_samples
        plan = self.encode_plan(param, d)
        plan.image_size, plan.stride = size, stride
        return plan

    def uniform_decode_plan(self, header, num_streams, stride, image_size):
        d = np.zeros(num_streams, dtype=STREAM_DESC_DTYPE)
        i = np.arange(num_streams, dtype=np.uint64)
        d["pcm_offset"] = i * np.uint64(header.num_samples * header.num_channels)
        d["data_offset"] = i * np.uint64(stride)
        d["data_size"] = image_size
        d["num_samples"] = header.num_samples
        return self.decode_plan(header, d, True)

    def encode_uniform(self, pcm, param, state=None):
        """pcm: int16 cuda tensor [streams, samples, channels] -> uint8 tensor [streams, stride]
        (each row starts with a complete .aad image of plan.image_size bytes)."""
        torch = self.torch
        streams, samples, ch = pcm.shape
        assert ch == param.num_channels and pcm.dtype == torch.int16 and pcm.is_contiguous()
        plan = self.uniform_encode_plan(param, streams, samples)
        out = torch.zeros((streams, plan.stride), dtype=torch.uint8, device=pcm.device)
        plan.run(pcm, out, state)
        return out, plan.image_size

    def decode_uniform(self, data, image_size):
        """data: uint8 cuda tensor [streams, stride] of same-format images -> int16 [streams, samples, channels]"""
        torch = self.torch
        head = bytes(data[0, :31].cpu().numpy())
        header = parse_header(head)
        plan = self.uniform_decode_plan(header, data.shape[0], data.shape[1], image_size)
        pcm = torch.zeros((data.shape[0], header.num_samples, header.num_channels), dtype=torch.int16, device=data.device)
        plan.run(data, pcm)
        return pcm, header

    # ---- reconstruction modes (the reference CLI's -r / -g / -c, src/main.c:275-503) ----------
    def reconstruct_uniform(self, pcm, param, residual=False, want_stats=True):
        """pcm: int16 cuda tensor [streams, samples, channels] -> (out int16 tensor of the same
        shape, stats float64 tensor [streams, 3] = RMSE, MSD, MaxAE or None).  Encode, decode and
        the comparison run back to back on the device; the images stay in a scratch tensor."""
        torch = self.torch
        streams, samples, ch = pcm.shape
        assert ch == param.num_channels and pcm.dtype == torch.int16 and pcm.is_contiguous()
        size = self.encoded_size(param, samples)
        if size == 0:
            raise ApiError("AADHip_CalculateEncodedSize", AADApiResult.INVALID_FORMAT)
        stride = _round_up(size, 16)
        d = np.zeros(streams, dtype=STREAM_DESC_DTYPE)
        i = np.arange(streams, dtype=np.uint64)
        d["pcm_offset"] = i * np.uint64(samples * ch)
        d["data_offset"] = i * np.uint64(stride)
        d["data_size"] = stride
        d["num_samples"] = samples
        plan = C.c_void_p()
        _check("AADHip_ReconstructPlanCreate",
               self.lib.AADHip_ReconstructPlanCreate(self._ctx, C.byref(param), streams, d.ctypes.data, C.byref(plan)))
        try:
            images = torch.empty((streams, stride), dtype=torch.uint8, device=pcm.device)
            out = torch.empty_like(pcm)
            stats = torch.empty((streams, 3), dtype=torch.float64, device=pcm.device) if want_stats else None
            cur = self._enter()
            _check("AADHip_ReconstructPlanRun",
                   self.lib.AADHip_ReconstructPlanRun(plan, pcm.data_ptr(), images.data_ptr(), out.data_ptr(),
                                                      RECONSTRUCT_RESIDUAL if residual else RECONSTRUCT_DECODED,
                                                      stats.data_ptr() if want_stats else None))
            self._exit(cur)
        finally:
            self.lib.AADHip_ReconstructPlanDestroy(plan)  # synchronises the stream first
        return out, stats

    def reconstruct_host(self, pcm_list, param, residual=False, want_pcm=True, want_stats=True):
        """pcm_list: int16 arrays [samples, channels] -> (list of int16 arrays or None,
        ERROR_STATS_DTYPE array [streams] or None) through AADHip_ReconstructBatch."""
        n = len(pcm_list)
        pcm_list = [np.ascontiguousarray(p, dtype=np.int16) for p in pcm_list]
        nsamp = np.array([p.shape[0] for p in pcm_list], dtype=np.uint32)
        outs = [np.zeros_like(p) for p in pcm_list] if want_pcm else None
        stats = np.zeros(n, dtype=ERROR_STATS_DTYPE) if want_stats else None
        pp = (C.c_void_p * n)(*[p.ctypes.data for p in pcm_list])
        op = (C.c_void_p * n)(*[o.ctypes.data for o in outs]) if want_pcm else None
        _check("AADHip_ReconstructBatch",
               self.lib.AADHip_ReconstructBatch(self._ctx, C.byref(param), n, pp, nsamp.ctypes.data,
                                                RECONSTRUCT_RESIDUAL if residual else RECONSTRUCT_DECODED, op,
                                                stats.ctypes.data if want_stats else None))
        return outs, stats

    # ---- host-memory batches ----------------------------------------------------------------
    def encode_host(self, pcm_list, param, state=None):
        """pcm_list: list of int16 arrays [samples, channels] -> list of bytes.  state: optional
        LANE_STATE_DTYPE array [streams * channels], updated in place."""
        n = len(pcm_list)
        pcm_list = [np.ascontiguousarray(p, dtype=np.int16) for p in pcm_list]
        nsamp = np.array([p.shape[0] for p in pcm_list], dtype=np.uint32)
        caps = np.array([max(self.encoded_size(param, int(s)), 64) for s in nsamp], dtype=np.uint64)
        outs = [np.zeros(int(c), dtype=np.uint8) for c in caps]
        sizes = np.zeros(n, dtype=np.uint64)
        pp = (C.c_void_p * n)(*[p.ctypes.data for p in pcm_list])
        op = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        sp = state.ctypes.data if state is not None else None
        _check("AADHip_EncodeBatch",
               self.lib.AADHip_EncodeBatch(self._ctx, C.byref(param), n, pp, nsamp.ctypes.data, op,
                                           caps.ctypes.data, sizes.ctypes.data, sp))
        return [o[: int(s)].tobytes() for o, s in zip(outs, sizes)]

    def decode_host(self, images):
        """images: list of bytes (one format) -> list of int16 arrays [samples, channels]"""
        n = len(images)
        bufs = [np.frombuffer(b, dtype=np.uint8) for b in images]
        heads = [parse_header(bytes(b[:31])) for b in images]
        pcms = [np.zeros((h.num_samples, h.num_channels), dtype=np.int16) for h in heads]
        sizes = np.array([len(b) for b in bufs], dtype=np.uint64)
        caps = np.array([h.num_samples for h in heads], dtype=np.uint32)
        got = np.zeros(n, dtype=np.uint32)
        dp = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
        pp = (C.c_void_p * n)(*[p.ctypes.data for p in pcms])
        _check("AADHip_DecodeBatch",
               self.lib.AADHip_DecodeBatch(self._ctx, n, dp, sizes.ctypes.data, pp, caps.ctypes.data, got.ctypes.data))
        return pcms


class EncodePlan:
    def __init__(self, engine, handle, param, descs):
        self.engine, self.handle, self.param, self.descs = engine, handle, param, descs
        self.image_size = self.stride = None

    def run(self, pcm, data, state=None, ordered=True):
        """pcm: int16 cuda tensor, data: uint8 cuda tensor, state: int32 cuda tensor [lanes, 10] or None.
        ordered=False: launch on the engine's stream without ordering it against torch's current stream
        (the caller orders the streams itself, with events - see bench.py's pipelined step)."""
        sp = state.data_ptr() if state is not None else None
        cur = self.engine._enter() if ordered else None
        _check("AADHip_EncodePlanRun",
               self.engine.lib.AADHip_EncodePlanRun(self.handle, pcm.data_ptr(), data.data_ptr(), sp))
        if ordered:
            self.engine._exit(cur)

    def close(self):
        if self.handle:
            self.engine.lib.AADHip_EncodePlanDestroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DecodePlan:
    def __init__(self, engine, handle, header, descs):
        self.engine, self.handle, self.header, self.descs = engine, handle, header, descs

    def run(self, data, pcm, ordered=True):
        cur = self.engine._enter() if ordered else None
        _check("AADHip_DecodePlanRun",
               self.engine.lib.AADHip_DecodePlanRun(self.handle, data.data_ptr(), pcm.data_ptr()))
        if ordered:
            self.engine._exit(cur)

    def close(self):
        if self.handle:
            self.engine.lib.AADHip_DecodePlanDestroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipEvent:
    """A hipEvent_t of our own (timing disabled): what AADHip_ContextSignalNextRun takes and hipStreamWaitEvent waits for.
    torch.cuda.Event cannot wrap a foreign handle and creates its own lazily, hence the few runtime calls made directly."""
    _hip = None

    @classmethod
    def runtime(cls):
        if cls._hip is None:
            # THE runtime this process already runs on (torch's / libaad_hip.so's): a second copy would not know our streams
            path = "libamdhip64.so"
            try:
                with open("/proc/self/maps") as maps:
                    for line in maps:
                        if "libamdhip64.so" in line:
                            path = line.split()[-1]
                            break
            except OSError:
                pass
            hip = C.CDLL(path)
            hip.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
            hip.hipEventDestroy.argtypes = [C.c_void_p]
            hip.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
            hip.hipEventSynchronize.argtypes = [C.c_void_p]
            hip.hipEventQuery.argtypes = [C.c_void_p]
            hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
            cls._hip = hip
        return cls._hip

    def __init__(self, timing=False):
        self.handle = C.c_void_p()
        rc = self.runtime().hipEventCreateWithFlags(C.byref(self.handle), 0x0 if timing else 0x2)  # hipEventDisableTiming
        if rc != 0:
            raise RuntimeError("hipEventCreateWithFlags failed (%d)" % rc)

    def wait_on(self, stream):
        """stream: a torch.cuda.Stream - everything queued on it afterwards waits for this event"""
        rc = self.runtime().hipStreamWaitEvent(C.c_void_p(stream.cuda_stream), self.handle, 0)
        if rc != 0:
            raise RuntimeError("hipStreamWaitEvent failed (%d)" % rc)

    def elapsed_ms(self, stop):
        """milliseconds from this (start) event to `stop`, both created with timing=True and both complete"""
        ms = C.c_float()
        rc = self.runtime().hipEventElapsedTime(C.byref(ms), self.handle, stop.handle)
        if rc != 0:
            raise RuntimeError("hipEventElapsedTime failed (%d)" % rc)
        return float(ms.value)

    def synchronize(self):
        rc = self.runtime().hipEventSynchronize(self.handle)
        if rc != 0:
            raise RuntimeError("hipEventSynchronize failed (%d)" % rc)

    def close(self):
        if self.handle:
            self.runtime().hipEventDestroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class EncodeDecodePipeline:
    """Encode on one context, decode on another: the encode of step k+1 runs while step k decodes.

    On a batch that fills only part of the chip (BASELINE's 1000 one-block streams occupy 125 of 256
    CUs with either kernel) the two kernels of consecutive steps share the device instead of taking
    turns.  Two streams ordered by events; the .aad images go through a ring of buffers so that an
    encode never overwrites what a decode still reads (one wait per half ring: the decode stream runs
    in order).  The "encoded" events ride on the encode kernels' own dispatch packets (Engine.signal_next): an
    event recorded BEHIND every encode costs the encode queue 3 us per step (tools/experiments/launch_gap_probe.py).
    Every step encodes its whole batch and decodes exactly what it encoded; `pcm` and `out`
    are the caller's and must stay untouched until the step's kernels have run."""

    def __init__(self, enc_engine, dec_engine, param, streams, samples, ring=8):
        assert ring >= 2 and ring % 2 == 0
        assert enc_engine.stream.cuda_stream != dec_engine.stream.cuda_stream, "the two contexts need streams of their own"
        torch = enc_engine.torch
        self.torch, self.ring, self.k = torch, ring, 0
        self.enc_engine, self.dec_engine = enc_engine, dec_engine
        self.enc = enc_engine.uniform_encode_plan(param, streams, samples)
        self.images = [torch.zeros((streams, self.enc.stride), dtype=torch.uint8, device="cuda:%d" % enc_engine.device)
                       for _ in range(ring)]
        # the header comes from a first encode of silence: the decode plan needs the block geometry
        zero = torch.zeros((streams, samples, param.num_channels), dtype=torch.int16, device=self.images[0].device)
        self.enc.run(zero, self.images[0])
        torch.cuda.synchronize()
        self.header = parse_header(bytes(self.images[0][0, :31].cpu().numpy()))
        self.dec = dec_engine.uniform_decode_plan(self.header, streams, self.enc.stride, self.enc.image_size)
        self.encoded = [HipEvent() for _ in range(ring)]
        self.decoded = [torch.cuda.Event() for _ in range(ring)]

    def step(self, pcm, out, timing=None):
        """timing: four HipEvent(timing=True) -> start / end of the encode kernel, start / end of the decode kernel, carried by
        the kernels' own dispatches like the ordering events (elapsed_ms between a pair = the kernel's own duration)"""
        k, ring = self.k, self.ring
        self.k += 1
        b = k % ring
        s_enc, s_dec = self.enc_engine.stream, self.dec_engine.stream
        if k >= ring and k % (ring // 2) == 0:  # covers the half ring of encodes that follows
            s_enc.wait_event(self.decoded[(k - ring // 2 - 1) % ring])
        encoded = self.encoded[b] if timing is None else timing[1]
        self.enc_engine.signal_next(encoded, start=None if timing is None else timing[0])
        self.enc.run(pcm, self.images[b], None, ordered=False)
        encoded.wait_on(s_dec)
        if timing is not None:
            self.dec_engine.signal_next(timing[3], start=timing[2])
        self.dec.run(self.images[b], out, ordered=False)
        self.decoded[b].record(s_dec)
        return self.images[b]

    def close(self):
        self.torch.cuda.synchronize()
        self.enc.close()
        self.dec.close()
        for e in self.encoded:
            e.close()


def parse_header(data):
    """31-byte big-endian file header (reference src/aad_decoder.c:99-170) -> AADHeaderInfo"""
    if len(data) < 31 or data[:4] != b"AAD\x00":
        raise ApiError("parse_header", AADApiResult.INVALID_FORMAT)
    be = lambda o, n: int.from_bytes(data[o:o + n], "big")
    return AADHeaderInfo(format_version=be(4, 4), codec_version=be(8, 4), num_channels=be(12, 2),
                         num_samples=be(14, 4), sampling_rate=be(18, 4), bits_per_sample=be(22, 2),
                         block_size=be(24, 2), num_samples_per_block=be(26, 4), ch_process_method=data[30])


__all__ = ["Engine", "EncodePlan", "DecodePlan", "EncodeDecodePipeline", "parse_header", "make_parameter", "LANE_STATE_DTYPE"]
