"""Host-side (no GPU) behaviour of the drop-in C API in aad_amd/libaad_hip.so: symbol export,
geometry known-answers, header byte layout, handle lifecycle and the error-code matrix of
reference test/test_aad_encoder.c:24-334 and test/test_aad_decoder.c:33-254."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import aad_amd
from aad_amd import AADApiResult as R
from aad_amd.capi import (HIP_SYMBOLS, LEGACY_SYMBOLS, SYNTH_SYMBOLS, WAV_SYMBOLS, AADEncodeParameter, AADHeaderInfo, AADWavInfo,
                          make_parameter)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    return aad_amd.load_library()


def test_library_exports_every_declared_symbol(lib):
    declared = set()
    for h in ("aad_api.h", "aad_hip.h", "aad_wav.h", "aad_synth.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        declared |= set(re.findall(r"\b(AAD(?:Encoder|Decoder|Hip|Wav|Synth)_[A-Za-z0-9]+)\s*\(", text))
    assert declared == set(LEGACY_SYMBOLS) | set(HIP_SYMBOLS) | set(WAV_SYMBOLS) | set(SYNTH_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_struct_layouts_match_c(lib):
    # natural C alignment of include/aad.h / aad_encoder.h structs on x86-64
    assert C.sizeof(AADHeaderInfo) == 32 and AADHeaderInfo.num_samples.offset == 12
    assert AADHeaderInfo.block_size.offset == 22 and AADHeaderInfo.ch_process_method.offset == 28
    assert C.sizeof(AADEncodeParameter) == 20 and AADEncodeParameter.num_encode_trials.offset == 16


def test_block_size_known_answers(lib):
    codec = aad_amd.LegacyCodec(lib)
    for (mbs, ch, bits), want in {(1024, 1, 4): (1024, 2016), (1024, 2, 4): (1024, 992), (1024, 1, 3): (1023, 2684),
                                  (1024, 2, 3): (1020, 1316), (1024, 1, 2): (1024, 4028), (1024, 2, 2): (1024, 1980),
                                  (128, 1, 3): (126, 292), (256, 2, 4): (256, 224)}.items():
        assert codec.block_size(mbs, ch, bits) == (R.OK,) + want
    assert codec.block_size(1024, 0, 4)[0] == R.INVALID_FORMAT
    assert codec.block_size(1024, 3, 4)[0] == R.INVALID_FORMAT   # legacy API keeps the reference's 2-channel cap
    assert codec.block_size(1024, 1, 0)[0] == R.INVALID_FORMAT
    assert codec.block_size(1024, 1, 5)[0] == R.INVALID_FORMAT
    assert codec.block_size(17, 1, 4)[0] == R.INVALID_FORMAT
    assert lib.AADEncoder_CalculateBlockSize(1024, 1, 4, None, None) == R.INVALID_ARGUMENT
    bs = C.c_uint16()
    assert lib.AADEncoder_CalculateBlockSize(1024, 1, 4, C.byref(bs), None) == R.OK and bs.value == 1024


def valid_header():
    return AADHeaderInfo(format_version=4, codec_version=18, num_channels=1, num_samples=1024, sampling_rate=44100,
                         bits_per_sample=4, block_size=256, num_samples_per_block=32, ch_process_method=0)


def test_header_bytes_and_roundtrip(lib):
    h = valid_header()
    h.num_channels, h.num_samples, h.sampling_rate, h.ch_process_method = 2, 0x01020304, 48000, 1
    h.block_size, h.num_samples_per_block = 0x0A0B, 0x0C0D0E0F
    buf = (C.c_uint8 * 31)()
    assert lib.AADEncoder_EncodeHeader(C.byref(h), buf, 31) == R.OK
    b = bytes(buf)
    assert b[:4] == b"AAD\x00" and b[4:8] == (4).to_bytes(4, "big") and b[8:12] == (18).to_bytes(4, "big")
    assert b[12:14] == b"\x00\x02" and b[14:18] == b"\x01\x02\x03\x04" and b[18:22] == (48000).to_bytes(4, "big")
    assert b[22:24] == b"\x00\x04" and b[24:26] == b"\x0a\x0b" and b[26:30] == b"\x0c\x0d\x0e\x0f" and b[30] == 1
    # writer ignores the struct's version fields (src/aad_encoder.c:195-200)
    h.format_version, h.codec_version = 99, 77
    assert lib.AADEncoder_EncodeHeader(C.byref(h), buf, 31) == R.OK and bytes(buf) == b
    out = AADHeaderInfo()
    assert lib.AADDecoder_DecodeHeader(buf, 31, C.byref(out)) == R.OK
    for f, _ in AADHeaderInfo._fields_:
        want = {"format_version": 4, "codec_version": 18}.get(f, getattr(h, f))
        assert getattr(out, f) == want, f


def test_encode_header_rejections(lib):
    buf = (C.c_uint8 * 31)()
    h = valid_header()
    assert lib.AADEncoder_EncodeHeader(None, buf, 31) == R.INVALID_ARGUMENT
    assert lib.AADEncoder_EncodeHeader(C.byref(h), None, 31) == R.INVALID_ARGUMENT
    assert lib.AADEncoder_EncodeHeader(C.byref(h), buf, 30) == R.INSUFFICIENT_DATA
    for field, bad in (("num_channels", 0), ("num_channels", 3), ("num_samples", 0), ("sampling_rate", 0),
                       ("bits_per_sample", 1), ("bits_per_sample", 5), ("block_size", 18), ("block_size", 0),
                       ("num_samples_per_block", 0), ("ch_process_method", 2), ("ch_process_method", 1)):
        h = valid_header()
        setattr(h, field, bad)
        assert lib.AADEncoder_EncodeHeader(C.byref(h), buf, 31) == R.INVALID_FORMAT, (field, bad)


def test_decode_header_and_set_header_rejections(lib):
    buf = (C.c_uint8 * 31)()
    assert lib.AADEncoder_EncodeHeader(C.byref(valid_header()), buf, 31) == R.OK
    out = AADHeaderInfo()
    assert lib.AADDecoder_DecodeHeader(None, 31, C.byref(out)) == R.INVALID_ARGUMENT
    assert lib.AADDecoder_DecodeHeader(buf, 31, None) == R.INVALID_ARGUMENT
    assert lib.AADDecoder_DecodeHeader(buf, 30, C.byref(out)) == R.INSUFFICIENT_DATA
    bad = (C.c_uint8 * 31)(*bytes(buf))
    bad[0] = ord("a")
    assert lib.AADDecoder_DecodeHeader(bad, 31, C.byref(out)) == R.INVALID_FORMAT
    dec = lib.AADDecoder_Create(None, 0)
    try:
        assert lib.AADDecoder_SetHeader(None, C.byref(valid_header())) == R.INVALID_ARGUMENT
        assert lib.AADDecoder_SetHeader(dec, None) == R.INVALID_ARGUMENT
        assert lib.AADDecoder_SetHeader(dec, C.byref(valid_header())) == R.OK
        for field, badv in (("format_version", 3), ("codec_version", 17), ("num_channels", 0), ("num_channels", 3),
                            ("num_samples", 0), ("sampling_rate", 0), ("bits_per_sample", 1), ("bits_per_sample", 5),
                            ("block_size", 18), ("num_samples_per_block", 0), ("ch_process_method", 2),
                            ("ch_process_method", 1)):
            h = valid_header()
            setattr(h, field, badv)
            assert lib.AADDecoder_SetHeader(dec, C.byref(h)) == R.INVALID_FORMAT, (field, badv)
    finally:
        lib.AADDecoder_Destroy(dec)


def test_header_field_offsets_by_corruption(lib):
    """reference test/test_aad_decoder.c:84-186 - offsets 4,8,12,14,18,22,24,26,30"""
    buf = (C.c_uint8 * 31)()
    h = valid_header()
    h.num_channels, h.block_size = 2, 300
    assert lib.AADEncoder_EncodeHeader(C.byref(h), buf, 31) == R.OK
    for off, field in ((4, "format_version"), (8, "codec_version"), (12, "num_channels"), (14, "num_samples"),
                       (18, "sampling_rate"), (22, "bits_per_sample"), (24, "block_size"),
                       (26, "num_samples_per_block"), (30, "ch_process_method")):
        mod = (C.c_uint8 * 31)(*bytes(buf))
        mod[off] ^= 0x40
        out = AADHeaderInfo()
        assert lib.AADDecoder_DecodeHeader(mod, 31, C.byref(out)) == R.OK
        changed = [f for f, _ in AADHeaderInfo._fields_
                   if getattr(out, f) != {"format_version": 4, "codec_version": 18}.get(f, getattr(h, f))]
        assert changed == [field], (off, changed)


def test_encoder_lifecycle(lib):
    assert lib.AADEncoder_CalculateWorkSize(0) == -1
    size = lib.AADEncoder_CalculateWorkSize(1024)
    assert size > 0
    work = (C.c_uint8 * (size + 32))()
    base = C.addressof(work) + 3  # any alignment is accepted (struct is placed at the next 16-byte boundary)
    enc = lib.AADEncoder_Create(1024, base, size)
    assert enc and enc % 16 == 0 and base <= enc < base + 16
    lib.AADEncoder_Destroy(enc)
    own = lib.AADEncoder_Create(1024, None, 0)
    assert own
    lib.AADEncoder_Destroy(own)
    lib.AADEncoder_Destroy(None)
    assert not lib.AADEncoder_Create(0, None, 0)
    assert not lib.AADEncoder_Create(0, base, size)
    assert not lib.AADEncoder_Create(1024, None, size)
    assert not lib.AADEncoder_Create(1024, base, 0)
    assert not lib.AADEncoder_Create(1024, base, size - 1)


def test_decoder_lifecycle(lib):
    size = lib.AADDecoder_CalculateWorkSize()
    work = (C.c_uint8 * (size + 32))()
    base = C.addressof(work) + 5
    dec = lib.AADDecoder_Create(base, size)
    assert dec and dec % 16 == 0
    lib.AADDecoder_Destroy(dec)
    own = lib.AADDecoder_Create(None, 0)
    assert own
    lib.AADDecoder_Destroy(own)
    assert not lib.AADDecoder_Create(None, size)
    assert not lib.AADDecoder_Create(base, 0)
    assert not lib.AADDecoder_Create(base, size - 1)


def test_set_encode_parameter_matrix(lib):
    enc = lib.AADEncoder_Create(256, None, 0)
    try:
        p = make_parameter(1, 4, 256, 8000, False, 1)
        assert lib.AADEncoder_SetEncodeParameter(None, C.byref(p)) == R.INVALID_ARGUMENT
        assert lib.AADEncoder_SetEncodeParameter(enc, None) == R.INVALID_ARGUMENT
        assert lib.AADEncoder_SetEncodeParameter(enc, C.byref(p)) == R.OK
        for field, bad in (("bits_per_sample", 0), ("bits_per_sample", 5), ("max_block_size", 0),
                           ("max_block_size", 17), ("ch_process_method", 2), ("num_channels", 0), ("num_channels", 3)):
            q = make_parameter(1, 4, 256, 8000, False, 1)
            setattr(q, field, bad)
            assert lib.AADEncoder_SetEncodeParameter(enc, C.byref(q)) == R.INVALID_FORMAT, (field, bad)
        # bits_per_sample == 1 passes here and fails at EncodeHeader time (SURVEY.md section 7 traps)
        q = make_parameter(1, 1, 256, 8000, False, 0)
        assert lib.AADEncoder_SetEncodeParameter(enc, C.byref(q)) == R.OK
    finally:
        lib.AADEncoder_Destroy(enc)


def test_encode_decode_argument_errors_without_gpu(lib):
    """every check that precedes device work behaves like the reference even with no GPU present"""
    x = np.zeros((1, 64), dtype=np.int32)
    rows = (C.POINTER(C.c_int32) * 1)(x[0].ctypes.data_as(C.POINTER(C.c_int32)))
    out = (C.c_uint8 * 4096)()
    size = C.c_uint32()
    enc = lib.AADEncoder_Create(256, None, 0)
    try:
        assert lib.AADEncoder_EncodeWhole(enc, rows, 64, out, 4096, C.byref(size)) == R.PARAMETER_NOT_SET
        p = make_parameter(1, 4, 256, 8000, False, 0)
        assert lib.AADEncoder_SetEncodeParameter(enc, C.byref(p)) == R.OK
        assert lib.AADEncoder_EncodeWhole(None, rows, 64, out, 4096, C.byref(size)) == R.INVALID_ARGUMENT
        assert lib.AADEncoder_EncodeWhole(enc, None, 64, out, 4096, C.byref(size)) == R.INVALID_ARGUMENT
        assert lib.AADEncoder_EncodeWhole(enc, rows, 64, None, 4096, C.byref(size)) == R.INVALID_ARGUMENT
        assert lib.AADEncoder_EncodeWhole(enc, rows, 64, out, 4096, None) == R.INVALID_ARGUMENT
        assert lib.AADEncoder_EncodeWhole(enc, rows, 64, out, 30, C.byref(size)) == R.INSUFFICIENT_DATA
        assert lib.AADEncoder_EncodeWhole(enc, rows, 0, out, 4096, C.byref(size)) == R.INVALID_FORMAT
        assert lib.AADEncoder_EncodeWhole(enc, rows, 64, out, 40, C.byref(size)) == R.INSUFFICIENT_BUFFER
        q = make_parameter(1, 1, 256, 8000, False, 0)
        assert lib.AADEncoder_SetEncodeParameter(enc, C.byref(q)) == R.OK
        assert lib.AADEncoder_EncodeWhole(enc, rows, 64, out, 4096, C.byref(size)) == R.INVALID_FORMAT
    finally:
        lib.AADEncoder_Destroy(enc)

    dec = lib.AADDecoder_Create(None, 0)
    got = C.c_uint32()
    try:
        assert lib.AADDecoder_DecodeBlock(dec, out, 256, rows, 1, 64, C.byref(got)) == R.PARAMETER_NOT_SET
        assert lib.AADDecoder_SetHeader(dec, C.byref(valid_header())) == R.OK
        assert lib.AADDecoder_DecodeBlock(None, out, 256, rows, 1, 64, C.byref(got)) == R.INVALID_ARGUMENT
        assert lib.AADDecoder_DecodeBlock(dec, None, 256, rows, 1, 64, C.byref(got)) == R.INVALID_ARGUMENT
        assert lib.AADDecoder_DecodeBlock(dec, out, 256, None, 1, 64, C.byref(got)) == R.INVALID_ARGUMENT
        assert lib.AADDecoder_DecodeBlock(dec, out, 256, rows, 1, 64, None) == R.INVALID_ARGUMENT
        assert lib.AADDecoder_DecodeBlock(dec, out, 17, rows, 1, 64, C.byref(got)) == R.INSUFFICIENT_DATA
        assert lib.AADDecoder_DecodeBlock(dec, out, 256, rows, 0, 64, C.byref(got)) == R.INSUFFICIENT_BUFFER
        hb = (C.c_uint8 * 31)()
        assert lib.AADEncoder_EncodeHeader(C.byref(valid_header()), hb, 31) == R.OK
        assert lib.AADDecoder_DecodeWhole(None, hb, 31, rows, 1, 1024) == R.INVALID_ARGUMENT
        assert lib.AADDecoder_DecodeWhole(dec, None, 31, rows, 1, 1024) == R.INVALID_ARGUMENT
        assert lib.AADDecoder_DecodeWhole(dec, hb, 31, None, 1, 1024) == R.INVALID_ARGUMENT
        assert lib.AADDecoder_DecodeWhole(dec, hb, 30, rows, 1, 1024) == R.INSUFFICIENT_DATA
        assert lib.AADDecoder_DecodeWhole(dec, hb, 31, rows, 0, 1024) == R.INSUFFICIENT_BUFFER
        assert lib.AADDecoder_DecodeWhole(dec, hb, 31, rows, 1, 1023) == R.INSUFFICIENT_BUFFER
    finally:
        lib.AADDecoder_Destroy(dec)


def test_hip_api_without_device_fails_loudly(lib):
    """No CPU fallback: on a box without a GPU the batched API refuses to create a context."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert lib.AADHip_GetDeviceCount() == 0
    ctx = C.c_void_p()
    assert lib.AADHip_ContextCreate(0, None, C.byref(ctx)) == R.NG and not ctx
    with pytest.raises(RuntimeError):
        from aad_amd.engine import Engine
        Engine()


def test_decode_plan_refuses_unbounded_blocks(lib):
    """The decoder's header checks (reference src/aad_decoder.c:173-225) accept any samples-per-block;
    the device's per-block loop counters are 32-bit and step by up to a pack unit (8 for 3-bit), so
    plan creation refuses a stream whose single block would hold 2^31 samples or more
    (AADFormat_DecodeWorkBounded, the host rule behind AADHip_DecodePlanCreate / DecodeBatch)."""
    lib.AADFormat_DecodeWorkBounded.argtypes = [C.POINTER(AADHeaderInfo), C.c_uint32]
    lib.AADFormat_DecodeWorkBounded.restype = C.c_int
    for bits in (4, 3, 2):
        h = AADHeaderInfo(format_version=4, codec_version=18, num_channels=2, num_samples=1, sampling_rate=48000,
                          bits_per_sample=bits, block_size=1024, num_samples_per_block=0xFFFFFFF0, ch_process_method=0)
        assert lib.AADFormat_DecodeWorkBounded(C.byref(h), 3000) == 1          # the block is cut at num_samples
        assert lib.AADFormat_DecodeWorkBounded(C.byref(h), 0x7FFFFFFF) == 1
        assert lib.AADFormat_DecodeWorkBounded(C.byref(h), 0x80000000) == 0
        assert lib.AADFormat_DecodeWorkBounded(C.byref(h), 0xFFFFFFF8) == 0     # the wrap case of the 3-bit loop
        h.num_samples_per_block = 992
        assert lib.AADFormat_DecodeWorkBounded(C.byref(h), 0xFFFFFFFF) == 1     # many ordinary blocks are fine


def test_context_option_needs_a_context(lib):
    assert lib.AADHip_ContextSetOption(None, 0, 1) == R.INVALID_ARGUMENT


def test_native_corpus_generator_equals_numpy_specification():
    """aad_amd/csrc/aad_synth.c against aad_amd/synth.py's numpy form, every kind, odd shapes."""
    from aad_amd.synth import _native_generator, synth_pcm
    assert _native_generator(), "libaad_hip.so lacks AADSynth_Generate"
    for kind in ("music", "noise", "nyquist"):
        for streams, n, ch, seed, first in ((3, 500, 2, 1234, 0), (2, 333, 8, 77, 5), (1, 1, 1, 0, 0), (5, 64, 3, 99, 1000)):
            a = synth_pcm(streams, n, ch, seed=seed, kind=kind, first_stream=first, native=False)
            b = synth_pcm(streams, n, ch, seed=seed, kind=kind, first_stream=first, native=True)
            assert np.array_equal(a, b), (kind, streams, n, ch)
    assert synth_pcm(0, 10, 2).shape == (0, 10, 2)


def test_encoded_size_matches_formula(lib):
    p = make_parameter(2, 4, 1024)
    assert lib.AADHip_CalculateEncodedSize(C.byref(p), 992) == 31 + 1024
    assert lib.AADHip_CalculateEncodedSize(C.byref(p), 992 * 1000) == 1024031
    assert lib.AADHip_CalculateEncodedSize(C.byref(p), 993) == 31 + 1024 + 36
    assert lib.AADHip_CalculateEncodedSize(C.byref(p), 997) == 31 + 1024 + 36 + 2
    assert lib.AADHip_CalculateEncodedSize(C.byref(p), 0) == 0
    m = make_parameter(1, 4, 1024)
    assert lib.AADHip_CalculateEncodedSize(C.byref(m), 24000) == 12223          # sin300Hz_mono.aad
    p8 = make_parameter(8, 3, 1024)
    assert lib.AADHip_CalculateEncodedSize(C.byref(p8), 292) == 31 + 1008


def test_wav_helpers_match_reference_fixtures(lib):
    """N1: the 16-bit PCM WAV payload is the device layout; the 44-byte header written for a decode
    equals the one in the reference's decoded fixtures (src/wav.c:545-627)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import GOLDEN, read_wav16
    fix = os.path.join(GOLDEN, "ref_fixtures")
    for name in ("sin300Hz.wav", "sin300Hz_mono.wav", "unit_impulse.wav", "unit_impulse_mono.wav", "sin300Hz_decoded.wav"):
        raw = np.frombuffer(open(os.path.join(fix, name), "rb").read(), dtype=np.uint8)
        info = AADWavInfo()
        assert lib.AADWav_ParseHeader(raw.ctypes.data, len(raw), C.byref(info)) == R.OK
        pcm, rate = read_wav16(os.path.join(fix, name))
        assert (info.format_tag, info.num_channels, info.sampling_rate, info.bits_per_sample) == (1, pcm.shape[1], rate, 16)
        assert info.num_samples == pcm.shape[0] and info.data_size == pcm.size * 2
        payload = raw[info.data_offset: info.data_offset + info.data_size].view("<i2").reshape(-1, pcm.shape[1])
        assert np.array_equal(payload, pcm)
    for name in ("sin300Hz_decoded.wav", "sin300Hz_mono_decoded.wav"):
        raw = open(os.path.join(fix, name), "rb").read()
        pcm, rate = read_wav16(os.path.join(fix, name))
        head = (C.c_uint8 * 44)()
        assert lib.AADWav_WriteHeader(head, 44, pcm.shape[1], rate, pcm.shape[0]) == R.OK
        assert bytes(head) == raw[:44]
    info = AADWavInfo()
    bad = np.frombuffer(b"RIFX" + bytes(60), dtype=np.uint8)
    assert lib.AADWav_ParseHeader(bad.ctypes.data, len(bad), C.byref(info)) == R.INVALID_FORMAT
    assert lib.AADWav_ParseHeader(bad.ctypes.data, 8, C.byref(info)) == R.INSUFFICIENT_DATA
    assert lib.AADWav_ParseHeader(None, 64, C.byref(info)) == R.INVALID_ARGUMENT
    assert lib.AADWav_WriteHeader((C.c_uint8 * 44)(), 43, 2, 48000, 10) == R.INSUFFICIENT_BUFFER


def test_signal_next_run_support_follows_the_runtime_setting(monkeypatch):
    """AADHip_SignalNextRunSupported (include/aad_hip.h): 0 exactly when ROC_SYSTEM_SCOPE_SIGNAL=0 - the setting under which a
    wait on a dispatch-carried event from another stream never returns (round-3 record); no device needed to ask.
    The refusal itself (AADHip_ContextSignalNextRun -> NG + LastError) needs a context: tests/test_gpu_signal.py."""
    lib = aad_amd.load_library()
    monkeypatch.delenv("ROC_SYSTEM_SCOPE_SIGNAL", raising=False)
    assert lib.AADHip_SignalNextRunSupported() == 1
    for value, want in (("0", 0), ("1", 1), ("", 1), ("00", 1)):
        monkeypatch.setenv("ROC_SYSTEM_SCOPE_SIGNAL", value)
        assert lib.AADHip_SignalNextRunSupported() == want, value
    # without a context there is nothing to refuse on: the argument check comes first
    assert lib.AADHip_ContextSignalNextRun(None, None, None) == 1  # AAD_APIRESULT_INVALID_ARGUMENT
