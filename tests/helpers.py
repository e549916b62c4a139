"""Shared test helpers: minimal RIFF/WAVE PCM16 reader/writer and hashing."""
import hashlib
import os
import struct

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def sha256(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def read_wav16(path):
    """-> (int16 [samples, channels], rate).  Skips unknown chunks like the reference reader
    (src/wav.c:176-193); only 16-bit PCM is needed for the fixtures."""
    b = open(path, "rb").read()
    assert b[:4] == b"RIFF" and b[8:12] == b"WAVE"
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(b):
        cid, size = b[pos:pos + 4], struct.unpack("<I", b[pos + 4:pos + 8])[0]
        body = b[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
        elif cid == b"data":
            data = body
            break
        pos += 8 + size + (size & 1)
    tag, ch, rate, _, _, bits = fmt
    assert tag == 1 and bits == 16
    x = np.frombuffer(data, dtype="<i2")
    return x[: len(x) // ch * ch].reshape(-1, ch).copy(), rate


def wav16_bytes(pcm, rate):
    """Canonical 44-byte-header WAV image, the layout the reference writer emits
    (src/wav.c:545-627) for 16-bit PCM."""
    pcm = np.ascontiguousarray(pcm, dtype="<i2")
    n, ch = pcm.shape
    payload = pcm.tobytes()
    head = b"RIFF" + struct.pack("<I", 36 + len(payload)) + b"WAVE" + b"fmt " + struct.pack(
        "<IHHIIHH", 16, 1, ch, rate, rate * 2 * ch, 2 * ch, 16) + b"data" + struct.pack("<I", len(payload))
    return head + payload


def cli_mode_input(case):
    """Rebuild the WAV input of a tests/golden/cli_modes.json case -> (pcm int16 [n, ch], rate, wav bytes)."""
    import os as _os
    src = case["source"]
    if "fixture" in src:
        path = _os.path.join(GOLDEN, "ref_fixtures", src["fixture"])
        pcm, rate = read_wav16(path)
        data = open(path, "rb").read()
    else:
        from aad_amd.synth import synth_pcm
        pcm, rate = synth_pcm(1, src["samples"], src["channels"], seed=src["seed"], kind=src["kind"])[0], 48000
        data = wav16_bytes(pcm, rate)
    assert sha256(data) == case["input_sha256"]
    return pcm, rate, data


def wav_bytes_depth(pcm, rate, depth, salt=0):
    """A PCM WAV image of `depth` bits per sample (8 / 16 / 24 / 32) whose top 16 bits per sample are
    `pcm` (int16 [samples, channels]) - 8-bit keeps only the top byte - and whose lower bits are
    deterministic filler (an LCG seeded by `salt`), so that a reader which does not discard them by
    the reference's rule (src/main.c:175-179, src/wav.c:392-417) gives different codes."""
    pcm = np.ascontiguousarray(pcm, dtype="<i2")
    n, ch = pcm.shape
    flat = pcm.reshape(-1).astype(np.int64)
    x = (np.arange(flat.size, dtype=np.uint64) * np.uint64(6364136223846793005) + np.uint64(1442695040888963407 + salt))
    low = ((x >> np.uint64(40)) & np.uint64(0xFFFF)).astype(np.int64)
    if depth == 8:
        payload = (((flat >> 8) + 128) & 0xFF).astype(np.uint8).tobytes()
    elif depth == 16:
        payload = pcm.tobytes()
    elif depth == 24:
        v = ((flat << 8) | (low & 0xFF)) & 0xFFFFFF
        b = np.empty((flat.size, 3), dtype=np.uint8)
        b[:, 0], b[:, 1], b[:, 2] = v & 0xFF, (v >> 8) & 0xFF, (v >> 16) & 0xFF
        payload = b.tobytes()
    elif depth == 32:
        payload = (((flat << 16) | low) & 0xFFFFFFFF).astype("<u4").tobytes()
    else:
        raise ValueError(depth)
    bps = depth // 8
    head = b"RIFF" + struct.pack("<I", 36 + len(payload)) + b"WAVE" + b"fmt " + struct.pack(
        "<IHHIIHH", 16, 1, ch, rate, rate * bps * ch, bps * ch, depth) + b"data" + struct.pack("<I", len(payload))
    return head + payload
