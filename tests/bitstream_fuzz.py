"""Crafted .aad images no encoder writes: random / adversarial block headers and code bodies.

The reference decoder defines a result for ANY bytes behind a valid file header: each block
reloads its whole predictor state from the 18-byte channel header - step index = u16 >> 4, weights
= (int16) << (u16 & 15), four int16 history samples (reference src/aad_decoder.c:364-391, the
shift at :376) - and then decodes whatever codes follow (:396-451).  An encoder only ever writes a
small corner of that space (shift 0-2 or so, weights it masked itself, an index its own walk
reached); this module builds the rest:

  * step index anywhere in 0..4087 (4081..4087 still index the last table entry: (idx + 8) >> 4 =
    255; 4088..4095 would read past the reference's 256-entry table, src/aad_tables.h:28 - undefined
    there, so not part of the parity contract) - sitting on / next to both clamps;
  * shift 0..15 with full-range int16 weights (|w| up to 2^30: the prediction's int32 sum wraps);
  * history anywhere in int16, rails included;
  * code bodies: random bytes, all-0x77 / 0xFF / 0x88 / 0x00 runs (maximum magnitude of either
    sign - the step index runs into its upper clamp and the output into both rails; minimum
    magnitude - into the lower clamp), alternating extremes, and random mixtures of those runs;
  * 1-8 channels, 2/3/4 bits, M/S, 1-3 blocks with a ragged last one, odd block sizes.

Everything is derived from SHAKE-256 of the case name - no numpy / Python RNG whose stream could
change between versions - so `tests/golden/bitstream_fuzz.json` (decode hashes from the compiled
reference, `tests/golden/make_bitstream_golden.py`) pins the same bytes everywhere.

TEST INFRASTRUCTURE (tests/ only).
"""
import hashlib
import struct

import numpy as np

import oracle_binding as ob

HEADER_BYTES = 31
INDEX_FIELD_MAX = 4087  # (idx + 8) >> 4 <= 255: the last header value the reference's table lookup is defined for


class Bytes:
    """deterministic byte / integer source: SHAKE-256(name) read front to back"""

    def __init__(self, name, size=1 << 16):
        self.buf = hashlib.shake_256(name.encode()).digest(size)
        self.pos = 0

    def take(self, n):
        if self.pos + n > len(self.buf):
            raise ValueError("fuzz source exhausted")
        b = self.buf[self.pos:self.pos + n]
        self.pos += n
        return b

    def u32(self):
        return struct.unpack("<I", self.take(4))[0]

    def below(self, n):
        """integer in [0, n) - modulo bias is irrelevant here"""
        return self.u32() % n

    def choice(self, seq):
        return seq[self.below(len(seq))]


def file_header(channels, num_samples, rate, bits, block_size, spb, ms):
    """the 31 big-endian bytes of reference src/aad_encoder.c:190-214"""
    return (b"AAD\0" + struct.pack(">IIHIIHHIB", 4, 18, channels, num_samples, rate, bits, block_size, spb,
                                   1 if ms else 0))


_PATTERNS = {
    4: [0x77, 0xFF, 0x88, 0x00, 0x7F, 0xF7, 0x08, 0x80, 0x70, 0x0F],
    # 3-bit units are 3 bytes; byte runs still give every code pattern that matters: 0xFF -> all 7 (max
    # magnitude, negative), 0x00 -> all 0, 0x6D/0xB6/0xDB (011 011 011 ...) -> max magnitude positive
    3: [0xFF, 0x00, 0x6D, 0xB6, 0xDB, 0x92, 0x49, 0x24, 0xE3, 0x1C],
    2: [0x55, 0xFF, 0xAA, 0x00, 0x5F, 0xF5, 0x0A, 0xA0, 0x77, 0xDD],
}


def _body(src, kind, size, bits):
    if size == 0:
        return b""
    if kind == "random":
        return hashlib.shake_256(src.take(16)).digest(size)
    pats = _PATTERNS[bits]
    if kind == "run":
        return bytes([src.choice(pats)]) * size
    if kind == "alternate":
        a, b = src.choice(pats), src.choice(pats)
        period = src.choice([1, 2, 3, 5, 8, 16, 33])
        return np.where((np.arange(size) // period) % 2 == 0, a, b).astype(np.uint8).tobytes()
    if kind == "mixture":
        out = bytearray()
        while len(out) < size:
            n = 1 + src.below(min(200, size))
            if src.below(3) == 0:
                out += hashlib.shake_256(src.take(8)).digest(n)
            else:
                out += bytes([src.choice(pats)]) * n
        return bytes(out[:size])
    raise ValueError(kind)


def _i16(src, kind):
    if kind == "rails":
        return src.choice([-32768, 32767, -32768, 32767, 0, -1, 1])
    if kind == "small":
        return src.below(513) - 256
    return src.below(65536) - 32768


def _channel_header(src, hkind):
    """18 bytes: u16 (idx << 4 | shift), then 4 x (i16 weight, i16 history)"""
    if hkind == "encoderlike":
        idx, shift = src.below(4081), src.below(3)
        wk, hk = "any", "any"
    elif hkind == "clamp_low":
        idx, shift = src.choice([0, 0, 1, 7, 8, 9, 15, 16, 17, 40]), src.below(16)
        wk, hk = src.choice(["any", "rails", "small"]), src.choice(["any", "rails"])
    elif hkind == "clamp_high":
        idx, shift = src.choice([4080, 4080, 4079, 4072, 4064, 4063, 4040, 4081, 4087, 4085]), src.below(16)
        wk, hk = src.choice(["any", "rails", "small"]), src.choice(["any", "rails"])
    elif hkind == "bigshift":
        idx, shift = src.below(INDEX_FIELD_MAX + 1), 3 + src.below(13)
        wk, hk = src.choice(["any", "rails"]), src.choice(["any", "rails"])
    else:  # "any"
        idx, shift = src.below(INDEX_FIELD_MAX + 1), src.below(16)
        wk, hk = "any", "any"
    out = struct.pack(">H", (idx << 4) | shift)
    for _ in range(4):
        out += struct.pack(">hh", _i16(src, wk), _i16(src, hk))
    return out


HEADER_KINDS = ["any", "encoderlike", "clamp_low", "clamp_high", "bigshift"]
BODY_KINDS = ["random", "run", "alternate", "mixture"]


def make_case(name, max_channels=2, channels=None, bits=None, max_block_size=None, ms=None, blocks=None, last=None,
              header_kind=None, body_kind=None):
    """-> dict(name, channels, bits, ms, max_block_size, block_size, spb, num_samples, image bytes).  The keyword
    arguments fix a property instead of drawing it (same-format batches: all but the name held equal); ms / blocks /
    last / the kinds are drawn all the same, so that fixing them does not shift the rest of the case's bytes."""
    src = Bytes("aad-bitstream-fuzz/" + name)
    ch = channels or (src.choice([1, 2, 2]) if max_channels <= 2 else src.choice([1, 2, 3, 4, 5, 6, 7, 8]))
    b = bits or src.choice([4, 3, 2])
    drawn_ms = bool(ch == 2 and src.below(3) == 0)
    ms = drawn_ms if ms is None else bool(ms)
    while True:
        mbs = max_block_size or src.choice([1024, 1024, 1024, 512, 256, 2048, 18 * ch + 1 + src.below(600), 4096])
        rc, block_size, spb = ob.geometry(mbs, ch, b)
        if rc == 0 and spb > 4:
            break
        if max_block_size:
            raise ValueError("no block fits %r" % ((mbs, ch, b),))
    drawn_blocks = 1 + src.below(3)
    blocks = blocks or drawn_blocks
    # a ragged last block: anywhere from 1 sample (header only, not even all four stored samples) to full
    drawn_last = src.choice([spb, spb, 1 + src.below(spb), 1 + src.below(min(spb, 12))])
    last = min(last, spb) if last else drawn_last
    num_samples = (blocks - 1) * spb + last
    unit_samples = {4: 2, 3: 8, 2: 4}[b]
    unit_bytes = {4: 1, 3: 3, 2: 1}[b] * ch
    drawn_hk, drawn_bk = src.choice(HEADER_KINDS), src.choice(BODY_KINDS)
    hk, bk = header_kind or drawn_hk, body_kind or drawn_bk
    image = bytearray(file_header(ch, num_samples, 48000, b, block_size, spb, ms))
    for k in range(blocks):
        n = spb if k + 1 < blocks else last
        for _ in range(ch):
            image += _channel_header(src, hk)
        data_bytes = -(-max(n - 4, 0) // unit_samples) * unit_bytes
        image += _body(src, bk, data_bytes, b)
    return {"name": name, "channels": ch, "bits": b, "ms": ms, "max_block_size": mbs, "block_size": block_size,
            "spb": spb, "num_samples": num_samples, "header_kind": hk, "body_kind": bk, "image": bytes(image)}


def case_names(count, prefix="c"):
    return ["%s%04d" % (prefix, i) for i in range(count)]


def oracle_decode(image):
    """int16 [samples, channels] from the oracle"""
    return ob.decode(image)[0]


def pcm_hash(pcm):
    return hashlib.sha256(np.ascontiguousarray(pcm, dtype="<i2").tobytes()).hexdigest()


def channel_as_mono_image(case, c):
    """Channel c of an N-channel image (no M/S) as a MONO image the reference can decode: the same 18
    header bytes and the same code units per block, un-interleaved (reference src/aad_decoder.c:396-451
    walks units channel by channel), under a mono block size with the same samples per block.  The
    reference stops at two channels (src/aad.h:13); this is how SURVEY.md section 8c pins the wider
    container: without M/S a channel's recurrence does not see its neighbours."""
    ch, b, spb, n = case["channels"], case["bits"], case["spb"], case["num_samples"]
    assert not case["ms"]
    ub = {4: 1, 3: 3, 2: 1}[b]
    us = {4: 2, 3: 8, 2: 4}[b]
    units_full = (spb - 4) // us
    mono_block = 18 + units_full * ub
    rc, bs, spb1 = ob.geometry(mono_block, 1, b)
    assert rc == 0 and bs == mono_block and spb1 == spb, (mono_block, bs, spb1, spb)
    img = case["image"]
    out = bytearray(file_header(1, n, 48000, b, mono_block, spb, False))
    pos, left = HEADER_BYTES, n
    while left > 0:
        k = min(left, spb)
        units = -(-max(k - 4, 0) // us)
        out += img[pos + 18 * c:pos + 18 * c + 18]
        body = pos + 18 * ch
        for u in range(units):
            o = body + (u * ch + c) * ub
            out += img[o:o + ub]
        pos += 18 * ch + units * ub * ch
        left -= k
    assert pos == len(img)
    return bytes(out)


def golden_cases():
    """the records of tests/golden/bitstream_fuzz.json"""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bitstream_fuzz.json")) as f:
        return json.load(f)["cases"]


def case_of_record(rec):
    """rebuild the image a golden record was made from (and check that the generator has not drifted)"""
    if rec["name"].startswith("g"):
        case = make_geometry_case(rec["name"])
    else:
        case = make_case(rec["name"]) if rec["name"].startswith("c") else make_case(rec["name"], channels=rec["channels"])
    if hashlib.sha256(case["image"]).hexdigest() != rec["image_sha256"]:
        raise AssertionError("bitstream_fuzz.make_case(%r) no longer produces the image the golden was made from" % rec["name"])
    return case


def make_geometry_case(name, max_channels=2):
    """A file header whose block_size and samples_per_block do NOT belong together.  The reference checks neither against the other
    (src/aad_decoder.c:173-225: block_size > 18 x channels, samples_per_block > 0, that is all): its block walk advances by
    block_size (:514-534) while a block's decode consumes what samples_per_block asks for (:396-451) - fewer bytes than the block
    holds, or MORE (reading on into the following blocks' bytes), and fewer than four samples per block still emits the header's
    stored samples (:386-391).  Defined as long as every read stays inside the file (the reference does not bound-check its reads in
    release builds): the image carries a tail of bytes behind its last block, long enough for the farthest read; the block walk
    never reaches it (it stops at num_samples), only overflowing code reads do."""
    src = Bytes("aad-geometry-fuzz/" + name)
    ch = src.choice([1, 2, 2]) if max_channels <= 2 else 1 + src.below(max_channels)
    b = src.choice([4, 3, 2])
    ms = bool(ch == 2 and src.below(4) == 0)
    unit_samples = {4: 2, 3: 8, 2: 4}[b]
    unit_bytes = {4: 1, 3: 3, 2: 1}[b] * ch
    block_size = 18 * ch + 1 + src.below(src.choice([20, 200, 1200]))
    fits = 4 + (block_size - 18 * ch) // unit_bytes * unit_samples  # samples a block of this size can hold
    spb = src.choice([1 + src.below(8), max(1, fits - src.below(min(fits, 40))), fits, fits + 1 + src.below(3 * fits + 50), 1 + src.below(3000)])
    blocks = 1 + src.below(4)
    # the last block: as many samples as stay inside its own bytes
    last_cap = min(spb, fits)
    last = 1 + src.below(last_cap)
    num_samples = (blocks - 1) * spb + last
    def needed(n):  # bytes a block's decode of n samples touches, from the block's first byte
        return 18 * ch + -(-max(n - 4, 0) // unit_samples) * unit_bytes
    reach = max([k * block_size + needed(spb) for k in range(blocks - 1)] + [(blocks - 1) * block_size + needed(last)])
    body = hashlib.shake_256(src.take(16)).digest(max(blocks * block_size, reach))
    image = bytearray(file_header(ch, num_samples, 48000, b, block_size, spb, ms)) + bytearray(body)
    hk = src.choice(HEADER_KINDS)
    for k in range(blocks):  # proper channel headers at the head of every block (index field <= 4087)
        for c in range(ch):
            o = HEADER_BYTES + k * block_size + 18 * c
            if o + 18 <= len(image):
                image[o:o + 18] = _channel_header(src, hk)
    return {"name": name, "channels": ch, "bits": b, "ms": ms, "block_size": block_size, "spb": spb, "fits": fits,
            "num_samples": num_samples, "blocks": blocks, "image": bytes(image)}
