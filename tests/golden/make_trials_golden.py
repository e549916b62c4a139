#!/usr/bin/env python3
"""Golden images for encode-trial counts beyond the reference CLI's default (num_encode_trials is a uint8_t,
reference src/aad_encoder.h:14; the search loop takes any count, src/aad_encoder.c:531-557: each trial = [previous block,]
current block from where the last trial ended, the candidate is the state in front of the pass over the current block, the
winner is the first strict minimum).  The rest of the goldens stop at 2 (one at 3).

Cases: t in {3, 4, 5, 7, 16, 255}, 2/3/4 bits, mono / stereo, M/S, several block sizes and ragged lengths, music / noise /
Nyquist inputs from the integer synthesiser (aad_amd/synth.py) - SHA-256 of the image the COMPILED reference
(oracle/_ref/libaadref.so) writes, and of its decode.  Output: tests/golden/trials_high.json.  Build container only."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import aad_amd  # noqa: E402
import oracle_binding as ob  # noqa: E402
from aad_amd.synth import synth_pcm  # noqa: E402
from helpers import sha256  # noqa: E402


def cases():
    out, k = [], 0
    for trials in (3, 4, 5, 7, 16, 255):
        for bits in (4, 3, 2):
            for ch, ms in ((1, False), (2, False), (2, True)):
                for rep in range(2):
                    k += 1
                    mbs = (1024, 256, 600, 128)[(k + rep) % 4]
                    rc, _, spb = ob.geometry(mbs, ch, bits)
                    assert rc == 0
                    if trials == 255:
                        n = (spb // 3, 2 * spb + 5)[rep] if mbs <= 256 else (spb // 2, spb + 9)[rep]
                    else:
                        n = (3 * spb + 17, spb - 3, 5 * spb, 2 * spb + 1)[(k + rep) % 4]
                    out.append(dict(trials=trials, bits=bits, channels=ch, ms=ms, max_block_size=mbs, num_samples=int(n),
                                    seed=7000 + k, kind=("music", "noise", "nyquist")[k % 3]))
    return out


def main():
    ref = aad_amd.LegacyCodec(aad_amd.load_library(ob.REF_SO, hip=False))
    recs = []
    for c in cases():
        pcm = synth_pcm(1, c["num_samples"], c["channels"], seed=c["seed"], kind=c["kind"])[0]
        image = ref.encode(pcm, c["bits"], c["max_block_size"], 48000, c["ms"], c["trials"])
        decoded, _ = ref.decode(image)
        recs.append(dict(c, aad_sha256=sha256(image), decoded_sha256=sha256(decoded.astype("<i2").tobytes()), aad_bytes=len(image)))
    path = os.path.join(HERE, "trials_high.json")
    with open(path, "w") as f:
        json.dump({"generator": "tests/golden/make_trials_golden.py", "source": "oracle/_ref/libaadref.so", "cases": recs}, f, indent=0, sort_keys=True)
        f.write("\n")
    print("wrote %s: %d cases" % (path, len(recs)))


if __name__ == "__main__":
    main()
