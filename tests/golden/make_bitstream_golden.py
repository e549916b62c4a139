#!/usr/bin/env python3
"""Golden decodes of CRAFTED bitstreams (tests/bitstream_fuzz.py) from the COMPILED reference.

The reference decoder defines a result for any bytes behind a valid file header (block header
reload with `<< shift` of any int16 weight, src/aad_decoder.c:364-380; any code sequence,
:396-451).  No encoder writes most of that space, so encoder-made goldens do not pin it.  This
script pushes the crafted images of `bitstream_fuzz.make_case` through
oracle/_ref/libaadref.so's AADDecoder_DecodeWhole and records, per case, the SHA-256 of the image
(so a generator that drifted is noticed) and of the decoded int16 PCM:

  c0000..  1-2 channels (M/S included): the reference decodes the image as it is
  m0000..  3-8 channels: the reference cannot (AAD_MAX_NUM_CHANNELS = 2, src/aad.h:13); every
           channel is decoded as the equivalent mono image (`channel_as_mono_image`, the rule of
           SURVEY.md section 8c) and the columns are put side by side
  g0000..  1-2 channels, file headers whose block_size and samples_per_block do not belong together
           (`make_geometry_case`): the reference checks neither against the other, walks blocks by
           block_size and reads codes by samples_per_block - on into the following blocks' bytes if need be

Output: tests/golden/bitstream_fuzz.json.  Runs only in the build container (needs oracle/_ref);
nothing of the reference's source is copied - only hashes of what it computed.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import aad_amd  # noqa: E402
import bitstream_fuzz as bf  # noqa: E402
import oracle_binding as ob  # noqa: E402
from helpers import sha256  # noqa: E402

STEREO_CASES = 600
WIDE_CASES = 200
GEOMETRY_CASES = 300


def reference_decode(ref, case):
    if case["channels"] <= 2:
        return ref.decode(case["image"])[0]
    cols = [ref.decode(bf.channel_as_mono_image(case, c))[0] for c in range(case["channels"])]
    return np.concatenate(cols, axis=1)


def main():
    ref = aad_amd.LegacyCodec(aad_amd.load_library(ob.REF_SO, hip=False))
    cases = []
    for prefix, count, maxch in (("c", STEREO_CASES, 2), ("m", WIDE_CASES, 8)):
        for i, name in enumerate(bf.case_names(count, prefix)):
            case = bf.make_case(name) if maxch <= 2 else bf.make_case(name, channels=3 + i % 6)
            pcm = reference_decode(ref, case)
            assert pcm.shape == (case["num_samples"], case["channels"])
            rec = {k: case[k] for k in ("name", "channels", "bits", "ms", "max_block_size", "block_size", "spb",
                                        "num_samples", "header_kind", "body_kind")}
            rec["image_sha256"] = sha256(case["image"])
            rec["decoded_sha256"] = bf.pcm_hash(pcm)
            cases.append(rec)
    for name in bf.case_names(GEOMETRY_CASES, "g"):
        case = bf.make_geometry_case(name)
        pcm = ref.decode(case["image"])[0]
        assert pcm.shape == (case["num_samples"], case["channels"])
        rec = {k: case[k] for k in ("name", "channels", "bits", "ms", "block_size", "spb", "num_samples", "fits", "blocks")}
        rec["header_kind"], rec["body_kind"] = "geometry", "random"
        rec["image_sha256"] = sha256(case["image"])
        rec["decoded_sha256"] = bf.pcm_hash(pcm)
        cases.append(rec)
    out = {"generator": "tests/golden/make_bitstream_golden.py", "source": "oracle/_ref/libaadref.so (AADDecoder_DecodeWhole)",
           "cases": cases}
    path = os.path.join(HERE, "bitstream_fuzz.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
        f.write("\n")
    print("wrote %s: %d cases" % (path, len(cases)))


if __name__ == "__main__":
    main()
