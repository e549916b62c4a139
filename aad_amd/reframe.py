"""Re-frame the channels of a multi-channel .aad batch as mono images (numpy, no codec work).

The reference stops at two channels (AAD_MAX_NUM_CHANNELS, src/aad.h:13); the >2-channel container
of BASELINE config 4 is pinned channel by channel: without M/S the
per-channel recurrence does not see the channel count, so channel c of an N-channel stream must
carry exactly the codes and block-header fields of the same samples encoded as a MONO stream
whose samples-per-block equals the N-channel geometry's.  This module only moves bytes: it undoes
the per-unit channel interleave (src/aad_encoder.c:663-718) and rewrites the two header fields
that differ (channel count, block size).  Used by tests and by bench.py's golden check.
"""
import math

import numpy as np


def channels_as_mono_images(images, channels, bits, block_size, mono_block_size):
    """images: uint8 array [streams, image_bytes] of ONE-BLOCK N-channel images (31-byte header +
    one block of `block_size` bytes or less) -> uint8 array [streams, channels, mono_image_bytes]."""
    images = np.asarray(images, dtype=np.uint8)
    streams, size = images.shape
    ub = math.lcm(8, bits) // 8
    body_bytes = size - 31 - 18 * channels
    assert size <= 31 + block_size and body_bytes >= 0 and body_bytes % (ub * channels) == 0
    units = body_bytes // (ub * channels)
    head = np.repeat(images[:, None, :31], channels, axis=1).copy()
    head[:, :, 12:14] = (0, 1)                                  # num_channels = 1
    head[:, :, 24:26] = (mono_block_size >> 8, mono_block_size & 0xFF)
    block = images[:, 31:]
    ch_head = block[:, :18 * channels].reshape(streams, channels, 18)
    body = block[:, 18 * channels:].reshape(streams, units, channels, ub).transpose(0, 2, 1, 3)
    return np.concatenate([head, ch_head, body.reshape(streams, channels, units * ub)], axis=2)
