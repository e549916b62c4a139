/*
 * aad_wav.h - RIFF/WAVE helpers for the 16-bit PCM fast path.
 *
 * A 16-bit PCM WAV payload IS the engine's device PCM layout (little-endian int16, channel-
 * interleaved frames), so "reading a WAV" is finding the payload and "writing a WAV" is a
 * 44-byte header in front of the decoded frames - no per-sample host loop, unlike the
 * reference's bit-serial reader/writer (src/wav.c:455-503, 545-627).  The decoded .wav bytes
 * are identical to what the reference CLI writes (src/main.c:122-128 + src/wav.c:545-665).
 * Other sample formats are outside the fast path: AADWav_ParseHeader reports them and the
 * caller converts.
 */
#ifndef AAD_WAV_H_INCLUDED
#define AAD_WAV_H_INCLUDED

#include <stdint.h>
#include "aad_api.h"

#define AAD_WAV_HEADER_SIZE 44

struct AADWavInfo {
  uint16_t format_tag;      /* 1 = PCM */
  uint16_t num_channels;
  uint32_t sampling_rate;
  uint16_t bits_per_sample;
  uint32_t num_samples;     /* frames */
  uint64_t data_offset;     /* byte offset of the first frame in the file image */
  uint64_t data_size;       /* payload bytes */
};

#ifdef __cplusplus
extern "C" {
#endif

/* Walk the RIFF chunks (unknown chunks are skipped like src/wav.c:176-193).  INVALID_FORMAT when
 * the image is not RIFF/WAVE or has no fmt/data chunk, INSUFFICIENT_DATA when truncated. */
AADApiResult AADWav_ParseHeader(const uint8_t *data, uint64_t data_size, struct AADWavInfo *info);

/* The canonical 44-byte header the reference writer emits for 16-bit PCM (src/wav.c:545-627). */
AADApiResult AADWav_WriteHeader(uint8_t *data, uint32_t data_size, uint16_t num_channels,
                                uint32_t sampling_rate, uint32_t num_samples);

/* 8 / 16 / 24 / 32-bit little-endian PCM -> the int16 the codec sees.  The reference reader widens
 * every sample to 32 bits (src/wav.c:392-417: 8-bit (v - 128) << 24, 16-bit << 16, 24-bit << 8) and
 * the CLI keeps the top 16 of those (src/main.c:175-179), i.e. 8-bit: (v - 128) << 8; 24- and 32-bit:
 * the two most significant bytes.  `count` = frames * channels.  INVALID_FORMAT for other widths. */
AADApiResult AADWav_ConvertToPcm16(const uint8_t *payload, uint16_t bits_per_sample, uint64_t count, int16_t *pcm);

#ifdef __cplusplus
}
#endif

#endif /* AAD_WAV_H_INCLUDED */
