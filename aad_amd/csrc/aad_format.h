/*
 * aad_format.h - host-side format arithmetic shared by the legacy C API and the batched engine:
 * block geometry, the 31-byte file header, header validation, encoded sizes.  Plain C, no GPU.
 */
#ifndef AAD_FORMAT_H
#define AAD_FORMAT_H

#include <stddef.h>
#include <stdint.h>
#include "../../include/aad.h"
#include "../../include/aad_encoder.h"

#ifdef __cplusplus
extern "C" {
#endif

#define AAD_NUM_TAPS 4                  /* reference src/aad_internal.h:10 (AAD_FILTER_ORDER) */
#define AAD_BLOCK_HEADER_BYTES_PER_CH 18 /* reference src/aad_internal.h:34 */

/* how codes of one channel are grouped into whole bytes: lcm(8, bits) bits per unit */
struct AADPackUnit {
  uint32_t bytes_per_channel; /* 4-bit: 1, 3-bit: 3, 2-bit: 1 */
  uint32_t samples;           /* 4-bit: 2, 3-bit: 8, 2-bit: 4 */
};
struct AADPackUnit AADFormat_PackUnit(uint32_t bits_per_sample);

/* reference src/aad_encoder.c:85-131 with the channel limit as a parameter */
AADApiResult AADFormat_BlockGeometry(uint32_t max_block_size, uint32_t num_channels, uint32_t bits_per_sample,
                                     uint32_t max_channels, uint16_t *block_size, uint32_t *samples_per_block);

/* field checks of reference src/aad_encoder.c:152-185 / src/aad_decoder.c:189-222 (no version check) */
int AADFormat_HeaderFieldsValid(const struct AADHeaderInfo *header, uint32_t max_channels);
/* + version check, reference src/aad_decoder.c:180-187 */
int AADFormat_HeaderAcceptedByDecoder(const struct AADHeaderInfo *header, uint32_t max_channels);

/* 31-byte big-endian header image; caller has validated the fields */
void AADFormat_PutHeader(const struct AADHeaderInfo *header, uint8_t *data);
/* returns 0 when the signature is wrong */
int AADFormat_GetHeader(const uint8_t *data, struct AADHeaderInfo *header);

/* parameter -> header (reference src/aad_encoder.c:730-776); INVALID_FORMAT on a bad parameter */
AADApiResult AADFormat_ParameterToHeader(const struct AADEncodeParameter *parameter, uint32_t num_samples,
                                         uint32_t max_channels, struct AADHeaderInfo *header);

/* bytes of a block holding n samples per channel (n <= samples_per_block) */
uint32_t AADFormat_BlockBytes(uint32_t n, uint32_t num_channels, uint32_t bits_per_sample);
/* bytes of header + all blocks */
uint64_t AADFormat_EncodedSize(const struct AADHeaderInfo *header);


/* The device's per-block loop counters are 32-bit and one lane decodes a whole block: a header may
 * claim any samples-per-block (nothing in reference src/aad_decoder.c:173-225 ties it to the block
 * size), so a stream whose largest block would hold 2^31 samples per channel or more is refused
 * (it could not be real data: a 65535-byte block holds at most 262 072 coded samples).
 * Returns non-zero when min(num_samples, samples_per_block) is below that bound. */
int AADFormat_DecodeWorkBounded(const struct AADHeaderInfo *header, uint32_t num_samples);

#ifdef __cplusplus
}
#endif
#endif
