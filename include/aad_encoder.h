/*
 * aad_encoder.h - encoder half of the AAD C API, backed by the MI355X HIP engine.
 *
 * Drop-in for reference src/aad_encoder.h:8-50: identical symbols, argument meaning and
 * return codes.  Differences a caller can observe are listed in INTEGRATION.md:
 *   - the handle's work area is smaller (the device owns the block buffers);
 *   - AADEncoder_EncodeWhole returns AAD_APIRESULT_INSUFFICIENT_BUFFER instead of writing
 *     past data_size (the reference only asserts, src/aad_encoder.c:666-667);
 *   - AADEncoder_EncodeWhole returns AAD_APIRESULT_NG if no HIP device is usable.
 */
#ifndef AAD_ENCODER_H_INCLDED
#define AAD_ENCODER_H_INCLDED

#include "aad.h"
#include <stdint.h>

/* reference src/aad_encoder.h:8-15 */
struct AADEncodeParameter {
  uint16_t num_channels;
  uint32_t sampling_rate;
  uint16_t bits_per_sample;
  uint16_t max_block_size;                    /* bytes */
  AADChannelProcessMethod ch_process_method;
  uint8_t  num_encode_trials;                 /* 0 = pure recurrence; the reference CLI default is 2 */
};

struct AADEncoder; /* opaque */

#ifdef __cplusplus
extern "C" {
#endif

/* block geometry for a parameter set - reference src/aad_encoder.h:25-27, src/aad_encoder.c:85-131 */
AADApiResult AADEncoder_CalculateBlockSize(
    uint16_t max_block_size, uint16_t num_channels, uint32_t bits_per_sample,
    uint16_t *block_size, uint32_t *num_samples_per_block);

/* serialise the 31-byte file header - reference src/aad_encoder.h:30-31, src/aad_encoder.c:134-221 */
AADApiResult AADEncoder_EncodeHeader(
    const struct AADHeaderInfo *header_info, uint8_t *data, uint32_t data_size);

/* handle lifecycle - reference src/aad_encoder.h:34-40, src/aad_encoder.c:224-327.
 * Create(max_block_size, NULL, 0) allocates; Create(.., work, work_size) places the handle in
 * caller memory of at least CalculateWorkSize bytes.  No GPU work happens before the first encode. */
int32_t AADEncoder_CalculateWorkSize(uint16_t max_block_size);
struct AADEncoder *AADEncoder_Create(uint16_t max_block_size, void *work, int32_t work_size);
void AADEncoder_Destroy(struct AADEncoder *encoder);

/* reference src/aad_encoder.h:43-44, src/aad_encoder.c:779-811 */
AADApiResult AADEncoder_SetEncodeParameter(
    struct AADEncoder *encoder, const struct AADEncodeParameter *parameter);

/* header + all blocks of one stream; input is planar int32 holding int16-range samples -
 * reference src/aad_encoder.h:47-50, src/aad_encoder.c:814-891 */
AADApiResult AADEncoder_EncodeWhole(
    struct AADEncoder *encoder,
    const int32_t *const *input, uint32_t num_samples,
    uint8_t *data, uint32_t data_size, uint32_t *output_size);

#ifdef __cplusplus
}
#endif

#endif /* AAD_ENCODER_H_INCLDED */
