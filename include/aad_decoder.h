/*
 * aad_decoder.h - decoder half of the AAD C API, backed by the MI355X HIP engine.
 *
 * Drop-in for reference src/aad_decoder.h:15-42: identical symbols, argument meaning and
 * return codes.  Bytes past data_size decode as zero (the reference reads out of bounds on a
 * truncated block, src/aad_decoder.c:396-451); see INTEGRATION.md.
 */
#ifndef AAD_DECODER_H_INCLDED
#define AAD_DECODER_H_INCLDED

#include "aad.h"
#include <stdint.h>

struct AADDecoder; /* opaque */

#ifdef __cplusplus
extern "C" {
#endif

/* parse the 31-byte file header - reference src/aad_decoder.h:15-16, src/aad_decoder.c:99-170 */
AADApiResult AADDecoder_DecodeHeader(
    const uint8_t *data, uint32_t data_size, struct AADHeaderInfo *header_info);

/* handle lifecycle - reference src/aad_decoder.h:19-25, src/aad_decoder.c:35-96 */
int32_t AADDecoder_CalculateWorkSize(void);
struct AADDecoder *AADDecoder_Create(void *work, int32_t work_size);
void AADDecoder_Destroy(struct AADDecoder *decoder);

/* validate and install a header - reference src/aad_decoder.h:28-29, src/aad_decoder.c:228-253 */
AADApiResult AADDecoder_SetHeader(
    struct AADDecoder *decoder, const struct AADHeaderInfo *header);

/* one block into planar int32 - reference src/aad_decoder.h:32-36, src/aad_decoder.c:321-475 */
AADApiResult AADDecoder_DecodeBlock(
    struct AADDecoder *decoder,
    const uint8_t *data, uint32_t data_size,
    int32_t **buffer, uint32_t buffer_num_channels, uint32_t buffer_num_samples,
    uint32_t *num_decode_samples);

/* header + all blocks - reference src/aad_decoder.h:39-42, src/aad_decoder.c:478-538 */
AADApiResult AADDecoder_DecodeWhole(
    struct AADDecoder *decoder,
    const uint8_t *data, uint32_t data_size,
    int32_t **buffer, uint32_t buffer_num_channels, uint32_t buffer_num_samples);

#ifdef __cplusplus
}
#endif

#endif /* AAD_DECODER_H_INCLDED */
