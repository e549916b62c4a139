/*
 * aad_api.h - the whole drop-in C API of the AAD codec (codec v18, .aad format v4) in one place:
 * common types, the encoder half and the decoder half.  The reference spreads these over
 * src/aad.h:7-53, src/aad_encoder.h:8-50 and src/aad_decoder.h:8-42; this repo keeps files of
 * those three names too (include/aad.h, aad_encoder.h, aad_decoder.h) so that existing
 * `#include "aad_encoder.h"` lines keep working - each of them simply includes this header.
 * Same symbols, argument meaning, struct layouts and return codes as the reference; the
 * implementation behind them is the MI355X HIP engine (aad_amd/csrc), there is no CPU codec in
 * the library.  Observable differences are listed in INTEGRATION.md.
 */
#ifndef AAD_API_H_INCLUDED
#define AAD_API_H_INCLUDED

#include <stdint.h>

/* ======================================================================= common types == */

#define AAD_CODEC_VERSION        18 /* reference src/aad.h:7  */
#define AAD_FORMAT_VERSION       4  /* reference src/aad.h:10 */
#define AAD_MAX_NUM_CHANNELS     2  /* legacy API limit, reference src/aad.h:13; the batched API (aad_hip.h) takes up to 8 */
#define AAD_MIN_BITS_PER_SAMPLE  2  /* reference src/aad.h:16 */
#define AAD_MAX_BITS_PER_SAMPLE  4  /* reference src/aad.h:19 */
#define AAD_HEADER_SIZE          31 /* bytes, reference src/aad.h:22 */

/* API result codes - values 0..6 in this order (reference src/aad.h:25-33) */
typedef enum AADApiResultTag {
  AAD_APIRESULT_OK = 0,
  AAD_APIRESULT_INVALID_ARGUMENT,
  AAD_APIRESULT_INVALID_FORMAT,
  AAD_APIRESULT_INSUFFICIENT_BUFFER,
  AAD_APIRESULT_INSUFFICIENT_DATA,
  AAD_APIRESULT_PARAMETER_NOT_SET,
  AAD_APIRESULT_NG
} AADApiResult;

/* multi-channel processing (reference src/aad.h:36-40) */
typedef enum AADChannelProcessMethodTag {
  AAD_CH_PROCESS_METHOD_NONE = 0,
  AAD_CH_PROCESS_METHOD_MS,      /* stereo mid/side */
  AAD_CH_PROCESS_METHOD_INVALID
} AADChannelProcessMethod;

/* decoded file header (reference src/aad.h:43-53) */
struct AADHeaderInfo {
  uint32_t format_version;
  uint32_t codec_version;
  uint16_t num_channels;
  uint32_t num_samples;            /* per channel */
  uint32_t sampling_rate;
  uint16_t bits_per_sample;
  uint16_t block_size;             /* bytes */
  uint32_t num_samples_per_block;  /* per channel */
  AADChannelProcessMethod ch_process_method;
};

/* ============================================================================ encoder == */
/*
 * Differences a caller can observe (success-path bytes are identical):
 *   - the handle's work area is smaller (the device owns the block buffers);
 *   - AADEncoder_EncodeWhole returns AAD_APIRESULT_INSUFFICIENT_BUFFER instead of writing past
 *     data_size (the reference only asserts, src/aad_encoder.c:666-667);
 *   - AADEncoder_EncodeWhole returns AAD_APIRESULT_NG if no HIP device is usable.
 */

#ifdef __cplusplus
extern "C" {
#endif

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

/* ============================================================================ decoder == */
/* Bytes past data_size decode as zero (the reference reads out of bounds on a truncated block,
 * src/aad_decoder.c:396-451). */

struct AADDecoder; /* opaque */


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

#endif /* AAD_API_H_INCLUDED */
