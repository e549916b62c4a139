/*
 * aad.h - common types of the AAD codec C API (codec v18, .aad format v4).
 *
 * Drop-in for reference src/aad.h:7-53: same constants, the same AADApiResult numbering
 * (the reference CLI prints the integer, src/main.c:96), the same AADChannelProcessMethod
 * values and a layout-compatible struct AADHeaderInfo.  The implementation behind these
 * types is the MI355X HIP engine in aad_amd/csrc (no CPU codec is compiled into the library).
 */
#ifndef AAD_H_INCLDED
#define AAD_H_INCLDED

#include <stdint.h>

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

#endif /* AAD_H_INCLDED */
