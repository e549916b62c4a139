/*
 * aad_legacy_api.c - the reference's 14 public functions (src/aad_encoder.h:25-50,
 * src/aad_decoder.h:15-42) as a thin host-C layer over the batched HIP engine.
 *
 * What stays on the host, in C, exactly as in the reference: argument checks, return codes,
 * handle placement in caller memory, header (de)serialisation, block geometry.  What moves to
 * the GPU: every per-sample loop.  EncodeWhole / DecodeWhole / DecodeBlock stage their buffers
 * and call the engine with a batch of one stream; a HIP context is created on first use and
 * released in Destroy.  There is no CPU codec in this library: without a usable device those
 * three calls return AAD_APIRESULT_NG.
 */
#include <stdlib.h>
#include <string.h>

#include "../../include/aad_decoder.h"
#include "../../include/aad_encoder.h"
#include "../../include/aad_hip.h"
#include "aad_format.h"
#include "aad_hip_internal.h"

#define AAD_HANDLE_ALIGNMENT 16 /* reference src/aad_internal.h:7 */

static uintptr_t align_up(uintptr_t v) { return (v + AAD_HANDLE_ALIGNMENT - 1) & ~(uintptr_t)(AAD_HANDLE_ALIGNMENT - 1); }

static int32_t default_device(void)
{
  const char *e = getenv("AAD_HIP_DEVICE");
  return e != NULL ? (int32_t)atoi(e) : 0;
}

/* ============================================================================ encoder ==== */

struct AADEncoder {
  struct AADHeaderInfo header;
  struct AADHipLaneState lane[AAD_MAX_NUM_CHANNELS]; /* persists across EncodeWhole calls (SURVEY.md section 7 traps) */
  struct AADEncodeParameter parameter;
  struct AADHipContext *context; /* created on first encode */
  void *work;
  uint8_t set_parameter;
  uint8_t owns_work;
};

AADApiResult AADEncoder_CalculateBlockSize(uint16_t max_block_size, uint16_t num_channels, uint32_t bits_per_sample,
                                           uint16_t *block_size, uint32_t *num_samples_per_block)
{
  return AADFormat_BlockGeometry(max_block_size, num_channels, bits_per_sample, AAD_MAX_NUM_CHANNELS,
                                 block_size, num_samples_per_block);
}

AADApiResult AADEncoder_EncodeHeader(const struct AADHeaderInfo *header_info, uint8_t *data, uint32_t data_size)
{
  if (header_info == NULL || data == NULL) return AAD_APIRESULT_INVALID_ARGUMENT;
  if (data_size < AAD_HEADER_SIZE) return AAD_APIRESULT_INSUFFICIENT_DATA;
  if (!AADFormat_HeaderFieldsValid(header_info, AAD_MAX_NUM_CHANNELS)) return AAD_APIRESULT_INVALID_FORMAT;
  AADFormat_PutHeader(header_info, data);
  return AAD_APIRESULT_OK;
}

int32_t AADEncoder_CalculateWorkSize(uint16_t max_block_size)
{
  uint16_t block_size;
  /* same acceptance rule as the reference (mono, 2-bit geometry must exist: src/aad_encoder.c:232-236) */
  if (AADFormat_BlockGeometry(max_block_size, 1, AAD_MIN_BITS_PER_SAMPLE, AAD_MAX_NUM_CHANNELS, &block_size, NULL)
      != AAD_APIRESULT_OK)
    return -1;
  return (int32_t)(AAD_HANDLE_ALIGNMENT + sizeof(struct AADEncoder));
}

struct AADEncoder *AADEncoder_Create(uint16_t max_block_size, void *work, int32_t work_size)
{
  struct AADEncoder *enc;
  uint8_t owns = 0;
  const int32_t need = AADEncoder_CalculateWorkSize(max_block_size);
  if (need < 0) return NULL;
  if (work == NULL && work_size == 0) {
    work = malloc((size_t)need);
    work_size = need;
    owns = 1;
  }
  if (work == NULL || work_size < need) return NULL;
  enc = (struct AADEncoder *)align_up((uintptr_t)work);
  memset(enc, 0, sizeof(*enc)); /* zero weights and history: src/aad_encoder.c:299-301 */
  enc->work = work;
  enc->owns_work = owns;
  return enc;
}

void AADEncoder_Destroy(struct AADEncoder *encoder)
{
  if (encoder == NULL) return;
  AADHip_ContextDestroy(encoder->context);
  encoder->context = NULL;
  if (encoder->owns_work) free(encoder->work);
}

AADApiResult AADEncoder_SetEncodeParameter(struct AADEncoder *encoder, const struct AADEncodeParameter *parameter)
{
  struct AADHeaderInfo h;
  uint32_t c;
  if (encoder == NULL || parameter == NULL) return AAD_APIRESULT_INVALID_ARGUMENT;
  if (AADFormat_ParameterToHeader(parameter, 0, AAD_MAX_NUM_CHANNELS, &h) != AAD_APIRESULT_OK)
    return AAD_APIRESULT_INVALID_FORMAT;
  /* only the step index is reset here, the weights carry over: src/aad_encoder.c:797-799 */
  for (c = 0; c < AAD_MAX_NUM_CHANNELS; c++) encoder->lane[c].stepsize_index = 0;
  encoder->parameter = *parameter;
  encoder->header = h;
  encoder->set_parameter = 1;
  return AAD_APIRESULT_OK;
}

AADApiResult AADEncoder_EncodeWhole(struct AADEncoder *encoder, const int32_t *const *input, uint32_t num_samples,
                                    uint8_t *data, uint32_t data_size, uint32_t *output_size)
{
  AADApiResult rc;
  uint64_t need, produced = 0, capacity;
  uint32_t ch, c, s;
  int16_t *pcm;
  const int16_t *pcm_list[1];
  uint8_t *data_list[1];

  if (encoder == NULL || input == NULL || data == NULL || output_size == NULL) return AAD_APIRESULT_INVALID_ARGUMENT;
  if (!encoder->set_parameter) return AAD_APIRESULT_PARAMETER_NOT_SET;
  encoder->header.num_samples = num_samples;
  /* the header goes out first and carries every format check: src/aad_encoder.c:840-844 */
  if ((rc = AADEncoder_EncodeHeader(&encoder->header, data, data_size)) != AAD_APIRESULT_OK) return rc;
  need = AADFormat_EncodedSize(&encoder->header);
  if (data_size < need) return AAD_APIRESULT_INSUFFICIENT_BUFFER;

  if (encoder->context == NULL &&
      AADHip_ContextCreate(default_device(), NULL, &encoder->context) != AAD_APIRESULT_OK)
    return AAD_APIRESULT_NG;

  ch = encoder->header.num_channels;
  for (c = 0; c < ch; c++)
    if (input[c] == NULL) return AAD_APIRESULT_INVALID_ARGUMENT;
  pcm = (int16_t *)malloc(sizeof(int16_t) * (size_t)num_samples * ch);
  if (pcm == NULL) return AAD_APIRESULT_NG;
  /* planar int32 -> interleaved int16 (the device layout); samples must already be in int16
   * range, which the reference only asserts (src/aad_encoder.c:612) - out-of-range input saturates */
  for (c = 0; c < ch; c++) {
    const int32_t *x = input[c];
    for (s = 0; s < num_samples; s++) {
      const int32_t v = x[s];
      pcm[(size_t)s * ch + c] = (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
    }
  }
  pcm_list[0] = pcm;
  data_list[0] = data;
  capacity = data_size;
  rc = AADHip_EncodeBatch(encoder->context, &encoder->parameter, 1, pcm_list, &num_samples,
                          data_list, &capacity, &produced, encoder->lane);
  free(pcm);
  if (rc != AAD_APIRESULT_OK) return rc;
  *output_size = (uint32_t)produced;
  return AAD_APIRESULT_OK;
}

/* ============================================================================ decoder ==== */

struct AADDecoder {
  struct AADHeaderInfo header;
  struct AADHipContext *context;
  void *work;
  uint8_t set_header;
  uint8_t owns_work;
};

int32_t AADDecoder_CalculateWorkSize(void) { return (int32_t)(AAD_HANDLE_ALIGNMENT + sizeof(struct AADDecoder)); }

struct AADDecoder *AADDecoder_Create(void *work, int32_t work_size)
{
  struct AADDecoder *dec;
  uint8_t owns = 0;
  const int32_t need = AADDecoder_CalculateWorkSize();
  if (work == NULL && work_size == 0) {
    work = malloc((size_t)need);
    work_size = need;
    owns = 1;
  }
  if (work == NULL || work_size < need) return NULL;
  dec = (struct AADDecoder *)align_up((uintptr_t)work);
  memset(dec, 0, sizeof(*dec));
  dec->work = work;
  dec->owns_work = owns;
  return dec;
}

void AADDecoder_Destroy(struct AADDecoder *decoder)
{
  if (decoder == NULL) return;
  AADHip_ContextDestroy(decoder->context);
  decoder->context = NULL;
  if (decoder->owns_work) free(decoder->work);
}

AADApiResult AADDecoder_DecodeHeader(const uint8_t *data, uint32_t data_size, struct AADHeaderInfo *header_info)
{
  struct AADHeaderInfo h;
  if (data == NULL || header_info == NULL) return AAD_APIRESULT_INVALID_ARGUMENT;
  if (data_size < AAD_HEADER_SIZE) return AAD_APIRESULT_INSUFFICIENT_DATA;
  if (!AADFormat_GetHeader(data, &h)) return AAD_APIRESULT_INVALID_FORMAT;
  *header_info = h; /* parsed, not validated - validation is SetHeader's job (src/aad_decoder.c:134) */
  return AAD_APIRESULT_OK;
}

AADApiResult AADDecoder_SetHeader(struct AADDecoder *decoder, const struct AADHeaderInfo *header)
{
  if (decoder == NULL || header == NULL) return AAD_APIRESULT_INVALID_ARGUMENT;
  if (!AADFormat_HeaderAcceptedByDecoder(header, AAD_MAX_NUM_CHANNELS)) return AAD_APIRESULT_INVALID_FORMAT;
  decoder->header = *header;
  decoder->set_header = 1;
  return AAD_APIRESULT_OK;
}

/* run the engine on one image / one bare block and widen the frames into the caller's planar int32 */
static AADApiResult decode_into_planar(struct AADDecoder *decoder, int32_t has_file_header,
                                       const uint8_t *data, uint32_t data_size, uint32_t want_frames,
                                       int32_t **buffer, uint32_t *got_frames)
{
  const uint32_t ch = decoder->header.num_channels;
  const uint8_t *data_list[1];
  int16_t *pcm_list[1];
  uint64_t size64 = data_size;
  uint32_t decoded = 0, c, s;
  AADApiResult rc;
  int16_t *pcm;

  for (c = 0; c < ch; c++)
    if (buffer[c] == NULL) return AAD_APIRESULT_INVALID_ARGUMENT;
  if (want_frames == 0) {
    *got_frames = 0;
    return AAD_APIRESULT_OK;
  }
  if (decoder->context == NULL &&
      AADHip_ContextCreate(default_device(), NULL, &decoder->context) != AAD_APIRESULT_OK)
    return AAD_APIRESULT_NG;
  pcm = (int16_t *)malloc(sizeof(int16_t) * (size_t)want_frames * ch);
  if (pcm == NULL) return AAD_APIRESULT_NG;
  data_list[0] = data;
  pcm_list[0] = pcm;
  rc = AADHipInternal_DecodeHost(decoder->context, &decoder->header, has_file_header, 1,
                                 data_list, &size64, &want_frames, pcm_list, &decoded);
  if (rc == AAD_APIRESULT_OK) {
    for (c = 0; c < ch; c++)
      for (s = 0; s < decoded; s++) buffer[c][s] = pcm[(size_t)s * ch + c];
    *got_frames = decoded;
  }
  free(pcm);
  return rc;
}

AADApiResult AADDecoder_DecodeBlock(struct AADDecoder *decoder, const uint8_t *data, uint32_t data_size,
                                    int32_t **buffer, uint32_t buffer_num_channels, uint32_t buffer_num_samples,
                                    uint32_t *num_decode_samples)
{
  uint32_t want;
  if (decoder == NULL || data == NULL || buffer == NULL || num_decode_samples == NULL)
    return AAD_APIRESULT_INVALID_ARGUMENT;
  if (!decoder->set_header) return AAD_APIRESULT_PARAMETER_NOT_SET;
  if (data_size < AAD_BLOCK_HEADER_BYTES_PER_CH * (uint32_t)decoder->header.num_channels)
    return AAD_APIRESULT_INSUFFICIENT_DATA;
  if (buffer_num_channels < decoder->header.num_channels) return AAD_APIRESULT_INSUFFICIENT_BUFFER;
  /* a short buffer decodes until it is full: src/aad_decoder.c:354-356 */
  want = decoder->header.num_samples_per_block < buffer_num_samples ? decoder->header.num_samples_per_block
                                                                    : buffer_num_samples;
  return decode_into_planar(decoder, 0, data, data_size, want, buffer, num_decode_samples);
}

AADApiResult AADDecoder_DecodeWhole(struct AADDecoder *decoder, const uint8_t *data, uint32_t data_size,
                                    int32_t **buffer, uint32_t buffer_num_channels, uint32_t buffer_num_samples)
{
  struct AADHeaderInfo h;
  uint32_t got = 0;
  AADApiResult rc;
  if (decoder == NULL || data == NULL || buffer == NULL) return AAD_APIRESULT_INVALID_ARGUMENT;
  if ((rc = AADDecoder_DecodeHeader(data, data_size, &h)) != AAD_APIRESULT_OK) return rc;
  if ((rc = AADDecoder_SetHeader(decoder, &h)) != AAD_APIRESULT_OK) return rc;
  if (buffer_num_channels < h.num_channels || buffer_num_samples < h.num_samples)
    return AAD_APIRESULT_INSUFFICIENT_BUFFER;
  /* exactly header.num_samples frames are produced; the reference may run past them into
   * whatever follows when handed a larger buffer (src/aad_decoder.c:524), this does not */
  return decode_into_planar(decoder, 1, data, data_size, h.num_samples, buffer, &got);
}
