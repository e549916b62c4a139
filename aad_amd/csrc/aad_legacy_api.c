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
#include <pthread.h>
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

/*
 * HIP contexts of the legacy handles.  The reference's callers create a handle per file
 * (src/main.c:182-198: Create -> SetEncodeParameter -> EncodeWhole -> Destroy); a context per handle
 * would pay a stream plus pinned and device staging allocations (milliseconds) for a kernel of tens
 * of microseconds.  Destroy therefore parks the context in a small process-wide pool and the next
 * handle's first encode / decode takes it from there, buffers and all.  A context is owned by one
 * handle at a time, so distinct handles stay usable from distinct threads as in the reference.
 * Parked contexts are left to process teardown (the HIP runtime may already be gone in an atexit).
 */
#define AAD_CONTEXT_POOL_SLOTS 8
static pthread_mutex_t pool_lock = PTHREAD_MUTEX_INITIALIZER;
static struct {
  struct AADHipContext *context;
  int32_t device;
} pool[AAD_CONTEXT_POOL_SLOTS];

static struct AADHipContext *acquire_context(void)
{
  const int32_t device = default_device();
  struct AADHipContext *ctx = NULL;
  int i;
  pthread_mutex_lock(&pool_lock);
  for (i = 0; i < AAD_CONTEXT_POOL_SLOTS && ctx == NULL; i++) {
    if (pool[i].context != NULL && pool[i].device == device) {
      ctx = pool[i].context;
      pool[i].context = NULL;
    }
  }
  pthread_mutex_unlock(&pool_lock);
  if (ctx != NULL) {
    AADHipInternal_ContextOptionsFromEnvironment(ctx); /* as a freshly created context would */
    return ctx;
  }
  if (AADHip_ContextCreate(device, NULL, &ctx) != AAD_APIRESULT_OK) return NULL;
  return ctx;
}

static void release_context(struct AADHipContext *ctx)
{
  int i;
  if (ctx == NULL) return;
  pthread_mutex_lock(&pool_lock);
  for (i = 0; i < AAD_CONTEXT_POOL_SLOTS; i++) {
    if (pool[i].context == NULL) {
      pool[i].context = ctx;
      pool[i].device = AADHipInternal_ContextDevice(ctx);
      ctx = NULL;
      break;
    }
  }
  pthread_mutex_unlock(&pool_lock);
  AADHip_ContextDestroy(ctx); /* pool full (NULL otherwise: a no-op) */
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
  release_context(encoder->context);
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
  uint64_t need, produced = 0;
  uint32_t ch, c;

  if (encoder == NULL || input == NULL || data == NULL || output_size == NULL) return AAD_APIRESULT_INVALID_ARGUMENT;
  if (!encoder->set_parameter) return AAD_APIRESULT_PARAMETER_NOT_SET;
  encoder->header.num_samples = num_samples;
  /* the header goes out first and carries every format check: src/aad_encoder.c:840-844 */
  if ((rc = AADEncoder_EncodeHeader(&encoder->header, data, data_size)) != AAD_APIRESULT_OK) return rc;
  need = AADFormat_EncodedSize(&encoder->header);
  if (data_size < need) return AAD_APIRESULT_INSUFFICIENT_BUFFER;

  if (encoder->context == NULL && (encoder->context = acquire_context()) == NULL) return AAD_APIRESULT_NG;

  ch = encoder->header.num_channels;
  for (c = 0; c < ch; c++)
    if (input[c] == NULL) return AAD_APIRESULT_INVALID_ARGUMENT;
  rc = AADHipInternal_EncodePlanar32(encoder->context, &encoder->parameter, input, num_samples, data, data_size,
                                     &produced, encoder->lane);
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
  release_context(decoder->context);
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
  uint32_t c;

  for (c = 0; c < ch; c++)
    if (buffer[c] == NULL) return AAD_APIRESULT_INVALID_ARGUMENT;
  if (want_frames == 0) {
    *got_frames = 0;
    return AAD_APIRESULT_OK;
  }
  if (decoder->context == NULL && (decoder->context = acquire_context()) == NULL) return AAD_APIRESULT_NG;
  return AADHipInternal_DecodePlanar32(decoder->context, &decoder->header, has_file_header, data, data_size,
                                       want_frames, buffer, got_frames);
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
