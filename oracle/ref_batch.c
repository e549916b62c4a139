/*
 * ref_batch.c - test/measurement infrastructure: a batch loop over the REFERENCE's public API
 * (AADEncoder_* / AADDecoder_*, reference src/aad_encoder.h:25-50, src/aad_decoder.h:15-42), linked
 * into oracle/_ref/libaadref.so next to the reference's own objects.  Written here, not copied:
 * it is the call sequence of the reference CLI (src/main.c:175-198 encode, :91-126 decode) around
 * many in-memory streams, so that bench.py's cpu_baseline can time the reference without a Python
 * round trip per stream (round-1 advice: ~20 us of ctypes per call understated the CPU figure).
 *
 * The planar int32 <-> interleaved int16 conversion the CLI does around the codec
 * (src/main.c:175-179, :122-126) is done OUTSIDE the timed calls by refbatch_planar_from_pcm.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "aad_decoder.h"
#include "aad_encoder.h"

/* interleaved int16 [streams][samples][ch] -> planar int32 [streams][ch][samples] */
void refbatch_planar_from_pcm(const int16_t *pcm, uint32_t streams, uint32_t samples, uint32_t ch, int32_t *planar)
{
  uint32_t s, c, n;
  for (s = 0; s < streams; s++)
    for (c = 0; c < ch; c++)
      for (n = 0; n < samples; n++)
        planar[((size_t)s * ch + c) * samples + n] = pcm[((size_t)s * samples + n) * ch + c];
}

/* one fresh encoder handle per stream, as the CLI does per file; returns 0 or the failing AADApiResult */
int refbatch_encode(const int32_t *planar, uint32_t streams, uint32_t samples, uint32_t ch, uint32_t bits,
                    uint32_t max_block_size, uint32_t trials, uint8_t *out, size_t out_stride, uint32_t *out_sizes)
{
  struct AADEncodeParameter p;
  const int32_t *rows[8];
  uint32_t s, c;
  memset(&p, 0, sizeof(p));
  p.num_channels = (uint16_t)ch;
  p.sampling_rate = 48000;
  p.bits_per_sample = (uint16_t)bits;
  p.max_block_size = (uint16_t)max_block_size;
  p.ch_process_method = AAD_CH_PROCESS_METHOD_NONE;
  p.num_encode_trials = (uint8_t)trials;
  if (ch > 8) return AAD_APIRESULT_INVALID_ARGUMENT;
  for (s = 0; s < streams; s++) {
    struct AADEncoder *e = AADEncoder_Create((uint16_t)max_block_size, NULL, 0);
    AADApiResult rc;
    uint32_t size = 0;
    if (e == NULL) return AAD_APIRESULT_NG;
    for (c = 0; c < ch; c++) rows[c] = planar + ((size_t)s * ch + c) * samples;
    rc = AADEncoder_SetEncodeParameter(e, &p);
    if (rc == AAD_APIRESULT_OK) rc = AADEncoder_EncodeWhole(e, rows, samples, out + (size_t)s * out_stride, (uint32_t)out_stride, &size);
    AADEncoder_Destroy(e);
    if (rc != AAD_APIRESULT_OK) return (int)rc;
    if (out_sizes) out_sizes[s] = size;
  }
  return 0;
}

/* decode every image into planar int32 [streams][ch][samples] */
int refbatch_decode(const uint8_t *data, uint32_t streams, size_t stride, const uint32_t *sizes, uint32_t samples,
                    uint32_t ch, int32_t *planar)
{
  int32_t *rows[8];
  uint32_t s, c;
  if (ch > 8) return AAD_APIRESULT_INVALID_ARGUMENT;
  for (s = 0; s < streams; s++) {
    struct AADDecoder *d = AADDecoder_Create(NULL, 0);
    AADApiResult rc;
    if (d == NULL) return AAD_APIRESULT_NG;
    for (c = 0; c < ch; c++) rows[c] = planar + ((size_t)s * ch + c) * samples;
    rc = AADDecoder_DecodeWhole(d, data + (size_t)s * stride, sizes[s], rows, ch, samples);
    AADDecoder_Destroy(d);
    if (rc != AAD_APIRESULT_OK) return (int)rc;
  }
  return 0;
}
