/*
 * pipeline_signal.c - INTEGRATION.md section 2's two-context pipeline as a real C program (tests/test_gpu_c_pipeline.py builds it
 * with gcc against include/ and libaad_hip.so and runs it on the GPU box): encode on one context, decode on another, the streams
 * ordered by events that the encode kernels carry on their own dispatch packets (AADHip_ContextSignalNextRun).  Every step's
 * decode is checked against the host-memory entry point (AADHip_DecodeBatch) on the images the pipeline produced; the images
 * against AADHip_EncodeBatch.  Prints "ok" and returns 0 when every byte agrees.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "aad_hip.h"
#include "aad_decoder.h"

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define AADCK(x) do { AADApiResult r_ = (x); if (r_ != AAD_APIRESULT_OK) { fprintf(stderr, "%s: %d\n", #x, (int)r_); return 3; } } while (0)

enum { STREAMS = 96, SAMPLES = 2500, CH = 2, RING = 4, STEPS = 11 };

int main(void)
{
  const struct AADEncodeParameter p = {CH, 48000, 4, 1024, AAD_CH_PROCESS_METHOD_NONE, 0};
  const uint64_t image = AADHip_CalculateEncodedSize(&p, SAMPLES), pitch = (image + 63) / 64 * 64;
  if (image == 0) return 1;

  /* STEPS different batches of synthetic PCM on the host (a cheap generator: every step differs) */
  const size_t pcm_elems = (size_t)STREAMS * SAMPLES * CH;
  int16_t *pcm = malloc(sizeof(int16_t) * pcm_elems * STEPS);
  uint32_t s = 12345u;
  for (size_t i = 0; i < pcm_elems * STEPS; i++) {
    s = s * 1664525u + 1013904223u;
    pcm[i] = (int16_t)(((int32_t)(s >> 16) - 32768) / ((i / pcm_elems) % 3 + 1));
  }

  hipStream_t se, sd;
  HIPCK(hipStreamCreateWithFlags(&se, hipStreamNonBlocking));
  HIPCK(hipStreamCreateWithFlags(&sd, hipStreamNonBlocking));
  struct AADHipContext *enc_ctx, *dec_ctx, *host_ctx;
  AADCK(AADHip_ContextCreate(0, se, &enc_ctx));
  AADCK(AADHip_ContextCreate(0, sd, &dec_ctx));
  AADCK(AADHip_ContextCreate(0, NULL, &host_ctx));

  struct AADHipStreamDesc table[STREAMS];
  for (int i = 0; i < STREAMS; i++) {
    table[i].pcm_offset = (uint64_t)i * SAMPLES * CH;
    table[i].data_offset = (uint64_t)i * pitch;
    table[i].data_size = pitch;
    table[i].num_samples = SAMPLES;
    table[i].reserved = 0;
  }
  struct AADHipEncodePlan *enc_plan;
  AADCK(AADHip_EncodePlanCreate(enc_ctx, &p, STREAMS, table, &enc_plan));

  int16_t *d_pcm, *d_out[STEPS];
  uint8_t *d_image[RING];
  HIPCK(hipMalloc((void **)&d_pcm, sizeof(int16_t) * pcm_elems * STEPS));
  HIPCK(hipMemcpy(d_pcm, pcm, sizeof(int16_t) * pcm_elems * STEPS, hipMemcpyHostToDevice));
  for (int k = 0; k < STEPS; k++) HIPCK(hipMalloc((void **)&d_out[k], sizeof(int16_t) * pcm_elems));
  for (int b = 0; b < RING; b++) HIPCK(hipMalloc((void **)&d_image[b], pitch * STREAMS));

  /* the decode plan wants the format: one encode, its header parsed with the reference's own call */
  AADCK(AADHip_EncodePlanRun(enc_plan, d_pcm, d_image[0], NULL));
  AADCK(AADHip_ContextSynchronize(enc_ctx));
  uint8_t head[31];
  HIPCK(hipMemcpy(head, d_image[0], sizeof(head), hipMemcpyDeviceToHost));
  struct AADHeaderInfo format;
  AADCK(AADDecoder_DecodeHeader(head, sizeof(head), &format));
  for (int i = 0; i < STREAMS; i++) table[i].data_size = image;
  struct AADHipDecodePlan *dec_plan;
  AADCK(AADHip_DecodePlanCreate(dec_ctx, &format, 1, STREAMS, table, &dec_plan));

  hipEvent_t encoded[RING], decoded[RING];
  for (int b = 0; b < RING; b++) {
    HIPCK(hipEventCreateWithFlags(&encoded[b], hipEventDisableTiming));
    HIPCK(hipEventCreateWithFlags(&decoded[b], hipEventDisableTiming));
  }
  uint8_t *images = malloc(pitch * STREAMS * STEPS);
  for (int k = 0; k < STEPS; k++) { /* INTEGRATION.md section 2, verbatim in structure */
    const int b = k % RING;
    if (k >= RING) HIPCK(hipStreamWaitEvent(se, decoded[b], 0));             /* image b has been read */
    AADCK(AADHip_ContextSignalNextRun(enc_ctx, NULL, encoded[b]));           /* the encode kernel carries the event itself */
    AADCK(AADHip_EncodePlanRun(enc_plan, d_pcm + (size_t)k * pcm_elems, d_image[b], NULL));
    HIPCK(hipStreamWaitEvent(sd, encoded[b], 0));
    AADCK(AADHip_DecodePlanRun(dec_plan, d_image[b], d_out[k]));
    HIPCK(hipMemcpyAsync(images + (size_t)k * pitch * STREAMS, d_image[b], pitch * STREAMS, hipMemcpyDeviceToHost, sd)); /* keep step k's images */
    HIPCK(hipEventRecord(decoded[b], sd));
  }
  HIPCK(hipDeviceSynchronize());

  /* check every step against the host-memory entry points of a third context */
  int16_t *got = malloc(sizeof(int16_t) * pcm_elems), *want = malloc(sizeof(int16_t) * pcm_elems);
  uint8_t *want_img = malloc(image * STREAMS);
  const int16_t *in_ptr[STREAMS];
  const uint8_t *img_ptr[STREAMS];
  uint8_t *enc_ptr[STREAMS];
  int16_t *out_ptr[STREAMS];
  uint32_t frames[STREAMS], cap[STREAMS], nsamp[STREAMS];
  uint64_t sizes[STREAMS], caps[STREAMS];
  for (int k = 0; k < STEPS; k++) {
    HIPCK(hipMemcpy(got, d_out[k], sizeof(int16_t) * pcm_elems, hipMemcpyDeviceToHost));
    for (int i = 0; i < STREAMS; i++) {
      in_ptr[i] = pcm + (size_t)k * pcm_elems + (size_t)i * SAMPLES * CH;
      img_ptr[i] = images + (size_t)k * pitch * STREAMS + (size_t)i * pitch;
      enc_ptr[i] = want_img + (size_t)i * image;
      out_ptr[i] = want + (size_t)i * SAMPLES * CH;
      sizes[i] = caps[i] = image;
      cap[i] = nsamp[i] = SAMPLES;
    }
    AADCK(AADHip_EncodeBatch(host_ctx, &p, STREAMS, in_ptr, nsamp, enc_ptr, caps, NULL, NULL));
    for (int i = 0; i < STREAMS; i++)
      if (memcmp(enc_ptr[i], img_ptr[i], image) != 0) { fprintf(stderr, "step %d stream %d: image differs\n", k, i); return 4; }
    AADCK(AADHip_DecodeBatch(host_ctx, STREAMS, img_ptr, sizes, out_ptr, cap, frames));
    if (memcmp(got, want, sizeof(int16_t) * pcm_elems) != 0) { fprintf(stderr, "step %d: decoded PCM differs\n", k); return 5; }
    for (int i = 0; i < STREAMS; i++) if (frames[i] != SAMPLES) { fprintf(stderr, "step %d stream %d: %u frames\n", k, i, frames[i]); return 6; }
  }
  AADHip_EncodePlanDestroy(enc_plan);
  AADHip_DecodePlanDestroy(dec_plan);
  AADHip_ContextDestroy(enc_ctx);
  AADHip_ContextDestroy(dec_ctx);
  AADHip_ContextDestroy(host_ctx);
  printf("ok: %d steps x %d streams, images and PCM identical to the host-memory entry points\n", STEPS, STREAMS);
  return 0;
}
