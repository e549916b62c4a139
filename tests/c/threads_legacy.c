/*
 * threads_legacy.c - the threading contract of the drop-in boundary (SURVEY.md section 8b "Threading": the reference has no
 * mutable globals - its tables are static const, reference src/aad_tables.c:8,58 - so DISTINCT handles are usable from
 * DISTINCT threads), as a C99 + pthreads program against include/ and libaad_hip.so.
 *
 * THREADS threads, ROUNDS rounds each, every round the reference CLI's two call sequences on handles of its own:
 *     AADEncoder_Create -> SetEncodeParameter -> EncodeWhole -> Destroy             (reference src/main.c:182-198)
 *     AADDecoder_Create -> DecodeHeader -> DecodeWhole -> Destroy                   (reference src/main.c:91-106)
 * with parameters that differ per (thread, round): 2/3/4 bits, mono / stereo, M/S, 0-2 encode trials, four block sizes,
 * ragged lengths.  This is what aad_legacy_api.c's process-wide context pool (a mutex, eight slots) sees under contention:
 * more threads than slots, handles created and destroyed all the time.
 *
 * Every image and every decode is compared byte for byte with the PARITY ORACLE (oracle/libaad_oracle.so - test
 * infrastructure, linked here as the checker only).  Prints "ok: ..." and returns 0 when everything agrees.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "aad_decoder.h"
#include "aad_encoder.h"
#include "aad_synth.h"
#include "aad_oracle.h"

enum { THREADS = 8, ROUNDS = 50, MAX_SAMPLES = 6000 };

struct Job {
  int thread;
  int failures;
  unsigned long samples_done;
  char first_failure[256];
};

static void fail(struct Job *job, const char *what, int round, int detail)
{
  if (job->failures++ == 0) snprintf(job->first_failure, sizeof(job->first_failure), "thread %d round %d: %s (%d)", job->thread, round, what, detail);
}

static void *worker(void *arg)
{
  struct Job *job = (struct Job *)arg;
  static const uint16_t kBlockSizes[4] = {1024, 256, 600, 2048};
  int16_t *pcm = malloc(sizeof(int16_t) * MAX_SAMPLES * 2);
  int16_t *want_pcm = malloc(sizeof(int16_t) * MAX_SAMPLES * 2);
  int32_t *planar = malloc(sizeof(int32_t) * MAX_SAMPLES * 2);
  int32_t *decoded = malloc(sizeof(int32_t) * MAX_SAMPLES * 2);
  const size_t cap = (size_t)MAX_SAMPLES * 4 + 4096;
  uint8_t *image = malloc(cap), *want = malloc(cap);
  uint32_t x = 2463534242u + 7919u * (uint32_t)job->thread;
  int round;
  for (round = 0; round < ROUNDS; round++) {
    struct AADEncodeParameter p;
    struct AADHeaderInfo hd;
    struct AADEncoder *enc;
    struct AADDecoder *dec;
    const int32_t *rows[2];
    int32_t *out_rows[2];
    AadoLane lanes[AADO_MAX_CHANNELS];
    uint32_t n, c, i, size = 0;
    size_t want_size = 0;
    AADApiResult rc;
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    memset(&p, 0, sizeof(p));
    p.num_channels = (uint16_t)(1 + (x & 1));
    p.sampling_rate = 48000;
    p.bits_per_sample = (uint16_t)(2 + (x >> 1) % 3);
    p.max_block_size = kBlockSizes[(x >> 4) & 3];
    p.ch_process_method = (p.num_channels == 2 && ((x >> 6) & 3) == 0) ? AAD_CH_PROCESS_METHOD_MS : AAD_CH_PROCESS_METHOD_NONE;
    p.num_encode_trials = (uint8_t)((x >> 8) % 3);
    n = 1 + (x >> 10) % MAX_SAMPLES;
    if (AADSynth_Generate(pcm, 1, n, p.num_channels, 1000u + (uint64_t)job->thread, 48000, (int32_t)((x >> 24) % 3),
                          (uint64_t)round) != 0) { fail(job, "AADSynth_Generate", round, 0); break; }
    for (c = 0; c < p.num_channels; c++)
      for (i = 0; i < n; i++) planar[c * MAX_SAMPLES + i] = pcm[i * p.num_channels + c];
    rows[0] = planar;
    rows[1] = planar + MAX_SAMPLES;

    /* encode: a handle per file, as the reference CLI does */
    enc = AADEncoder_Create(p.max_block_size, NULL, 0);
    if (enc == NULL) { fail(job, "AADEncoder_Create", round, 0); break; }
    rc = AADEncoder_SetEncodeParameter(enc, &p);
    if (rc == AAD_APIRESULT_OK) rc = AADEncoder_EncodeWhole(enc, rows, n, image, (uint32_t)cap, &size);
    AADEncoder_Destroy(enc);
    if (rc != AAD_APIRESULT_OK) { fail(job, "AADEncoder_EncodeWhole", round, (int)rc); continue; }
    memset(lanes, 0, sizeof(lanes));
    if (aado_encode_stream(pcm, n, p.num_channels, 48000, p.bits_per_sample, p.max_block_size, p.ch_process_method,
                           p.num_encode_trials, lanes, want, cap, &want_size) != AADO_OK) { fail(job, "oracle encode", round, 0); continue; }
    if (want_size != size || memcmp(image, want, size) != 0) { fail(job, "image differs from the oracle's", round, (int)size); continue; }

    /* decode */
    dec = AADDecoder_Create(NULL, 0);
    if (dec == NULL) { fail(job, "AADDecoder_Create", round, 0); break; }
    out_rows[0] = decoded;
    out_rows[1] = decoded + MAX_SAMPLES;
    rc = AADDecoder_DecodeHeader(image, size, &hd);
    if (rc == AAD_APIRESULT_OK) rc = AADDecoder_DecodeWhole(dec, image, size, out_rows, hd.num_channels, hd.num_samples);
    AADDecoder_Destroy(dec);
    if (rc != AAD_APIRESULT_OK || hd.num_samples != n) { fail(job, "AADDecoder_DecodeWhole", round, (int)rc); continue; }
    if (aado_decode_stream(want, want_size, 2, want_pcm, n, NULL) != AADO_OK) { fail(job, "oracle decode", round, 0); continue; }
    for (c = 0; c < p.num_channels; c++)
      for (i = 0; i < n; i++)
        if (decoded[c * MAX_SAMPLES + i] != want_pcm[i * p.num_channels + c]) { fail(job, "decode differs from the oracle's", round, (int)i); i = n; c = p.num_channels; }
    job->samples_done += (unsigned long)n * p.num_channels;
  }
  free(pcm); free(want_pcm); free(planar); free(decoded); free(image); free(want);
  return NULL;
}

int main(void)
{
  pthread_t tid[THREADS];
  struct Job jobs[THREADS];
  unsigned long samples = 0;
  int t, failures = 0;
  memset(jobs, 0, sizeof(jobs));
  for (t = 0; t < THREADS; t++) {
    jobs[t].thread = t;
    if (pthread_create(&tid[t], NULL, worker, &jobs[t]) != 0) { fprintf(stderr, "pthread_create failed\n"); return 2; }
  }
  for (t = 0; t < THREADS; t++) pthread_join(tid[t], NULL);
  for (t = 0; t < THREADS; t++) {
    failures += jobs[t].failures;
    samples += jobs[t].samples_done;
    if (jobs[t].failures) fprintf(stderr, "%s (+%d more)\n", jobs[t].first_failure, jobs[t].failures - 1);
  }
  if (failures) return 1;
  printf("ok: %d threads x %d rounds of Create/SetEncodeParameter/EncodeWhole/Destroy + Create/DecodeHeader/DecodeWhole/Destroy, "
         "%lu channel-samples, every image and decode byte-equal to the oracle\n", THREADS, ROUNDS, samples);
  return 0;
}
