/*
 * host_api_bench.c - measurement aid (not on the product path): wall-clock cost of the host-memory
 * entry points of libaad_hip.so as a C caller sees them, without any Python in the timed region.
 *
 *   1. AADHip_EncodeBatch / AADHip_DecodeBatch on STREAMS stereo 4-bit streams x BLOCKS blocks
 *      (pageable caller buffers; stage + H2D + kernel + D2H + copy back), best and median of REPS;
 *   2. the legacy call pattern of the reference CLI (src/main.c:182-198, :91-106) for ONE stream
 *      of one block: EncodeWhole / DecodeWhole on a warm handle, and the whole
 *      Create -> SetEncodeParameter -> EncodeWhole -> Destroy cycle per file.
 *
 * build: cc -O2 -std=c99 -I include -o build/host_api_bench tools/host_api_bench.c -L aad_amd -laad_hip \
 *           -Wl,-rpath,$PWD/aad_amd -Wl,-rpath,/opt/rocm/lib
 * usage: host_api_bench [streams=1000] [blocks=1] [reps=30] [trials=0]
 * Prints one JSON object.
 */
#define _POSIX_C_SOURCE 200809L
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "aad_decoder.h"
#include "aad_encoder.h"
#include "aad_hip.h"

static double now_ms(void)
{
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

static int cmp_double(const void *a, const void *b)
{
  const double x = *(const double *)a, y = *(const double *)b;
  return x < y ? -1 : x > y;
}

struct Stat { double best, median; };
static struct Stat summarise(double *t, int n)
{
  struct Stat s;
  qsort(t, (size_t)n, sizeof(double), cmp_double);
  s.best = t[0];
  s.median = t[n / 2];
  return s;
}

/* two triangle partials + noise, integer only (timing input; parity is the tests' business) */
static void synth(int16_t *x, uint32_t frames, uint32_t ch, uint64_t seed)
{
  uint64_t s = seed * 0x9E3779B97F4A7C15ull + 1;
  uint32_t p1 = 0, p2 = 0, f, c;
  const uint32_t i1 = (uint32_t)(40000000u + (seed % 97) * 900000u), i2 = (uint32_t)(300000000u + (seed % 89) * 2000000u);
  for (f = 0; f < frames; f++) {
    p1 += i1;
    p2 += i2;
    for (c = 0; c < ch; c++) {
      int32_t t1 = (int32_t)((p1 + c * 0x20000000u) >> 16), t2 = (int32_t)(p2 >> 16), v;
      t1 = t1 < 32768 ? t1 - 16384 : 49152 - t1;
      t2 = t2 < 32768 ? t2 - 16384 : 49152 - t2;
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      v = t1 * 7 / 10 + t2 * 4 / 10 + (int32_t)(s & 0xFFF) - 2048;
      x[(size_t)f * ch + c] = (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
    }
  }
}

int main(int argc, char **argv)
{
  const uint32_t streams = argc > 1 ? (uint32_t)atoi(argv[1]) : 1000;
  const uint32_t blocks = argc > 2 ? (uint32_t)atoi(argv[2]) : 1;
  const int reps = argc > 3 ? atoi(argv[3]) : 30;
  const uint32_t trials = argc > 4 ? (uint32_t)atoi(argv[4]) : 0;
  const uint32_t ch = 2, frames = 992 * blocks;
  struct AADEncodeParameter prm;
  struct AADHipContext *ctx = NULL;
  uint32_t i;
  int r;

  memset(&prm, 0, sizeof(prm));
  prm.num_channels = (uint16_t)ch;
  prm.sampling_rate = 48000;
  prm.bits_per_sample = 4;
  prm.max_block_size = 1024;
  prm.ch_process_method = AAD_CH_PROCESS_METHOD_NONE;
  prm.num_encode_trials = (uint8_t)trials;

  if (AADHip_ContextCreate(0, NULL, &ctx) != AAD_APIRESULT_OK) {
    fprintf(stderr, "no HIP device\n");
    return 1;
  }
  const uint64_t image = AADHip_CalculateEncodedSize(&prm, frames);
  int16_t **pcm = malloc(sizeof(*pcm) * streams), **dec = malloc(sizeof(*dec) * streams);
  uint8_t **img = malloc(sizeof(*img) * streams);
  uint32_t *nsamp = malloc(sizeof(*nsamp) * streams), *got = malloc(sizeof(*got) * streams);
  uint64_t *cap = malloc(sizeof(*cap) * streams), *size = malloc(sizeof(*size) * streams);
  for (i = 0; i < streams; i++) {
    pcm[i] = malloc(sizeof(int16_t) * (size_t)frames * ch);
    dec[i] = malloc(sizeof(int16_t) * (size_t)frames * ch);
    img[i] = malloc(image);
    synth(pcm[i], frames, ch, i);
    nsamp[i] = frames;
    cap[i] = image;
  }
  double *te = malloc(sizeof(double) * (size_t)reps), *td = malloc(sizeof(double) * (size_t)reps);
  for (r = -2; r < reps; r++) { /* two untimed passes: staging allocation, first kernel launches */
    const double t0 = now_ms();
    if (AADHip_EncodeBatch(ctx, &prm, streams, (const int16_t *const *)pcm, nsamp, img, cap, size, NULL) != AAD_APIRESULT_OK) {
      fprintf(stderr, "EncodeBatch failed: %s\n", AADHip_ContextLastError(ctx));
      return 1;
    }
    const double t1 = now_ms();
    if (AADHip_DecodeBatch(ctx, streams, (const uint8_t *const *)img, size, dec, nsamp, got) != AAD_APIRESULT_OK) {
      fprintf(stderr, "DecodeBatch failed: %s\n", AADHip_ContextLastError(ctx));
      return 1;
    }
    const double t2 = now_ms();
    if (r >= 0) {
      te[r] = t1 - t0;
      td[r] = t2 - t1;
    }
  }
  const struct Stat se = summarise(te, reps), sd = summarise(td, reps);
  const double n = (double)streams * frames * ch;

  /* ---- legacy API, one stream of one block (the reference CLI's call pattern) ---- */
  const uint32_t lf = 992;
  int32_t *rows[2], *out_rows[2];
  uint8_t *limg = malloc(4096);
  uint32_t lsize = 0;
  for (i = 0; i < ch; i++) {
    uint32_t k;
    rows[i] = malloc(sizeof(int32_t) * lf);
    out_rows[i] = malloc(sizeof(int32_t) * lf);
    for (k = 0; k < lf; k++) rows[i][k] = pcm[0][(size_t)k * ch + i];
  }
  struct AADEncoder *enc = AADEncoder_Create(1024, NULL, 0);
  struct AADDecoder *dcd = AADDecoder_Create(NULL, 0);
  prm.num_encode_trials = 0;
  AADEncoder_SetEncodeParameter(enc, &prm);
  double *tl = malloc(sizeof(double) * 200), *tdl = malloc(sizeof(double) * 200), *tc = malloc(sizeof(double) * 200);
  for (r = -5; r < 200; r++) {
    const double t0 = now_ms();
    if (AADEncoder_EncodeWhole(enc, (const int32_t *const *)rows, lf, limg, 4096, &lsize) != AAD_APIRESULT_OK) return 2;
    const double t1 = now_ms();
    if (AADDecoder_DecodeWhole(dcd, limg, lsize, out_rows, ch, lf) != AAD_APIRESULT_OK) return 2;
    const double t2 = now_ms();
    if (r >= 0) {
      tl[r] = t1 - t0;
      tdl[r] = t2 - t1;
    }
  }
  for (r = -5; r < 200; r++) { /* a handle per file, as src/main.c does */
    const double t0 = now_ms();
    struct AADEncoder *e = AADEncoder_Create(1024, NULL, 0);
    AADEncoder_SetEncodeParameter(e, &prm);
    if (AADEncoder_EncodeWhole(e, (const int32_t *const *)rows, lf, limg, 4096, &lsize) != AAD_APIRESULT_OK) return 2;
    AADEncoder_Destroy(e);
    if (r >= 0) tc[r] = now_ms() - t0;
  }
  const struct Stat sl = summarise(tl, 200), sdl = summarise(tdl, 200), sc = summarise(tc, 200);

  printf("{\"workload\": \"%u stereo 4-bit streams x %u block(s), trials %u, pageable caller buffers\", "
         "\"encode_batch_ms\": {\"best\": %.4f, \"median\": %.4f}, \"decode_batch_ms\": {\"best\": %.4f, \"median\": %.4f}, "
         "\"encode_msps\": %.1f, \"decode_msps\": %.1f, "
         "\"legacy_one_block\": {\"encode_whole_us\": {\"best\": %.1f, \"median\": %.1f}, "
         "\"decode_whole_us\": {\"best\": %.1f, \"median\": %.1f}, "
         "\"create_set_encode_destroy_us\": {\"best\": %.1f, \"median\": %.1f}}}\n",
         streams, blocks, trials, se.best, se.median, sd.best, sd.median, n / se.median / 1e3, n / sd.median / 1e3,
         sl.best * 1e3, sl.median * 1e3, sdl.best * 1e3, sdl.median * 1e3, sc.best * 1e3, sc.median * 1e3);
  AADEncoder_Destroy(enc);
  AADDecoder_Destroy(dcd);
  AADHip_ContextDestroy(ctx);
  return 0;
}
