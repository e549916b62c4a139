/*
 * aad_batch - many-file front end of the MI355X AAD engine.
 *
 * The reference CLI handles one file per process (src/main.c:518-625) through a bit-serial WAV
 * reader; a GPU needs many independent streams in flight (encode: stream x channel lanes).  This
 * tool keeps the reference's encode options and defaults (src/main.c:20-58: -b 4, -s 1024, -t 2,
 * -m 0) and feeds ALL input files to one AADHip_EncodeBatch / AADHip_DecodeBatch call per format:
 *
 *   aad_batch -e [-b bits] [-s max_block_size] [-t trials] [-m 0|1] -o OUTDIR in1.wav in2.wav ...
 *   aad_batch -d -o OUTDIR in1.aad in2.aad ...
 *
 * Every output is byte-identical to what `aad -e` / `aad -d` of the reference writes for the
 * same input (16-bit PCM WAV; the payload is used as the device PCM layout without conversion).
 * Host C only; all codec work happens in libaad_hip.so.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/aad_hip.h"
#include "../../include/aad_wav.h"

struct File {
  const char *path;
  uint8_t *bytes;
  uint64_t size;
  struct AADWavInfo wav;      /* encode */
  struct AADHeaderInfo head;  /* decode */
  uint8_t *out;
  uint64_t out_size;
  int done;
};

static int slurp(struct File *f)
{
  FILE *fp = fopen(f->path, "rb");
  long n;
  if (fp == NULL) return 0;
  if (fseek(fp, 0, SEEK_END) != 0 || (n = ftell(fp)) < 0 || fseek(fp, 0, SEEK_SET) != 0) {
    fclose(fp);
    return 0;
  }
  f->bytes = (uint8_t *)malloc((size_t)n + 16);
  f->size = (uint64_t)n;
  if (f->bytes == NULL || fread(f->bytes, 1, (size_t)n, fp) != (size_t)n) {
    fclose(fp);
    return 0;
  }
  fclose(fp);
  return 1;
}

static int write_out(const char *outdir, const char *inpath, const char *ext, const uint8_t *head, size_t head_size,
                     const uint8_t *body, size_t body_size)
{
  char path[4096];
  const char *base = strrchr(inpath, '/');
  const char *dot;
  size_t stem;
  FILE *fp;
  base = base ? base + 1 : inpath;
  dot = strrchr(base, '.');
  stem = dot ? (size_t)(dot - base) : strlen(base);
  if (snprintf(path, sizeof(path), "%s/%.*s%s", outdir, (int)stem, base, ext) >= (int)sizeof(path)) return 0;
  fp = fopen(path, "wb");
  if (fp == NULL) return 0;
  if ((head_size && fwrite(head, 1, head_size, fp) != head_size) || fwrite(body, 1, body_size, fp) != body_size) {
    fclose(fp);
    return 0;
  }
  return fclose(fp) == 0;
}

static int usage(void)
{
  fprintf(stderr, "usage: aad_batch -e [-b bits] [-s max_block_size] [-t trials] [-m 0|1] -o OUTDIR in.wav...\n"
                  "       aad_batch -d -o OUTDIR in.aad...\n");
  return 2;
}

int main(int argc, char **argv)
{
  int mode = 0, i, nfiles = 0, rc = 1, start;
  const char *outdir = NULL;
  struct AADEncodeParameter param;
  struct AADHipContext *ctx = NULL;
  struct File *files;
  const char *dev = getenv("AAD_HIP_DEVICE");

  param.num_channels = 0;
  param.sampling_rate = 0;
  param.bits_per_sample = 4;     /* reference defaults, src/main.c:39-50 */
  param.max_block_size = 1024;
  param.ch_process_method = AAD_CH_PROCESS_METHOD_NONE;
  param.num_encode_trials = 2;

  for (i = 1; i < argc && argv[i][0] == '-'; i++) {
    if (strcmp(argv[i], "-e") == 0) mode = 'e';
    else if (strcmp(argv[i], "-d") == 0) mode = 'd';
    else if (i + 1 < argc && strcmp(argv[i], "-b") == 0) param.bits_per_sample = (uint16_t)atoi(argv[++i]);
    else if (i + 1 < argc && strcmp(argv[i], "-s") == 0) param.max_block_size = (uint16_t)atoi(argv[++i]);
    else if (i + 1 < argc && strcmp(argv[i], "-t") == 0) param.num_encode_trials = (uint8_t)atoi(argv[++i]);
    else if (i + 1 < argc && strcmp(argv[i], "-m") == 0) param.ch_process_method = atoi(argv[++i]) ? AAD_CH_PROCESS_METHOD_MS : AAD_CH_PROCESS_METHOD_NONE;
    else if (i + 1 < argc && strcmp(argv[i], "-o") == 0) outdir = argv[++i];
    else return usage();
  }
  if (mode == 0 || outdir == NULL || i >= argc) return usage();
  start = i;
  nfiles = argc - start;
  files = (struct File *)calloc((size_t)nfiles, sizeof(*files));
  if (files == NULL) return 1;

  for (i = 0; i < nfiles; i++) {
    files[i].path = argv[start + i];
    if (!slurp(&files[i])) {
      fprintf(stderr, "aad_batch: cannot read %s\n", files[i].path);
      goto cleanup;
    }
    if (mode == 'e') {
      AADApiResult r = AADWav_ParseHeader(files[i].bytes, files[i].size, &files[i].wav);
      if (r != AAD_APIRESULT_OK || files[i].wav.format_tag != 1 || files[i].wav.bits_per_sample != 16) {
        fprintf(stderr, "aad_batch: %s is not 16-bit PCM WAV (result %d)\n", files[i].path, (int)r);
        goto cleanup;
      }
    } else {
      AADApiResult r = AADDecoder_DecodeHeader(files[i].bytes, (uint32_t)files[i].size, &files[i].head);
      if (r != AAD_APIRESULT_OK) {
        fprintf(stderr, "aad_batch: %s: bad header (result %d)\n", files[i].path, (int)r);
        goto cleanup;
      }
    }
  }

  if (AADHip_ContextCreate(dev ? atoi(dev) : 0, NULL, &ctx) != AAD_APIRESULT_OK) {
    fprintf(stderr, "aad_batch: no usable HIP device\n");
    goto cleanup;
  }

  /* one engine call per format group: files whose (channels, rate) / header format agree */
  for (;;) {
    int lead = -1, n = 0, k;
    const int16_t **pcm;
    uint32_t *nsamp;
    uint8_t **data;
    uint64_t *cap, *sz;
    int *members;
    AADApiResult r;
    for (i = 0; i < nfiles; i++)
      if (!files[i].done) {
        lead = i;
        break;
      }
    if (lead < 0) break;
    members = (int *)malloc(sizeof(int) * (size_t)nfiles);
    for (i = lead; i < nfiles; i++) {
      int same;
      if (files[i].done) continue;
      if (mode == 'e')
        same = files[i].wav.num_channels == files[lead].wav.num_channels && files[i].wav.sampling_rate == files[lead].wav.sampling_rate;
      else
        same = files[i].head.num_channels == files[lead].head.num_channels && files[i].head.bits_per_sample == files[lead].head.bits_per_sample &&
               files[i].head.block_size == files[lead].head.block_size && files[i].head.num_samples_per_block == files[lead].head.num_samples_per_block &&
               files[i].head.ch_process_method == files[lead].head.ch_process_method;
      if (same) members[n++] = i;
    }
    pcm = (const int16_t **)malloc(sizeof(*pcm) * (size_t)n);
    nsamp = (uint32_t *)malloc(sizeof(*nsamp) * (size_t)n);
    data = (uint8_t **)malloc(sizeof(*data) * (size_t)n);
    cap = (uint64_t *)malloc(sizeof(*cap) * (size_t)n);
    sz = (uint64_t *)malloc(sizeof(*sz) * (size_t)n);
    if (mode == 'e') {
      param.num_channels = files[lead].wav.num_channels;
      param.sampling_rate = files[lead].wav.sampling_rate;
      for (k = 0; k < n; k++) {
        struct File *f = &files[members[k]];
        pcm[k] = (const int16_t *)(f->bytes + f->wav.data_offset); /* the WAV payload is the device layout */
        nsamp[k] = f->wav.num_samples;
        cap[k] = AADHip_CalculateEncodedSize(&param, nsamp[k]);
        f->out = (uint8_t *)malloc((size_t)cap[k] + 16);
        data[k] = f->out;
      }
      r = AADHip_EncodeBatch(ctx, &param, (uint32_t)n, pcm, nsamp, data, cap, sz, NULL);
      if (r != AAD_APIRESULT_OK) {
        fprintf(stderr, "aad_batch: encode failed, API result:%d (%s)\n", (int)r, AADHip_ContextLastError(ctx));
        goto cleanup;
      }
      for (k = 0; k < n; k++) {
        struct File *f = &files[members[k]];
        if (!write_out(outdir, f->path, ".aad", NULL, 0, f->out, (size_t)sz[k])) {
          fprintf(stderr, "aad_batch: cannot write output for %s\n", f->path);
          goto cleanup;
        }
        f->done = 1;
      }
    } else {
      uint32_t *frames = nsamp, *got = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n);
      const uint8_t **img = (const uint8_t **)malloc(sizeof(*img) * (size_t)n);
      int16_t **out = (int16_t **)malloc(sizeof(*out) * (size_t)n);
      for (k = 0; k < n; k++) {
        struct File *f = &files[members[k]];
        img[k] = f->bytes;
        sz[k] = f->size;
        frames[k] = f->head.num_samples;
        f->out = (uint8_t *)calloc((size_t)f->head.num_samples * f->head.num_channels + 8, 2);
        out[k] = (int16_t *)f->out;
      }
      r = AADHip_DecodeBatch(ctx, (uint32_t)n, img, sz, out, frames, got);
      if (r != AAD_APIRESULT_OK) {
        fprintf(stderr, "aad_batch: decode failed, API result:%d (%s)\n", (int)r, AADHip_ContextLastError(ctx));
        goto cleanup;
      }
      for (k = 0; k < n; k++) {
        struct File *f = &files[members[k]];
        uint8_t head[AAD_WAV_HEADER_SIZE];
        AADWav_WriteHeader(head, sizeof(head), f->head.num_channels, f->head.sampling_rate, f->head.num_samples);
        if (!write_out(outdir, f->path, ".wav", head, sizeof(head), f->out,
                       (size_t)f->head.num_samples * f->head.num_channels * 2)) {
          fprintf(stderr, "aad_batch: cannot write output for %s\n", f->path);
          goto cleanup;
        }
        f->done = 1;
      }
      free(got);
      free(img);
      free(out);
    }
    free(pcm);
    free(nsamp);
    free(data);
    free(cap);
    free(sz);
    free(members);
  }
  rc = 0;

cleanup:
  AADHip_ContextDestroy(ctx);
  for (i = 0; i < nfiles; i++) {
    free(files[i].bytes);
    free(files[i].out);
  }
  free(files);
  return rc;
}
