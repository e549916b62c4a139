/*
 * aad_batch - many-file, many-GPU front end of the MI355X AAD engine (SURVEY.md section 8f rows
 * N1/N2/N3, section 8e partitioning).
 *
 * The reference CLI handles one file per process (src/main.c:518-625) through a bit-serial WAV
 * reader; a GPU needs many independent streams in flight (encode: stream x channel lanes).  This
 * tool keeps the reference's modes, option letters, long names and defaults (src/main.c:20-58:
 * -b 4, -s 1024, -t 2, no M/S) and hands ALL inputs of one format to one AADHip_*Batch call:
 *
 *   aad_batch -e|-r|-g [-b bits] [-s max_block_size] [-t trials] [-m] -o OUTDIR in.wav ...
 *   aad_batch -c       [-b bits] [-s max_block_size] [-t trials] [-m] in.wav ...
 *   aad_batch -d -o OUTDIR in.aad ...
 *
 *   -e encode (.wav -> OUTDIR/<stem>.aad)           -d decode (.aad -> OUTDIR/<stem>.wav)
 *   -r reconstruct (.wav -> encode -> decode -> OUTDIR/<stem>.wav)
 *   -g gap / residual (.wav -> original minus reconstruction -> OUTDIR/<stem>.wav)
 *   -c calculate: one line per input, "<path>\t" followed by exactly what `aad -c` prints
 *
 * Batch additions: -o/--output-dir DIR; -l/--list FILE (one input path per line, added to the
 * positional ones); -D/--devices 0,1,... (default: $AAD_HIP_DEVICE or 0).  With several devices
 * the inputs are dealt longest-first onto the least-loaded device (the static partition of
 * SURVEY.md section 8e); each device gets its own host thread, context, stream and pinned
 * staging, and there is no traffic between them.
 *
 * Every output is byte-identical to what the reference CLI writes for the same input (16-bit
 * PCM WAV; the payload is used as the device PCM layout without conversion).  Host C only; all
 * codec work happens in libaad_hip.so.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/aad_hip.h"
#include "../../include/aad_wav.h"

#define MAX_DEVICES 16

struct File {
  const char *path;
  uint8_t *bytes;
  uint64_t size;
  struct AADWavInfo wav;      /* WAV-input modes */
  struct AADHeaderInfo head;  /* decode */
  uint8_t *out;
  struct AADHipErrorStats stats;
  int device_slot;
  int done;
};

struct Options {
  int mode; /* 'e' 'd' 'r' 'g' 'c' */
  const char *outdir;
  struct AADEncodeParameter param;
};

struct Worker {
  pthread_t thread;
  int slot, device, failed;
  const struct Options *opt;
  struct File *files;
  int nfiles;
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

static int write_wav(const char *outdir, const struct File *f, uint16_t channels, uint32_t rate, uint32_t frames)
{
  uint8_t head[AAD_WAV_HEADER_SIZE];
  AADWav_WriteHeader(head, sizeof(head), channels, rate, frames);
  return write_out(outdir, f->path, ".wav", head, sizeof(head), f->out, (size_t)frames * channels * 2);
}

static int same_format(int mode, const struct File *a, const struct File *b)
{
  if (mode != 'd') return a->wav.num_channels == b->wav.num_channels && a->wav.sampling_rate == b->wav.sampling_rate;
  return a->head.num_channels == b->head.num_channels && a->head.bits_per_sample == b->head.bits_per_sample &&
         a->head.block_size == b->head.block_size && a->head.num_samples_per_block == b->head.num_samples_per_block &&
         a->head.ch_process_method == b->head.ch_process_method;
}

/* one engine call for `n` same-format files of this worker; returns 0 on failure */
static int run_group(struct Worker *w, struct AADHipContext *ctx, struct File **g, int n)
{
  const struct Options *opt = w->opt;
  const int mode = opt->mode;
  struct AADEncodeParameter param = opt->param;
  const void **in = (const void **)malloc(sizeof(*in) * (size_t)n);
  void **out = (void **)malloc(sizeof(*out) * (size_t)n);
  uint32_t *frames = (uint32_t *)malloc(sizeof(*frames) * (size_t)n);
  uint64_t *sizes = (uint64_t *)malloc(sizeof(*sizes) * (size_t)n);
  uint64_t *got = (uint64_t *)malloc(sizeof(*got) * (size_t)n);
  struct AADHipErrorStats *stats = (struct AADHipErrorStats *)malloc(sizeof(*stats) * (size_t)n);
  AADApiResult r = AAD_APIRESULT_NG;
  int k, ok = 0;
  if (!in || !out || !frames || !sizes || !got || !stats) goto done;

  if (mode == 'd') {
    for (k = 0; k < n; k++) {
      in[k] = g[k]->bytes;
      sizes[k] = g[k]->size;
      frames[k] = g[k]->head.num_samples;
      out[k] = g[k]->out = (uint8_t *)calloc((size_t)frames[k] * g[k]->head.num_channels + 8, 2);
      if (out[k] == NULL) goto done;
    }
    r = AADHip_DecodeBatch(ctx, (uint32_t)n, (const uint8_t *const *)in, sizes, (int16_t *const *)out, frames, NULL);
  } else {
    param.num_channels = g[0]->wav.num_channels;
    param.sampling_rate = g[0]->wav.sampling_rate;
    for (k = 0; k < n; k++) {
      in[k] = g[k]->bytes + g[k]->wav.data_offset; /* the WAV payload is the device layout */
      frames[k] = g[k]->wav.num_samples;
      sizes[k] = mode == 'e' ? AADHip_CalculateEncodedSize(&param, frames[k]) : (uint64_t)frames[k] * param.num_channels * 2;
      out[k] = NULL;
      if (mode != 'c') {
        out[k] = g[k]->out = (uint8_t *)malloc((size_t)sizes[k] + 16);
        if (out[k] == NULL) goto done;
      }
    }
    if (mode == 'e')
      r = AADHip_EncodeBatch(ctx, &param, (uint32_t)n, (const int16_t *const *)in, frames, (uint8_t *const *)out, sizes, got, NULL);
    else
      r = AADHip_ReconstructBatch(ctx, &param, (uint32_t)n, (const int16_t *const *)in, frames,
                                  mode == 'g' ? AAD_HIP_RECONSTRUCT_RESIDUAL : AAD_HIP_RECONSTRUCT_DECODED,
                                  mode == 'c' ? NULL : (int16_t *const *)out, mode == 'c' ? stats : NULL);
  }
  if (r != AAD_APIRESULT_OK) {
    fprintf(stderr, "aad_batch: device %d: failed, API result:%d (%s)\n", w->device, (int)r, AADHip_ContextLastError(ctx));
    goto done;
  }
  for (k = 0; k < n; k++) {
    int wrote = 1;
    if (mode == 'e') wrote = write_out(opt->outdir, g[k]->path, ".aad", NULL, 0, g[k]->out, (size_t)got[k]);
    else if (mode == 'd') wrote = write_wav(opt->outdir, g[k], g[k]->head.num_channels, g[k]->head.sampling_rate, g[k]->head.num_samples);
    else if (mode == 'c') g[k]->stats = stats[k];
    else wrote = write_wav(opt->outdir, g[k], g[k]->wav.num_channels, g[k]->wav.sampling_rate, g[k]->wav.num_samples);
    if (!wrote) {
      fprintf(stderr, "aad_batch: cannot write output for %s\n", g[k]->path);
      goto done;
    }
    free(g[k]->out);
    g[k]->out = NULL;
    g[k]->done = 1;
  }
  ok = 1;
done:
  free(in);
  free(out);
  free(frames);
  free(sizes);
  free(got);
  free(stats);
  return ok;
}

static void *worker_main(void *arg)
{
  struct Worker *w = (struct Worker *)arg;
  struct AADHipContext *ctx = NULL;
  struct File **group = (struct File **)malloc(sizeof(*group) * (size_t)(w->nfiles ? w->nfiles : 1));
  int i;
  w->failed = 1;
  if (group == NULL) return NULL;
  if (AADHip_ContextCreate(w->device, NULL, &ctx) != AAD_APIRESULT_OK) {
    fprintf(stderr, "aad_batch: HIP device %d is not usable\n", w->device);
    free(group);
    return NULL;
  }
  for (;;) { /* one engine call per format group among this worker's files */
    int n = 0;
    struct File *lead = NULL;
    for (i = 0; i < w->nfiles; i++) {
      struct File *f = &w->files[i];
      if (f->device_slot != w->slot || f->done) continue;
      if (lead == NULL) lead = f;
      if (same_format(w->opt->mode, lead, f)) group[n++] = f;
    }
    if (n == 0) {
      w->failed = 0;
      break;
    }
    if (!run_group(w, ctx, group, n)) break;
  }
  AADHip_ContextDestroy(ctx);
  free(group);
  return NULL;
}

static int usage(void)
{
  fprintf(stderr, "usage: aad_batch -e|-r|-g [-b bits] [-s max_block_size] [-t trials] [-m] [-D dev,dev,...] -o OUTDIR [-l LIST] in.wav...\n"
                  "       aad_batch -c       [-b bits] [-s max_block_size] [-t trials] [-m] [-D dev,dev,...] [-l LIST] in.wav...\n"
                  "       aad_batch -d [-D dev,dev,...] -o OUTDIR [-l LIST] in.aad...\n");
  return 2;
}

static int is_opt(const char *arg, const char *shortname, const char *longname)
{
  return strcmp(arg, shortname) == 0 || strcmp(arg, longname) == 0;
}

int main(int argc, char **argv)
{
  struct Options opt;
  struct Worker workers[MAX_DEVICES];
  uint64_t load[MAX_DEVICES];
  int devices[MAX_DEVICES], ndev = 0;
  const char *list = NULL, *devarg = getenv("AAD_HIP_DEVICE");
  char **paths = NULL, *listbuf = NULL;
  struct File *files = NULL;
  int i, k, npaths = 0, rc = 1;

  memset(&opt, 0, sizeof(opt));
  opt.param.bits_per_sample = 4; /* reference defaults, src/main.c:39-50 */
  opt.param.max_block_size = 1024;
  opt.param.ch_process_method = AAD_CH_PROCESS_METHOD_NONE;
  opt.param.num_encode_trials = 2;

  paths = (char **)malloc(sizeof(*paths) * (size_t)(argc + 1));
  if (paths == NULL) return 1;
  for (i = 1; i < argc; i++) {
    const char *a = argv[i];
    const int has_value = i + 1 < argc;
    if (is_opt(a, "-e", "--encode")) opt.mode = 'e';
    else if (is_opt(a, "-d", "--decode")) opt.mode = 'd';
    else if (is_opt(a, "-r", "--reconstruct")) opt.mode = 'r';
    else if (is_opt(a, "-g", "--gap")) opt.mode = 'g';
    else if (is_opt(a, "-c", "--calculate")) opt.mode = 'c';
    else if (is_opt(a, "-m", "--ms-conversion")) opt.param.ch_process_method = AAD_CH_PROCESS_METHOD_MS;
    else if (has_value && is_opt(a, "-b", "--bits-per-sample")) opt.param.bits_per_sample = (uint16_t)strtol(argv[++i], NULL, 10);
    else if (has_value && is_opt(a, "-s", "--max-block-size")) opt.param.max_block_size = (uint16_t)strtol(argv[++i], NULL, 10);
    else if (has_value && is_opt(a, "-t", "--num-encode-trials")) opt.param.num_encode_trials = (uint8_t)strtol(argv[++i], NULL, 10);
    else if (has_value && is_opt(a, "-o", "--output-dir")) opt.outdir = argv[++i];
    else if (has_value && is_opt(a, "-l", "--list")) list = argv[++i];
    else if (has_value && is_opt(a, "-D", "--devices")) devarg = argv[++i];
    else if (a[0] == '-' && a[1] != 0) return usage();
    else paths[npaths++] = argv[i];
  }
  if (opt.mode == 0 || (opt.mode != 'c' && opt.outdir == NULL)) return usage();

  if (list != NULL) { /* one path per line */
    struct File lf;
    char *p, **grown;
    int lines = 0;
    memset(&lf, 0, sizeof(lf));
    lf.path = list;
    if (!slurp(&lf)) {
      fprintf(stderr, "aad_batch: cannot read list %s\n", list);
      goto cleanup;
    }
    listbuf = (char *)lf.bytes;
    listbuf[lf.size] = 0;
    for (p = listbuf; *p; p++) lines += *p == '\n';
    grown = (char **)realloc(paths, sizeof(*paths) * (size_t)(npaths + lines + 2));
    if (grown == NULL) goto cleanup;
    paths = grown;
    for (p = strtok(listbuf, "\r\n"); p != NULL; p = strtok(NULL, "\r\n"))
      if (*p) paths[npaths++] = p;
  }
  if (npaths == 0) return usage();

  for (ndev = 0; devarg != NULL && *devarg && ndev < MAX_DEVICES;) {
    char *end;
    const long d = strtol(devarg, &end, 10);
    if (end == devarg || d < 0) return usage();
    devices[ndev++] = (int)d;
    devarg = *end == ',' ? end + 1 : end;
    if (*end != ',' && *end != 0) return usage();
  }
  if (ndev == 0) devices[ndev++] = 0;

  files = (struct File *)calloc((size_t)npaths, sizeof(*files));
  if (files == NULL) goto cleanup;
  for (i = 0; i < npaths; i++) {
    AADApiResult r;
    files[i].path = paths[i];
    if (!slurp(&files[i])) {
      fprintf(stderr, "aad_batch: cannot read %s\n", files[i].path);
      goto cleanup;
    }
    if (opt.mode != 'd') {
      r = AADWav_ParseHeader(files[i].bytes, files[i].size, &files[i].wav);
      if (r != AAD_APIRESULT_OK || files[i].wav.format_tag != 1 || files[i].wav.bits_per_sample != 16) {
        fprintf(stderr, "aad_batch: %s is not 16-bit PCM WAV (result %d)\n", files[i].path, (int)r);
        goto cleanup;
      }
    } else {
      r = AADDecoder_DecodeHeader(files[i].bytes, (uint32_t)files[i].size, &files[i].head);
      if (r != AAD_APIRESULT_OK) {
        fprintf(stderr, "aad_batch: %s: bad header (result %d)\n", files[i].path, (int)r);
        goto cleanup;
      }
    }
  }

  /* longest-processing-time-first onto the least-loaded device */
  memset(load, 0, sizeof(load));
  for (i = 0; i < npaths; i++) files[i].device_slot = -1;
  for (k = 0; k < npaths; k++) {
    int best = -1, slot = 0;
    uint64_t best_work = 0;
    for (i = 0; i < npaths; i++) {
      const uint64_t work = opt.mode != 'd' ? (uint64_t)files[i].wav.num_samples * files[i].wav.num_channels
                                            : (uint64_t)files[i].head.num_samples * files[i].head.num_channels;
      if (files[i].device_slot < 0 && (best < 0 || work > best_work)) {
        best = i;
        best_work = work;
      }
    }
    for (i = 1; i < ndev; i++)
      if (load[i] < load[slot]) slot = i;
    files[best].device_slot = slot;
    load[slot] += best_work + 1;
  }

  for (i = 0; i < ndev; i++) {
    workers[i].slot = i;
    workers[i].device = devices[i];
    workers[i].failed = 1;
    workers[i].opt = &opt;
    workers[i].files = files;
    workers[i].nfiles = npaths;
  }
  if (ndev == 1) {
    worker_main(&workers[0]);
  } else {
    for (i = 0; i < ndev; i++)
      if (pthread_create(&workers[i].thread, NULL, worker_main, &workers[i]) != 0) {
        fprintf(stderr, "aad_batch: cannot start a thread for device %d\n", devices[i]);
        for (k = 0; k < i; k++) pthread_join(workers[k].thread, NULL);
        goto cleanup;
      }
    for (i = 0; i < ndev; i++) pthread_join(workers[i].thread, NULL);
  }
  rc = 0;
  for (i = 0; i < ndev; i++) rc |= workers[i].failed;
  if (rc == 0 && opt.mode == 'c')
    for (i = 0; i < npaths; i++) /* the reference's line (src/main.c:493-497) behind the path */
      printf("%s\tRMSE:%f MSD:%f MaxAE:%f \n", files[i].path, files[i].stats.rms_error, files[i].stats.mean_abs_error,
             files[i].stats.max_abs_error);

cleanup:
  for (i = 0; files != NULL && i < npaths; i++) {
    free(files[i].bytes);
    free(files[i].out);
  }
  free(files);
  free(listbuf);
  free(paths);
  return rc;
}
