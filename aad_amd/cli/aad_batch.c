/*
 * aad_batch - many-file, many-GPU front end of the MI355X AAD engine (SURVEY.md section 8f rows
 * N1/N2/N3, section 8e partitioning).
 *
 * The reference CLI handles one file per process (src/main.c:518-625) through a bit-serial WAV
 * reader; a GPU needs many independent streams in flight (encode: stream x channel lanes).  This
 * tool keeps the reference's modes, option letters, long names and defaults (src/main.c:20-58:
 * -b 4, -s 1024, -t 2, no M/S) and hands MANY inputs of one format to one AADHip_*Batch call:
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
 * positional ones); -D/--devices 0,1,... (default: $AAD_HIP_DEVICE or 0; a device may be named
 * more than once to run several contexts on it).
 *
 * Partitioning: inputs are sorted by size and dealt longest-first onto the
 * least-loaded device, O(n log n); sizes come from stat(), nothing is read for it.  Each device
 * slot is a pipeline of three threads joined by two-deep queues -
 *     reader : loads the slot's files in waves of <= 512 MB, parses headers, converts 8/24/32-bit
 *              PCM to the codec's int16 by the reference's top-16-bit rule (src/main.c:175-179)
 *     device : one AADHip_*Batch call per format group of a wave (context, stream and pinned
 *              staging of its own; no traffic between devices)
 *     writer : writes the wave's outputs and releases its memory
 * - so file reads, device work and file writes of consecutive waves overlap.  Waves are large on
 * purpose: an encoder launch takes about `blocks of the longest file` x 64 us however many files it
 * holds (the blocks of a file are chained), so the more long files share a wave the better; the
 * library cuts the wave into tiles that fit its pinned staging blocks by itself.
 *
 * Output names are OUTDIR/<stem><ext>; two inputs with the same stem would overwrite each other,
 * so that is refused up front.  Every output is byte-identical to what the reference CLI writes
 * for the same input.  Host C only; all codec work happens in libaad_hip.so.
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include "../../include/aad_hip.h"
#include "../../include/aad_wav.h"

#define MAX_DEVICES 16
#define WAVE_BYTES_DEFAULT (512ull << 20) /* input bytes per wave of one device slot ($AAD_BATCH_WAVE_BYTES overrides: tests) */
#define WAVE_FILES 8192
#define QUEUE_DEPTH 2

struct File {
  const char *path;
  uint64_t disk_size;
  uint8_t *bytes;             /* file image */
  uint64_t size;
  int16_t *converted;         /* int16 PCM made from 8/24/32-bit input, else NULL */
  struct AADWavInfo wav;      /* WAV-input modes */
  struct AADHeaderInfo head;  /* decode */
  uint8_t *out;
  uint64_t out_size;
  struct AADHipErrorStats stats;
  int device_slot;
  int done;
};

struct Options {
  int mode; /* 'e' 'd' 'r' 'g' 'c' */
  uint64_t wave_bytes;
  const char *outdir;
  struct AADEncodeParameter param;
};

/* a wave: files [first, last) of a slot's list */
struct Wave {
  int first, last;
  int failed; /* set by the stage that could not do its part; later stages only release */
  int end;    /* sentinel: no more waves */
};

struct Queue {
  pthread_mutex_t lock;
  pthread_cond_t changed;
  struct Wave item[QUEUE_DEPTH];
  int count, head;
};

struct Slot {
  pthread_t reader, device_thread, writer;
  int slot, device, failed;
  const struct Options *opt;
  struct File **files; /* this slot's files, longest first */
  int nfiles;
  struct Queue loaded, computed;
};

static void queue_init(struct Queue *q)
{
  pthread_mutex_init(&q->lock, NULL);
  pthread_cond_init(&q->changed, NULL);
  q->count = q->head = 0;
}

static void queue_push(struct Queue *q, struct Wave w)
{
  pthread_mutex_lock(&q->lock);
  while (q->count == QUEUE_DEPTH) pthread_cond_wait(&q->changed, &q->lock);
  q->item[(q->head + q->count) % QUEUE_DEPTH] = w;
  q->count++;
  pthread_cond_broadcast(&q->changed);
  pthread_mutex_unlock(&q->lock);
}

static struct Wave queue_pop(struct Queue *q)
{
  struct Wave w;
  pthread_mutex_lock(&q->lock);
  while (q->count == 0) pthread_cond_wait(&q->changed, &q->lock);
  w = q->item[q->head];
  q->head = (q->head + 1) % QUEUE_DEPTH;
  q->count--;
  pthread_cond_broadcast(&q->changed);
  pthread_mutex_unlock(&q->lock);
  return w;
}

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

static void release_file(struct File *f)
{
  free(f->bytes);
  free(f->converted);
  free(f->out);
  f->bytes = f->out = NULL;
  f->converted = NULL;
}

/* "<stem>" of a path: the file name without directory and without its last extension */
static void stem_of(const char *path, const char **start, size_t *len)
{
  const char *base = strrchr(path, '/'), *dot;
  base = base ? base + 1 : path;
  dot = strrchr(base, '.');
  *start = base;
  *len = dot && dot != base ? (size_t)(dot - base) : strlen(base);
}

static int write_out(const char *outdir, const char *inpath, const char *ext, const uint8_t *head, size_t head_size,
                     const uint8_t *body, size_t body_size)
{
  char path[4096];
  const char *base;
  size_t stem;
  FILE *fp;
  stem_of(inpath, &base, &stem);
  if (snprintf(path, sizeof(path), "%s/%.*s%s", outdir, (int)stem, base, ext) >= (int)sizeof(path)) return 0;
  fp = fopen(path, "wb");
  if (fp == NULL) return 0;
  if ((head_size && fwrite(head, 1, head_size, fp) != head_size) || fwrite(body, 1, body_size, fp) != body_size) {
    fclose(fp);
    return 0;
  }
  return fclose(fp) == 0;
}

static int same_format(int mode, const struct File *a, const struct File *b)
{
  if (mode != 'd') return a->wav.num_channels == b->wav.num_channels && a->wav.sampling_rate == b->wav.sampling_rate;
  return a->head.num_channels == b->head.num_channels && a->head.bits_per_sample == b->head.bits_per_sample &&
         a->head.block_size == b->head.block_size && a->head.num_samples_per_block == b->head.num_samples_per_block &&
         a->head.ch_process_method == b->head.ch_process_method;
}

/* ---- reader stage ------------------------------------------------------------------------ */

static int load_file(const struct Options *opt, struct File *f)
{
  AADApiResult r;
  if (!slurp(f)) {
    fprintf(stderr, "aad_batch: cannot read %s\n", f->path);
    return 0;
  }
  if (opt->mode == 'd') {
    r = AADDecoder_DecodeHeader(f->bytes, (uint32_t)(f->size > 0xFFFFFFFFu ? 0xFFFFFFFFu : f->size), &f->head);
    if (r != AAD_APIRESULT_OK) {
      fprintf(stderr, "aad_batch: %s: bad header (result %d)\n", f->path, (int)r);
      return 0;
    }
    return 1;
  }
  r = AADWav_ParseHeader(f->bytes, f->size, &f->wav);
  if (r != AAD_APIRESULT_OK || f->wav.format_tag != 1) {
    fprintf(stderr, "aad_batch: %s is not a PCM WAV file (result %d)\n", f->path, (int)r);
    return 0;
  }
  if (f->wav.bits_per_sample != 16) { /* 8 / 24 / 32-bit PCM: the codec sees the top 16 bits (src/main.c:175-179) */
    const uint64_t count = (uint64_t)f->wav.num_samples * f->wav.num_channels;
    f->converted = (int16_t *)malloc(sizeof(int16_t) * (size_t)(count + 8));
    if (f->converted == NULL ||
        AADWav_ConvertToPcm16(f->bytes + f->wav.data_offset, f->wav.bits_per_sample, count, f->converted) != AAD_APIRESULT_OK) {
      fprintf(stderr, "aad_batch: %s: unsupported PCM width %u\n", f->path, (unsigned)f->wav.bits_per_sample);
      return 0;
    }
    free(f->bytes); /* only the converted samples are needed from here on */
    f->bytes = NULL;
  }
  return 1;
}

static void *reader_main(void *arg)
{
  struct Slot *s = (struct Slot *)arg;
  struct Wave w;
  int i = 0;
  memset(&w, 0, sizeof(w));
  while (i < s->nfiles && !w.failed) {
    uint64_t bytes = 0;
    w.first = i;
    while (i < s->nfiles && (i == w.first || (bytes + s->files[i]->disk_size <= s->opt->wave_bytes && i - w.first < WAVE_FILES))) {
      bytes += s->files[i]->disk_size;
      if (!load_file(s->opt, s->files[i])) w.failed = 1;
      i++;
    }
    w.last = i;
    queue_push(&s->loaded, w);
  }
  memset(&w, 0, sizeof(w));
  w.end = 1;
  queue_push(&s->loaded, w);
  return NULL;
}

/* ---- device stage ------------------------------------------------------------------------ */

/* one engine call for `n` same-format files; returns 0 on failure */
static int run_group(struct Slot *s, struct AADHipContext *ctx, struct File **g, int n)
{
  const struct Options *opt = s->opt;
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
      g[k]->out_size = (uint64_t)frames[k] * g[k]->head.num_channels * 2;
      out[k] = g[k]->out = (uint8_t *)calloc((size_t)frames[k] * g[k]->head.num_channels + 8, 2);
      if (out[k] == NULL) goto done;
    }
    r = AADHip_DecodeBatch(ctx, (uint32_t)n, (const uint8_t *const *)in, sizes, (int16_t *const *)out, frames, NULL);
  } else {
    param.num_channels = g[0]->wav.num_channels;
    param.sampling_rate = g[0]->wav.sampling_rate;
    for (k = 0; k < n; k++) {
      /* a 16-bit WAV payload IS the device layout; other widths were converted by the reader */
      in[k] = g[k]->converted ? (const void *)g[k]->converted : (const void *)(g[k]->bytes + g[k]->wav.data_offset);
      frames[k] = g[k]->wav.num_samples;
      sizes[k] = mode == 'e' ? AADHip_CalculateEncodedSize(&param, frames[k]) : (uint64_t)frames[k] * param.num_channels * 2;
      out[k] = NULL;
      if (mode != 'c') {
        out[k] = g[k]->out = (uint8_t *)malloc((size_t)sizes[k] + 16);
        g[k]->out_size = sizes[k];
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
    fprintf(stderr, "aad_batch: device %d: failed, API result:%d (%s)\n", s->device, (int)r, AADHip_ContextLastError(ctx));
    goto done;
  }
  for (k = 0; k < n; k++) {
    if (mode == 'e') g[k]->out_size = got[k];
    if (mode == 'c') g[k]->stats = stats[k];
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

static void *device_main(void *arg)
{
  struct Slot *s = (struct Slot *)arg;
  struct AADHipContext *ctx = NULL;
  struct File **group = (struct File **)malloc(sizeof(*group) * (size_t)(s->nfiles ? s->nfiles : 1));
  int usable = group != NULL && AADHip_ContextCreate(s->device, NULL, &ctx) == AAD_APIRESULT_OK;
  if (!usable) fprintf(stderr, "aad_batch: HIP device %d is not usable\n", s->device);
  for (;;) {
    struct Wave w = queue_pop(&s->loaded);
    if (w.end) {
      queue_push(&s->computed, w);
      break;
    }
    if (!usable) w.failed = 1;
    while (!w.failed) { /* one engine call per format group of the wave */
      int n = 0, i;
      struct File *lead = NULL;
      for (i = w.first; i < w.last; i++) {
        struct File *f = s->files[i];
        if (f->done) continue;
        if (lead == NULL) lead = f;
        if (same_format(s->opt->mode, lead, f)) group[n++] = f;
      }
      if (n == 0) break;
      if (!run_group(s, ctx, group, n)) w.failed = 1;
    }
    queue_push(&s->computed, w);
  }
  AADHip_ContextDestroy(ctx);
  free(group);
  return NULL;
}

/* ---- writer stage ------------------------------------------------------------------------ */

static void *writer_main(void *arg)
{
  struct Slot *s = (struct Slot *)arg;
  const struct Options *opt = s->opt;
  int failed = 0;
  for (;;) {
    struct Wave w = queue_pop(&s->computed);
    int i;
    if (w.end) break;
    failed |= w.failed;
    for (i = w.first; i < w.last; i++) {
      struct File *f = s->files[i];
      int wrote = 1;
      if (!failed && f->done && opt->mode != 'c') {
        if (opt->mode == 'e') {
          wrote = write_out(opt->outdir, f->path, ".aad", NULL, 0, f->out, (size_t)f->out_size);
        } else {
          uint8_t head[AAD_WAV_HEADER_SIZE];
          const int d = opt->mode == 'd';
          const uint16_t ch = d ? f->head.num_channels : f->wav.num_channels;
          const uint32_t frames = d ? f->head.num_samples : f->wav.num_samples;
          AADWav_WriteHeader(head, sizeof(head), ch, d ? f->head.sampling_rate : f->wav.sampling_rate, frames);
          wrote = write_out(opt->outdir, f->path, ".wav", head, sizeof(head), f->out, (size_t)frames * ch * 2);
        }
        if (!wrote) {
          fprintf(stderr, "aad_batch: cannot write output for %s\n", f->path);
          failed = 1;
        }
      }
      release_file(f);
    }
  }
  s->failed = failed;
  return NULL;
}

/* ---- main ---------------------------------------------------------------------------------- */

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

static int by_size_desc(const void *a, const void *b)
{
  const struct File *x = *(const struct File *const *)a, *y = *(const struct File *const *)b;
  if (x->disk_size != y->disk_size) return x->disk_size < y->disk_size ? 1 : -1;
  return x < y ? -1 : x > y; /* ties: input order */
}

static int by_stem(const void *a, const void *b)
{
  const char *sa, *sb;
  size_t la, lb;
  int c;
  stem_of((*(const struct File *const *)a)->path, &sa, &la);
  stem_of((*(const struct File *const *)b)->path, &sb, &lb);
  c = strncmp(sa, sb, la < lb ? la : lb);
  return c != 0 ? c : (la < lb ? -1 : la > lb);
}

int main(int argc, char **argv)
{
  struct Options opt;
  struct Slot slots[MAX_DEVICES];
  uint64_t load[MAX_DEVICES];
  int devices[MAX_DEVICES], ndev = 0, counts[MAX_DEVICES];
  const char *list = NULL, *devarg = getenv("AAD_HIP_DEVICE");
  char **paths = NULL, *listbuf = NULL;
  struct File *files = NULL, **order = NULL;
  int i, k, npaths = 0, cap, rc = 1, started = 0;

  memset(&opt, 0, sizeof(opt));
  memset(slots, 0, sizeof(slots));
  opt.param.bits_per_sample = 4; /* reference defaults, src/main.c:39-50 */
  opt.param.max_block_size = 1024;
  opt.param.ch_process_method = AAD_CH_PROCESS_METHOD_NONE;
  opt.param.num_encode_trials = 2;
  opt.wave_bytes = WAVE_BYTES_DEFAULT;
  if (getenv("AAD_BATCH_WAVE_BYTES") != NULL) {
    const long long v = atoll(getenv("AAD_BATCH_WAVE_BYTES"));
    if (v > 0) opt.wave_bytes = (uint64_t)v;
  }

  cap = argc + 1;
  paths = (char **)malloc(sizeof(*paths) * (size_t)cap);
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
    else if (a[0] == '-' && a[1] != 0) goto bad_usage;
    else paths[npaths++] = argv[i];
  }
  if (opt.mode == 0 || (opt.mode != 'c' && opt.outdir == NULL)) goto bad_usage;

  if (list != NULL) { /* one path per line; LF, CRLF or bare CR endings */
    struct File lf;
    char *p;
    memset(&lf, 0, sizeof(lf));
    lf.path = list;
    if (!slurp(&lf)) {
      fprintf(stderr, "aad_batch: cannot read list %s\n", list);
      free(lf.bytes);
      goto cleanup;
    }
    listbuf = (char *)lf.bytes;
    listbuf[lf.size] = 0;
    for (p = strtok(listbuf, "\r\n"); p != NULL; p = strtok(NULL, "\r\n")) {
      if (*p == 0) continue;
      if (npaths == cap) { /* the array grows with the tokens actually found */
        char **grown = (char **)realloc(paths, sizeof(*paths) * (size_t)cap * 2);
        if (grown == NULL) goto cleanup;
        paths = grown;
        cap *= 2;
      }
      paths[npaths++] = p;
    }
  }
  if (npaths == 0) goto bad_usage;

  for (ndev = 0; devarg != NULL && *devarg && ndev < MAX_DEVICES;) {
    char *end;
    const long d = strtol(devarg, &end, 10);
    if (end == devarg || d < 0) goto bad_usage;
    devices[ndev++] = (int)d;
    devarg = *end == ',' ? end + 1 : end;
    if (*end != ',' && *end != 0) goto bad_usage;
  }
  if (ndev == 0) devices[ndev++] = 0;

  files = (struct File *)calloc((size_t)npaths, sizeof(*files));
  order = (struct File **)malloc(sizeof(*order) * (size_t)npaths);
  if (files == NULL || order == NULL) goto cleanup;
  for (i = 0; i < npaths; i++) {
    struct stat st;
    files[i].path = paths[i];
    if (stat(paths[i], &st) != 0 || !S_ISREG(st.st_mode)) {
      fprintf(stderr, "aad_batch: cannot read %s\n", paths[i]);
      goto cleanup;
    }
    files[i].disk_size = (uint64_t)st.st_size;
    order[i] = &files[i];
  }

  /* OUTDIR/<stem><ext> must be unique: a/x.wav and b/x.wav would silently overwrite each other */
  if (opt.mode != 'c') {
    qsort(order, (size_t)npaths, sizeof(*order), by_stem);
    for (i = 1; i < npaths; i++) {
      if (by_stem(&order[i - 1], &order[i]) == 0) {
        fprintf(stderr, "aad_batch: %s and %s would write the same output file\n", order[i - 1]->path, order[i]->path);
        goto cleanup;
      }
    }
  }

  /* longest-processing-time-first onto the least-loaded device: sort once,
   * then one pass; work is taken as the file size */
  qsort(order, (size_t)npaths, sizeof(*order), by_size_desc);
  memset(load, 0, sizeof(load));
  memset(counts, 0, sizeof(counts));
  for (k = 0; k < npaths; k++) {
    int slot = 0;
    for (i = 1; i < ndev; i++)
      if (load[i] < load[slot]) slot = i;
    order[k]->device_slot = slot;
    load[slot] += order[k]->disk_size + 1;
    counts[slot]++;
  }

  for (i = 0; i < ndev; i++) {
    struct Slot *s = &slots[i];
    int n = 0;
    s->slot = i;
    s->device = devices[i];
    s->failed = 1;
    s->opt = &opt;
    s->files = (struct File **)malloc(sizeof(*s->files) * (size_t)(counts[i] ? counts[i] : 1));
    if (s->files == NULL) goto cleanup;
    for (k = 0; k < npaths; k++)
      if (order[k]->device_slot == i) s->files[n++] = order[k];
    s->nfiles = n;
    queue_init(&s->loaded);
    queue_init(&s->computed);
  }
  for (i = 0; i < ndev; i++) {
    struct Slot *s = &slots[i];
    if (pthread_create(&s->reader, NULL, reader_main, s) != 0 || pthread_create(&s->device_thread, NULL, device_main, s) != 0 ||
        pthread_create(&s->writer, NULL, writer_main, s) != 0) {
      fprintf(stderr, "aad_batch: cannot start the threads of device slot %d\n", i);
      exit(1); /* a half-started pipeline cannot be drained */
    }
    started++;
  }
  rc = 0;
  for (i = 0; i < started; i++) {
    pthread_join(slots[i].reader, NULL);
    pthread_join(slots[i].device_thread, NULL);
    pthread_join(slots[i].writer, NULL);
    rc |= slots[i].failed;
  }
  if (rc == 0 && opt.mode == 'c')
    for (i = 0; i < npaths; i++) /* the reference's line (src/main.c:493-497) behind the path, in input order */
      printf("%s\tRMSE:%f MSD:%f MaxAE:%f \n", files[i].path, files[i].stats.rms_error, files[i].stats.mean_abs_error,
             files[i].stats.max_abs_error);
  goto cleanup;

bad_usage:
  rc = usage();
cleanup:
  for (i = 0; files != NULL && i < npaths; i++) release_file(&files[i]);
  for (i = 0; i < MAX_DEVICES; i++) free(slots[i].files);
  free(files);
  free(order);
  free(listbuf);
  free(paths);
  return rc;
}
