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
 *     reader : maps the slot's files (read-only, no copy) in waves, parses headers, converts
 *              8/24/32-bit PCM to the codec's int16 by the reference's top-16-bit rule (src/main.c:175-179)
 *     device : one AADHip_*Batch call per format group of a wave (context, stream and pinned
 *              staging of its own; no traffic between devices).  Every output file is created at
 *              its final size and mapped, so the library's staging threads copy results straight
 *              into the page cache (a buffer plus a write() pass by one thread was the slowest
 *              stage of -d, whose output is four times its input)
 *     writer : closes the wave's outputs (trims or removes them after a failure) and unmaps its inputs
 * - so header parsing, device work and file completion of consecutive waves overlap.  A wave is
 * up to 8192 files or 64 GB (mapped files cost page cache, not process memory): an encoder launch
 * takes about `blocks of the longest file` x 64 us (x3 with the default trial search) however many
 * files it holds - the blocks of a file are chained - so the more long files share a wave the
 * better; for -e and -d the library cuts the wave into tiles that fit its pinned staging blocks by
 * itself; the reconstruction modes (-r / -g / -c) keep a whole wave on the device (encode, decode and
 * the statistics of all its files in one run) and stage it through the same pinned blocks in chunks.
 *
 * Output names are OUTDIR/<stem><ext>; two inputs with the same stem would overwrite each other,
 * so that is refused up front.  Every output is byte-identical to what the reference CLI writes
 * for the same input.  Host C only; all codec work happens in libaad_hip.so.
 */
#define _POSIX_C_SOURCE 200809L
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/aad_hip.h"
#include "../../include/aad_wav.h"

#define MAX_DEVICES 16
#define WAVE_BYTES_DEFAULT (64ull << 30) /* bytes a wave of one device slot spans, max(inputs, outputs) ($AAD_BATCH_WAVE_BYTES overrides: tests) */
#define WAVE_FILES 8192
#define QUEUE_DEPTH 2

struct File {
  const char *path;
  uint64_t disk_size;
  uint8_t *bytes;             /* file image: the file itself, mapped read-only (no copy: the library stages straight from the page cache) */
  uint64_t size;
  int mapped;                 /* bytes came from mmap */
  uint64_t in_dev, in_ino;    /* the input file's identity: an output that would land on it is not mapped over it */
  int16_t *converted;         /* int16 PCM made from 8/24/32-bit input, else NULL */
  struct AADWavInfo wav;      /* WAV-input modes */
  struct AADHeaderInfo head;  /* decode */
  uint8_t *out;               /* where the device stage delivers: inside out_map, or a malloc'd buffer (fallback) */
  uint64_t out_size;
  uint8_t *out_map;           /* the output FILE, created at its final size and mapped shared: results land in the page cache */
  uint64_t out_map_size, out_head;
  struct AADHipErrorStats stats;
  int device_slot;
  int done;
};

struct Options {
  int mode; /* 'e' 'd' 'r' 'g' 'c'; 'i' never reaches the device slots */
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

/* a private, writable copy of a (small) file: the list of inputs */
static int slurp_copy(struct File *f)
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
  f->mapped = 0;
  if (f->bytes == NULL || fread(f->bytes, 1, (size_t)n, fp) != (size_t)n) {
    fclose(fp);
    return 0;
  }
  fclose(fp);
  return 1;
}

/* an input file, mapped read-only */
static int slurp(struct File *f)
{
  struct stat st;
  const int fd = open(f->path, O_RDONLY);
  void *m;
  if (fd < 0) return 0;
  if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
    close(fd);
    return 0;
  }
  f->size = (uint64_t)st.st_size;
  f->in_dev = (uint64_t)st.st_dev;
  f->in_ino = (uint64_t)st.st_ino;
  f->mapped = 0;
  if (f->size == 0) { /* nothing to map; the parsers refuse it */
    close(fd);
    f->bytes = (uint8_t *)calloc(1, 16);
    return f->bytes != NULL;
  }
  m = mmap(NULL, (size_t)f->size, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (m == MAP_FAILED) return 0;
  (void)posix_madvise(m, (size_t)f->size, POSIX_MADV_SEQUENTIAL);
  f->bytes = (uint8_t *)m;
  f->mapped = 1;
  return 1;
}

static void drop_input(struct File *f)
{
  if (f->bytes != NULL) {
    if (f->mapped) (void)munmap(f->bytes, (size_t)f->size);
    else free(f->bytes);
  }
  f->bytes = NULL;
  f->mapped = 0;
}

static void release_file(struct File *f)
{
  drop_input(f);
  free(f->converted);
  if (f->out_map != NULL) (void)munmap(f->out_map, (size_t)f->out_map_size);
  else free(f->out);
  f->out = f->out_map = NULL;
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

static int out_path(char *path, size_t cap, const char *outdir, const char *inpath, const char *ext)
{
  const char *base;
  size_t stem;
  stem_of(inpath, &base, &stem);
  return snprintf(path, cap, "%s/%.*s%s", outdir, (int)stem, base, ext) < (int)cap;
}

/* Create the output file at its final size and map it: the library then copies its results straight
 * into the page cache from its staging threads - no buffer of our own, no write() pass by a single
 * thread (decode output is four times its input: that pass was the slowest stage).  Leaves
 * f->out_map NULL when the file system will not do it; the caller then falls back to a buffer. */
static void map_output(const struct Options *opt, struct File *f, const char *ext, const uint8_t *head, size_t head_size, uint64_t body_size)
{
  char path[4096];
  int fd;
  void *m;
  struct stat st;
  f->out_map = NULL;
  if (head_size + body_size == 0 || !out_path(path, sizeof(path), opt->outdir, f->path, ext)) return;
  /* `-r -o .` on ./x.wav: the output IS the (mapped) input.  Truncating it now would pull the pages from
   * under the device stage; the buffered path writes it after the input has been consumed, as before. */
  if (stat(path, &st) == 0 && (uint64_t)st.st_dev == f->in_dev && (uint64_t)st.st_ino == f->in_ino) return;
  /* The file is built under "<name>.part" and gets its name when it is complete (finish_mapped): whatever ends this
   * process early - SIGBUS on a page the device cannot back included - never leaves a file of the final name and size
   * with holes that read back as silence.  Its blocks are reserved up front (posix_fallocate: a full device is reported
   * HERE, and the caller falls back to the buffered write, which names the failing file); a file system that cannot
   * reserve (EOPNOTSUPP / EINVAL) gets the sparse file of before. */
  if (strlen(path) + 6 > sizeof(path)) return;
  strcat(path, ".part");
  fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0666);
  if (fd < 0) return;
  {
    const int e = posix_fallocate(fd, 0, (off_t)(head_size + body_size));
    if ((e != 0 && e != EOPNOTSUPP && e != EINVAL) || (e != 0 && ftruncate(fd, (off_t)(head_size + body_size)) != 0)) {
      close(fd);
      (void)unlink(path);
      return;
    }
  }
  m = mmap(NULL, (size_t)(head_size + body_size), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) {
    (void)unlink(path);
    return;
  }
  f->out_map = (uint8_t *)m;
  f->out_map_size = head_size + body_size;
  f->out_head = head_size;
  if (head_size) memcpy(f->out_map, head, head_size);
  f->out = f->out_map + head_size;
}

/* a mapped output is complete once the device stage is done with it; returns 0 on failure */
static int finish_mapped(const struct Options *opt, struct File *f, const char *ext, int keep)
{
  char path[4096], part[4096 + 8];
  const uint64_t final_size = f->out_head + f->out_size;
  (void)munmap(f->out_map, (size_t)f->out_map_size);
  f->out_map = f->out = NULL;
  if (!out_path(path, sizeof(path), opt->outdir, f->path, ext)) return 0;
  snprintf(part, sizeof(part), "%s.part", path);
  if (!keep) return unlink(part) == 0;
  if (final_size != f->out_map_size && truncate(part, (off_t)final_size) != 0) {
    (void)unlink(part);
    return 0;
  }
  if (rename(part, path) != 0) { /* complete: now it gets its name */
    (void)unlink(part);
    return 0;
  }
  return 1;
}

static int write_out(const char *outdir, const char *inpath, const char *ext, const uint8_t *head, size_t head_size,
                     const uint8_t *body, size_t body_size)
{
  char path[4096];
  FILE *fp;
  if (!out_path(path, sizeof(path), outdir, inpath, ext)) return 0;
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
    drop_input(f); /* only the converted samples are needed from here on */
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
    /* a wave is sized by what it holds in memory at once: its inputs are mapped, its outputs are
     * allocated - a quarter of the input for -e, up to four times the input for -d */
    const uint64_t weight = s->opt->mode == 'd' ? 4 : 1;
    /* -r / -g / -c take the same waves as -e / -d: AADHip_ReconstructBatch stages its PCM through the pinned blocks in chunks
     * and keeps only device memory for the whole wave (round 4; before that the call was staged whole and waves were capped at 256 MB) */
    const uint64_t wave_bytes = s->opt->wave_bytes;
    while (i < s->nfiles && (i == w.first || (bytes + weight * s->files[i]->disk_size <= wave_bytes && i - w.first < WAVE_FILES))) {
      bytes += weight * s->files[i]->disk_size;
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
  uint32_t *decoded = (uint32_t *)malloc(sizeof(*decoded) * (size_t)n);
  uint64_t *sizes = (uint64_t *)malloc(sizeof(*sizes) * (size_t)n);
  uint64_t *got = (uint64_t *)malloc(sizeof(*got) * (size_t)n);
  struct AADHipErrorStats *stats = (struct AADHipErrorStats *)malloc(sizeof(*stats) * (size_t)n);
  AADApiResult r = AAD_APIRESULT_NG;
  int k, ok = 0;
  if (!in || !out || !frames || !decoded || !sizes || !got || !stats) goto done;

  if (mode == 'd') {
    for (k = 0; k < n; k++) {
      in[k] = g[k]->bytes;
      sizes[k] = g[k]->size;
      frames[k] = g[k]->head.num_samples;
      g[k]->out_size = (uint64_t)frames[k] * g[k]->head.num_channels * 2;
      {
        uint8_t head[AAD_WAV_HEADER_SIZE];
        AADWav_WriteHeader(head, sizeof(head), g[k]->head.num_channels, g[k]->head.sampling_rate, frames[k]);
        map_output(opt, g[k], ".wav", head, sizeof(head), g[k]->out_size);
      }
      if (g[k]->out_map == NULL) g[k]->out = (uint8_t *)malloc(((size_t)frames[k] * g[k]->head.num_channels + 8) * 2);
      out[k] = g[k]->out;
      if (out[k] == NULL) goto done;
    }
    r = AADHip_DecodeBatch(ctx, (uint32_t)n, (const uint8_t *const *)in, sizes, (int16_t *const *)out, frames, decoded);
    for (k = 0; r == AAD_APIRESULT_OK && k < n; k++) /* an image that ends early: the frames it does not hold are silence */
      if (decoded[k] < frames[k])
        memset(out[k] + (size_t)decoded[k] * g[k]->head.num_channels * 2, 0, (size_t)(frames[k] - decoded[k]) * g[k]->head.num_channels * 2);
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
        g[k]->out_size = sizes[k];
        if (mode == 'e') {
          map_output(opt, g[k], ".aad", NULL, 0, sizes[k]);
        } else {
          uint8_t head[AAD_WAV_HEADER_SIZE];
          AADWav_WriteHeader(head, sizeof(head), g[k]->wav.num_channels, g[k]->wav.sampling_rate, frames[k]);
          map_output(opt, g[k], ".wav", head, sizeof(head), sizes[k]);
        }
        if (g[k]->out_map == NULL) g[k]->out = (uint8_t *)malloc((size_t)sizes[k] + 16);
        out[k] = g[k]->out;
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
  free(decoded);
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
      if (f->out_map != NULL) { /* already in the file: keep it, or take it away again after a failure */
        wrote = finish_mapped(opt, f, opt->mode == 'e' ? ".aad" : ".wav", !failed && f->done);
        if (!wrote && !failed && f->done) {
          fprintf(stderr, "aad_batch: cannot write output for %s\n", f->path);
          failed = 1;
        }
      } else if (!failed && f->done && opt->mode != 'c') {
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

static void print_usage_lines(FILE *to)
{
  fprintf(to, "usage: aad_batch -e|-r|-g [-b bits] [-s max_block_size] [-t trials] [-m] [-D dev,dev,...] -o OUTDIR [-l LIST] in.wav...\n"
              "       aad_batch -c       [-b bits] [-s max_block_size] [-t trials] [-m] [-D dev,dev,...] [-l LIST] in.wav...\n"
              "       aad_batch -d [-D dev,dev,...] -o OUTDIR [-l LIST] in.aad...\n"
              "       aad_batch -i [-l LIST] in.aad...\n"
              "       aad_batch -h | -v\n");
}

static int usage(void)
{
  print_usage_lines(stderr);
  return 2;
}

/* -h: the reference's option table (src/main.c:20-58) in the layout its parser prints it
 * (src/command_line_parser.c:81-100: "  -c, --long" in 20 columns, "(needs argument)" in 18, the text),
 * then the options only this front end has */
static int print_help(void)
{
  static const struct { char c; const char *name; int arg; const char *text; } spec[] = {
    {'e', "encode", 0, "Encode mode (wav file -> .aad file)"},
    {'d', "decode", 0, "Decode mode (.aad file -> wav file)"},
    {'r', "reconstruct", 0, "Reconstruction mode (wav file -> (encode -> decode) -> decoded wav file)"},
    {'g', "gap", 0, "Gap(residual output) mode (wav file -> (encode -> decode) -> residual wav file)"},
    {'c', "calculate", 0, "Calculate statistics(e.g. RMS error) between original and reconstructed wav"},
    {'i', "information", 0, "Show information of encoded .aad file"},
    {'b', "bits-per-sample", 1, "Specify bits per sample(in 2,3,4) (default: 4)"},
    {'s', "max-block-size", 1, "Specify max block size (default: 1024)"},
    {'t', "num-encode-trials", 1, "Specify number of encode Trials (default: 2)"},
    {'m', "ms-conversion", 0, "Switch to use LR to MS conversion (default: no)"},
    {'h', "help", 0, "Show help message"},
    {'v', "version", 0, "Show version information"},
    {'o', "output-dir", 1, "Directory the outputs go to, as <input stem>.aad / .wav (every mode but -c and -i)"},
    {'l', "list", 1, "File with one input path per line, in addition to the paths on the command line"},
    {'D', "devices", 1, "Comma-separated device indices; a device may be named more than once (default: AAD_HIP_DEVICE or 0)"},
  };
  size_t k;
  print_usage_lines(stdout);
  printf("options: \n");
  for (k = 0; k < sizeof(spec) / sizeof(spec[0]); k++) {
    char command[64];
    snprintf(command, sizeof(command), "  -%c, --%s", spec[k].c, spec[k].name);
    printf("%-20s %-18s  %s \n", command, spec[k].arg ? "(needs argument)" : "", spec[k].text);
  }
  return 0;
}

/* -v: the reference's line (src/main.c:511-514) */
static int print_version(void)
{
  printf("AAD(Ayashi Adaptive Differential pulse code modulation) encoder/decoder Version.%d \n", AAD_CODEC_VERSION);
  return 0;
}

/* -i: the header of every input, in the reference's ten lines (src/main.c:229-272).  No device work.  With more than
 * one input every report is preceded by a line naming the file. */
static int print_information(char *const *paths, int npaths)
{
  static const char *const ch_process[] = {"None", "MS-Conversion"};
  int i, rc = 0;
  for (i = 0; i < npaths; i++) {
    uint8_t buffer[AAD_HEADER_SIZE];
    struct AADHeaderInfo h;
    AADApiResult ret;
    FILE *fp = fopen(paths[i], "rb");
    if (fp == NULL) {
      fprintf(stderr, "Failed to open %s. \n", paths[i]);
      rc = 1;
      continue;
    }
    if (fread(buffer, 1, AAD_HEADER_SIZE, fp) < AAD_HEADER_SIZE) {
      fprintf(stderr, "Failed to read from %s. \n", paths[i]);
      fclose(fp);
      rc = 1;
      continue;
    }
    fclose(fp);
    if ((ret = AADDecoder_DecodeHeader(buffer, AAD_HEADER_SIZE, &h)) != AAD_APIRESULT_OK) {
      fprintf(stderr, "Failed to read header. API result: %d \n", (int)ret);
      rc = 1;
      continue;
    }
    if (npaths > 1) printf("%s\n", paths[i]);
    printf("%-30s %-9d   \n", "Format Version:", (int)h.format_version);
    printf("%-30s %-9d   \n", "Codec Version:", (int)h.codec_version);
    printf("%-30s %-9d   \n", "Number of Channels:", (int)h.num_channels);
    printf("%-30s %-9d   \n", "Number of Samples per Channel:", (int)h.num_samples);
    printf("%-30s %-9d   \n", "Sampling Rate:", (int)h.sampling_rate);
    printf("%-30s %-9d   \n", "Bits per Sample:", (int)h.bits_per_sample);
    printf("%-30s %-9d   \n", "Block size:", (int)h.block_size);
    printf("%-30s %-9d   \n", "Number of Samples per Block:", (int)h.num_samples_per_block);
    printf("%-30s %-9s   \n", "Channel Processing:", ch_process[h.ch_process_method == AAD_CH_PROCESS_METHOD_MS ? 1 : 0]);
    printf("%-30s %-8.1f \n", "Bits per Second(bps):", (8.0f * (double)h.block_size * h.sampling_rate) / h.num_samples_per_block);
  }
  return rc;
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

/* a mapped output file that cannot get its pages (device full) or a mapped input that shrank */
static void on_sigbus(int sig)
{
  static const char msg[] = "aad_batch: I/O error on a mapped file (output device full, or an input changed while it was read)\n";
  (void)sig;
  if (write(2, msg, sizeof(msg) - 1) < 0) _exit(1);
  _exit(1);
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

  signal(SIGBUS, on_sigbus);
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
    else if (is_opt(a, "-i", "--information")) opt.mode = 'i';
    else if (is_opt(a, "-h", "--help")) { /* as in the reference: help and version win over everything else */
      free(paths);
      return print_help();
    } else if (is_opt(a, "-v", "--version")) {
      free(paths);
      return print_version();
    }
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
  if (opt.mode == 0 || (opt.mode != 'c' && opt.mode != 'i' && opt.outdir == NULL)) goto bad_usage;

  if (list != NULL) { /* one path per line; LF, CRLF or bare CR endings */
    struct File lf;
    char *p;
    memset(&lf, 0, sizeof(lf));
    lf.path = list;
    if (!slurp_copy(&lf)) {
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
  if (opt.mode == 'i') {
    rc = print_information(paths, npaths);
    goto cleanup;
  }

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
