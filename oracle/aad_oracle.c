/*
 * aad_oracle.c - CPU restatement of the AAD encode/decode path (see aad_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY - never linked into the product library.
 * Parity status: PINNED against the reference's fixtures and the compiled
 * reference (oracle/_ref); see tests/test_oracle_golden.py, tests/test_oracle_vs_ref.py.
 *
 * Integer widths follow SURVEY.md finding 5: all state arithmetic is 32-bit
 * two's complement with wraparound (done here in uint32_t so it is defined C),
 * right shifts of negatives are arithmetic, the quantiser divides truncating.
 */
#include "aad_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../aad_amd/csrc/aad_tables_data.h"

#define BLOCK_HEADER_BYTES_PER_CH 18u /* reference src/aad_internal.h:34 */
#define FORMAT_VERSION 4u             /* reference src/aad.h:10 */
#define CODEC_VERSION 18u             /* reference src/aad.h:7 */

static const uint16_t k_step[AAD_STEP_TABLE_LEN] = {AAD_STEP_TABLE_VALUES};
static const int16_t k_delta4[8] = {AAD_INDEX_DELTA_4BIT};
static const int16_t k_delta3[4] = {AAD_INDEX_DELTA_3BIT};
static const int16_t k_delta2[2] = {AAD_INDEX_DELTA_2BIT};

const uint16_t *aado_step_table(void) { return k_step; }

const int16_t *aado_index_deltas(uint32_t bits)
{
  return bits == 4 ? k_delta4 : bits == 3 ? k_delta3 : bits == 2 ? k_delta2 : NULL;
}

/* ---- wrap-around 32-bit helpers ------------------------------------------------------ */
static int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
static int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static int32_t wneg(int32_t a) { return (int32_t)(0u - (uint32_t)a); }
static int32_t clip16(int32_t v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }

/* ---- the recurrence ------------------------------------------------------------------ */

/* (16384 + sum h*w) >> 15 : reference src/aad_encoder.c:359-363, src/aad_decoder.c:291-295 */
static int32_t lane_predict(const AadoLane *l)
{
  int32_t acc = 16384;
  for (int i = 0; i < AADO_TAPS; i++) acc = wadd(acc, wmul(l->h[i], l->w[i]));
  return acc >> 15;
}

/* step index, LMS and history update shared by both directions:
 * reference src/aad_encoder.c:386-406, src/aad_decoder.c:303-315 */
static void lane_advance(AadoLane *l, uint32_t mag, uint32_t bits, int32_t qd, int32_t y)
{
  int32_t idx = l->idx + aado_index_deltas(bits)[mag];
  l->idx = idx < 0 ? 0 : idx > AAD_STEP_INDEX_MAX ? AAD_STEP_INDEX_MAX : idx;
  for (int i = 0; i < AADO_TAPS; i++)
    l->w[i] = wadd(l->w[i], wadd(wmul(qd, l->h[i]), 16384) >> 18);
  l->h[3] = l->h[2];
  l->h[2] = l->h[1];
  l->h[1] = l->h[0];
  l->h[0] = y;
}

static int32_t lane_step_size(const AadoLane *l) /* reference src/aad_tables.h:15,28 */
{
  return k_step[(l->idx + 8) >> 4];
}

/* one encoder step: reference src/aad_encoder.c:343-410 */
uint32_t aado_encode_step(AadoLane *l, int32_t x, uint32_t bits)
{
  const uint32_t signbit = 1u << (bits - 1), magmax = signbit - 1;
  const int32_t step = lane_step_size(l);
  const int32_t p = lane_predict(l);
  const int32_t d = wadd(x, wneg(p));
  const int neg = d < 0;
  const int32_t a = neg ? wneg(d) : d;
  uint32_t mag = (uint32_t)(wmul(a, 1 << (bits - 2)) / step); /* truncating division */
  if (mag > magmax) mag = magmax;
  int32_t qd = (step * (int32_t)(2 * mag + 1)) >> (bits - 1);
  if (neg) qd = -qd;
  l->qerr = qd;
  lane_advance(l, mag, bits, qd, clip16(wadd(qd, p)));
  return mag | (neg ? signbit : 0u);
}

/* one decoder step: reference src/aad_decoder.c:269-318 */
int32_t aado_decode_step(AadoLane *l, uint32_t code, uint32_t bits)
{
  const uint32_t signbit = 1u << (bits - 1), mag = code & (signbit - 1);
  const int32_t step = lane_step_size(l);
  int32_t qd = (step * (int32_t)(2 * mag + 1)) >> (bits - 1);
  if (code & signbit) qd = -qd;
  const int32_t y = clip16(wadd(qd, lane_predict(l)));
  lane_advance(l, mag, bits, qd, y);
  return y;
}

/* ---- geometry and headers ------------------------------------------------------------ */

static uint32_t gcd_u32(uint32_t a, uint32_t b)
{
  while (b) {
    uint32_t t = a % b;
    a = b;
    b = t;
  }
  return a;
}

typedef struct {
  uint32_t unit_bytes_per_ch; /* lcm(8,bits)/8 */
  uint32_t unit_samples;      /* lcm(8,bits)/bits */
} PackUnit;

static PackUnit pack_unit(uint32_t bits)
{
  PackUnit u;
  uint32_t l = 8u * bits / gcd_u32(8u, bits);
  u.unit_bytes_per_ch = l / 8u;
  u.unit_samples = l / bits;
  return u;
}

/* reference src/aad_encoder.c:85-131 (without the reference's 2-channel cap) */
int aado_block_geometry(uint32_t max_block_size, uint32_t channels, uint32_t bits,
                        uint32_t *block_size, uint32_t *samples_per_block)
{
  if (!block_size) return AADO_INVALID_ARGUMENT;
  if (channels == 0 || channels > AADO_MAX_CHANNELS || bits == 0 || bits > 4) return AADO_INVALID_FORMAT;
  if (max_block_size < BLOCK_HEADER_BYTES_PER_CH * channels) return AADO_INVALID_FORMAT;
  PackUnit u = pack_unit(bits);
  uint32_t unit_bytes = u.unit_bytes_per_ch * channels;
  uint32_t units = (max_block_size - BLOCK_HEADER_BYTES_PER_CH * channels) / unit_bytes;
  *block_size = (BLOCK_HEADER_BYTES_PER_CH * channels + units * unit_bytes) & 0xFFFFu;
  if (samples_per_block) *samples_per_block = AADO_TAPS + units * u.unit_samples;
  return AADO_OK;
}

static uint8_t *put_be(uint8_t *p, uint32_t v, int bytes)
{
  for (int i = bytes - 1; i >= 0; i--) *p++ = (uint8_t)(v >> (8 * i));
  return p;
}

static uint32_t get_be(const uint8_t *p, int bytes)
{
  uint32_t v = 0;
  for (int i = 0; i < bytes; i++) v = (v << 8) | p[i];
  return v;
}

/* field checks shared by writer and reader: reference src/aad_encoder.c:152-185, src/aad_decoder.c:189-222 */
static int header_fields_ok(const AadoHeader *hd, uint32_t max_channels)
{
  if (hd->num_channels == 0 || hd->num_channels > max_channels) return 0;
  if (hd->num_samples == 0 || hd->sampling_rate == 0) return 0;
  if (hd->bits_per_sample < 2 || hd->bits_per_sample > 4) return 0;
  if (hd->block_size <= BLOCK_HEADER_BYTES_PER_CH * hd->num_channels) return 0;
  if (hd->samples_per_block == 0) return 0;
  if (hd->ch_process_method >= 2) return 0;
  if (hd->ch_process_method == 1 && hd->num_channels == 1) return 0;
  return 1;
}

int aado_put_header(const AadoHeader *hd, uint8_t *out, size_t cap)
{
  if (!hd || !out) return AADO_INVALID_ARGUMENT;
  if (cap < AADO_FILE_HEADER_BYTES) return AADO_INSUFFICIENT_DATA;
  if (!header_fields_ok(hd, AADO_MAX_CHANNELS)) return AADO_INVALID_FORMAT;
  uint8_t *p = out;
  *p++ = 'A';
  *p++ = 'A';
  *p++ = 'D';
  *p++ = 0;
  p = put_be(p, FORMAT_VERSION, 4); /* the struct's version fields are ignored: src/aad_encoder.c:195-200 */
  p = put_be(p, CODEC_VERSION, 4);
  p = put_be(p, hd->num_channels, 2);
  p = put_be(p, hd->num_samples, 4);
  p = put_be(p, hd->sampling_rate, 4);
  p = put_be(p, hd->bits_per_sample, 2);
  p = put_be(p, hd->block_size, 2);
  p = put_be(p, hd->samples_per_block, 4);
  p = put_be(p, hd->ch_process_method, 1);
  return AADO_OK;
}

int aado_get_header(const uint8_t *d, size_t size, AadoHeader *hd)
{
  if (!d || !hd) return AADO_INVALID_ARGUMENT;
  if (size < AADO_FILE_HEADER_BYTES) return AADO_INSUFFICIENT_DATA;
  if (d[0] != 'A' || d[1] != 'A' || d[2] != 'D' || d[3] != 0) return AADO_INVALID_FORMAT;
  hd->format_version = get_be(d + 4, 4);
  hd->codec_version = get_be(d + 8, 4);
  hd->num_channels = get_be(d + 12, 2);
  hd->num_samples = get_be(d + 14, 4);
  hd->sampling_rate = get_be(d + 18, 4);
  hd->bits_per_sample = get_be(d + 22, 2);
  hd->block_size = get_be(d + 24, 2);
  hd->samples_per_block = get_be(d + 26, 4);
  hd->ch_process_method = d[30];
  return AADO_OK;
}

int aado_check_header(const AadoHeader *hd, uint32_t max_channels)
{
  if (hd->format_version != FORMAT_VERSION || hd->codec_version != CODEC_VERSION) return AADO_INVALID_FORMAT;
  return header_fields_ok(hd, max_channels) ? AADO_OK : AADO_INVALID_FORMAT;
}

static size_t block_bytes(uint32_t n, uint32_t channels, uint32_t bits)
{
  PackUnit u = pack_unit(bits);
  size_t units = n > AADO_TAPS ? (n - AADO_TAPS + u.unit_samples - 1) / u.unit_samples : 0;
  return (size_t)BLOCK_HEADER_BYTES_PER_CH * channels + units * u.unit_bytes_per_ch * channels;
}

size_t aado_encoded_size(uint32_t num_samples, uint32_t channels, uint32_t bits, uint32_t max_block_size)
{
  uint32_t bs, spb;
  if (aado_block_geometry(max_block_size, channels, bits, &bs, &spb) != AADO_OK) return 0;
  size_t full = num_samples / spb, tail = num_samples % spb;
  return AADO_FILE_HEADER_BYTES + full * block_bytes(spb, channels, bits) + (tail ? block_bytes(tail, channels, bits) : 0);
}

/* ---- encoder ------------------------------------------------------------------------- */

/* RMSE pass over one block while adapting `l`: reference src/aad_encoder.c:431-467.
 * x is one channel's samples (already mid/side transformed), planar. */
static double rmse_pass(AadoLane *l, const int32_t *x, uint32_t n, uint32_t bits)
{
  if (n < AADO_TAPS) return 0.0;
  for (uint32_t k = 0; k < AADO_TAPS; k++) l->h[AADO_TAPS - 1 - k] = x[k];
  double sum = 0.0;
  for (uint32_t s = AADO_TAPS; s < n; s++) {
    aado_encode_step(l, x[s], bits);
    sum += (double)wmul(l->qerr, l->qerr); /* int32-wrapped square, then to double */
  }
  return sqrt(sum / (double)n);
}

/* trial search for one channel: reference src/aad_encoder.c:470-562 */
static void search_best_lane(AadoLane *lane, const int32_t *x, uint32_t progress, uint32_t n,
                             uint32_t spb, uint32_t bits, uint32_t trials)
{
  const int have_prev = progress >= spb;
  AadoLane best = *lane, probe = *lane, run = *lane;
  double best_rmse = rmse_pass(&probe, x + progress, n, bits);
  for (uint32_t t = 0; t < trials; t++) {
    if (have_prev) (void)rmse_pass(&run, x + progress - spb, spb, bits);
    AadoLane cand = run;
    double r = rmse_pass(&run, x + progress, n, bits);
    if (best_rmse > r) {
      best_rmse = r;
      best = cand;
    }
  }
  *lane = best;
}

/* MSB-first bit packer for one pack unit */
static uint8_t *emit_unit(uint8_t *p, const uint32_t *codes, uint32_t count, uint32_t bits, uint32_t bytes)
{
  uint32_t acc = 0;
  for (uint32_t i = 0; i < count; i++) acc = (acc << bits) | codes[i];
  return put_be(p, acc, (int)bytes);
}

/* one block: reference src/aad_encoder.c:565-727.  x[c] planar, n <= spb valid samples at x[c][0..n). */
static size_t encode_block(AadoLane *lanes, const int32_t *const *x, uint32_t n, uint32_t channels,
                           uint32_t bits, uint8_t *out)
{
  const PackUnit u = pack_unit(bits);
  uint8_t *p = out;
  for (uint32_t c = 0; c < channels; c++) {
    AadoLane *l = &lanes[c];
    for (uint32_t k = 0; k < AADO_TAPS; k++) l->h[AADO_TAPS - 1 - k] = k < n ? x[c][k] : 0;
    /* weight shift so the largest magnitude fits 16 bits; low bits are dropped from the
     * encoder's own state too (src/aad_encoder.c:623-641) */
    int32_t maxabs = 0;
    for (int k = 0; k < AADO_TAPS; k++) {
      int32_t a = l->w[k] >= 0 ? l->w[k] : wneg(l->w[k]);
      if (maxabs < a) maxabs = a;
    }
    uint32_t shift = 0;
    while (maxabs > 32767) {
      maxabs >>= 1;
      shift++;
    }
    const int32_t mask = (int32_t)~((1u << shift) - 1u);
    for (int k = 0; k < AADO_TAPS; k++) l->w[k] &= mask;
    p = put_be(p, (((uint32_t)l->idx << 4) & 0xFFFFu) | (shift & 0xFu), 2);
    for (int k = 0; k < AADO_TAPS; k++) {
      p = put_be(p, (uint32_t)(l->w[k] >> shift) & 0xFFFFu, 2);
      p = put_be(p, (uint32_t)l->h[k] & 0xFFFFu, 2);
    }
  }
  for (uint32_t s = AADO_TAPS; s < n; s += u.unit_samples) {
    for (uint32_t c = 0; c < channels; c++) {
      uint32_t codes[8];
      for (uint32_t k = 0; k < u.unit_samples; k++) /* samples past n are zero padding (:592-593) */
        codes[k] = aado_encode_step(&lanes[c], s + k < n ? x[c][s + k] : 0, bits);
      p = emit_unit(p, codes, u.unit_samples, bits, u.unit_bytes_per_ch);
    }
  }
  return (size_t)(p - out);
}

/* planar int32 copy of the stream with the optional L/R -> M/S transform applied
 * (src/aad_encoder.c:413-428; per-sample, so doing it once up front is equivalent) */
static int32_t **planarize(const int16_t *pcm, uint32_t n, uint32_t channels, int ms)
{
  int32_t **x = (int32_t **)calloc(channels, sizeof(*x));
  if (!x) return NULL;
  for (uint32_t c = 0; c < channels; c++) {
    x[c] = (int32_t *)malloc(sizeof(int32_t) * (n ? n : 1));
    for (uint32_t s = 0; s < n; s++) x[c][s] = pcm[(size_t)s * channels + c];
  }
  if (ms && channels >= 2) {
    for (uint32_t s = 0; s < n; s++) {
      int32_t l = x[0][s], r = x[1][s];
      x[0][s] = clip16((l + r) >> 1);
      x[1][s] = clip16((l - r) >> 1);
    }
  }
  return x;
}

static void free_planar(int32_t **x, uint32_t channels)
{
  if (!x) return;
  for (uint32_t c = 0; c < channels; c++) free(x[c]);
  free(x);
}

int aado_encode_stream(const int16_t *pcm, uint32_t num_samples, uint32_t channels,
                       uint32_t sampling_rate, uint32_t bits, uint32_t max_block_size,
                       uint32_t ch_process_method, uint32_t trials, AadoLane *lanes,
                       uint8_t *out, size_t cap, size_t *out_size)
{
  if (!pcm || !lanes || !out || !out_size) return AADO_INVALID_ARGUMENT;
  AadoHeader hd;
  memset(&hd, 0, sizeof(hd));
  if (ch_process_method >= 2) return AADO_INVALID_FORMAT;
  if (aado_block_geometry(max_block_size, channels, bits, &hd.block_size, &hd.samples_per_block) != AADO_OK)
    return AADO_INVALID_FORMAT;
  hd.num_channels = channels;
  hd.num_samples = num_samples;
  hd.sampling_rate = sampling_rate;
  hd.bits_per_sample = bits;
  hd.ch_process_method = ch_process_method;
  int rc = aado_put_header(&hd, out, cap);
  if (rc != AADO_OK) return rc;
  if (cap < aado_encoded_size(num_samples, channels, bits, max_block_size)) return AADO_INSUFFICIENT_BUFFER;

  int32_t **x = planarize(pcm, num_samples, channels, ch_process_method == 1);
  if (!x) return AADO_NG;
  const uint32_t spb = hd.samples_per_block;
  uint8_t *p = out + AADO_FILE_HEADER_BYTES;
  for (uint32_t progress = 0; progress < num_samples;) {
    const uint32_t n = num_samples - progress < spb ? num_samples - progress : spb;
    const int32_t *cur[AADO_MAX_CHANNELS];
    for (uint32_t c = 0; c < channels; c++) {
      if (trials) search_best_lane(&lanes[c], x[c], progress, n, spb, bits, trials);
      cur[c] = x[c] + progress;
    }
    p += encode_block(lanes, cur, n, channels, bits, p);
    progress += n;
  }
  free_planar(x, channels);
  *out_size = (size_t)(p - out);
  return AADO_OK;
}

/* ---- decoder ------------------------------------------------------------------------- */

/* `size`: what the caller declares as the block's bytes (the header check of :351-353); `readable`: bytes from `block` on that
 * may be READ.  The reference's code walk has no bound of its own (asserts only, :399-400): when the header's
 * samples_per_block asks for more codes than block_size holds (the header checks, :173-225, relate neither to the other) it
 * reads on into the bytes that follow - inside AADDecoder_DecodeWhole that is the rest of the file.  Beyond `readable`
 * (where the reference would leave its buffer) bytes read as zero. */
static int decode_block_readable(const AadoHeader *hd, const uint8_t *block, size_t size, size_t readable,
                                 int16_t *pcm, uint32_t want_frames, uint32_t *got_frames)
{
  if (!hd || !block || !pcm || !got_frames) return AADO_INVALID_ARGUMENT;
  const uint32_t ch = hd->num_channels, bits = hd->bits_per_sample;
  if (size < (size_t)BLOCK_HEADER_BYTES_PER_CH * ch) return AADO_INSUFFICIENT_DATA;
  const PackUnit u = pack_unit(bits);
  const uint32_t n = hd->samples_per_block < want_frames ? hd->samples_per_block : want_frames;
  AadoLane lanes[AADO_MAX_CHANNELS];
  const uint8_t *p = block;
  for (uint32_t c = 0; c < ch; c++) { /* reference src/aad_decoder.c:364-380 */
    uint32_t v = get_be(p, 2);
    p += 2;
    lanes[c].idx = (int32_t)(v >> 4);
    /* The reference takes the 12-bit field as it is (:365-366).  4081..4087 still select the last table entry
     * ((idx + 8) >> 4 = 255) and the walk continues from the unclamped value; 4088..4095 index past its
     * 256-entry table (undefined there) and are taken as 4087 here. */
    if (lanes[c].idx > AAD_STEP_INDEX_MAX + 7) lanes[c].idx = AAD_STEP_INDEX_MAX + 7;
    const uint32_t shift = v & 0xFu;
    for (int k = 0; k < AADO_TAPS; k++) {
      lanes[c].w[k] = (int32_t)((uint32_t)(int32_t)(int16_t)get_be(p, 2) << shift);
      lanes[c].h[k] = (int16_t)get_be(p + 2, 2);
      p += 4;
    }
    for (uint32_t k = 0; k < AADO_TAPS && k < want_frames; k++) /* :386-391 */
      pcm[(size_t)k * ch + c] = (int16_t)lanes[c].h[AADO_TAPS - 1 - k];
  }
  const uint8_t *end = block + readable;
  for (uint32_t s = AADO_TAPS; s < n; s += u.unit_samples) { /* :394-455 */
    for (uint32_t c = 0; c < ch; c++) {
      uint32_t acc = 0;
      for (uint32_t b = 0; b < u.unit_bytes_per_ch; b++, p++) acc = (acc << 8) | (p < end ? *p : 0u);
      for (uint32_t k = 0; k < u.unit_samples; k++) {
        uint32_t code = (acc >> (bits * (u.unit_samples - 1 - k))) & ((1u << bits) - 1u);
        int32_t y = aado_decode_step(&lanes[c], code, bits);
        if (s + k < n) pcm[(size_t)(s + k) * ch + c] = (int16_t)y;
      }
    }
  }
  if (hd->ch_process_method == 1) { /* :458-470 */
    if (ch < 2) return AADO_INVALID_FORMAT;
    for (uint32_t s = 0; s < n; s++) {
      int32_t m = pcm[(size_t)s * ch], sd = pcm[(size_t)s * ch + 1];
      pcm[(size_t)s * ch] = (int16_t)clip16(m + sd);
      pcm[(size_t)s * ch + 1] = (int16_t)clip16(m - sd);
    }
  }
  *got_frames = n;
  return AADO_OK;
}

int aado_decode_block(const AadoHeader *hd, const uint8_t *block, size_t size,
                      int16_t *pcm, uint32_t want_frames, uint32_t *got_frames)
{
  return decode_block_readable(hd, block, size, size, pcm, want_frames, got_frames);
}

int aado_decode_stream(const uint8_t *data, size_t size, uint32_t max_channels,
                       int16_t *pcm, uint32_t pcm_frames, AadoHeader *hd_out)
{
  if (!data || !pcm) return AADO_INVALID_ARGUMENT;
  AadoHeader hd;
  int rc = aado_get_header(data, size, &hd);
  if (rc != AADO_OK) return rc;
  if ((rc = aado_check_header(&hd, max_channels)) != AADO_OK) return rc;
  if (hd_out) *hd_out = hd;
  if (pcm_frames < hd.num_samples) return AADO_INSUFFICIENT_BUFFER;
  size_t off = AADO_FILE_HEADER_BYTES;
  uint32_t progress = 0;
  while (progress < hd.num_samples && off < size) { /* reference src/aad_decoder.c:514-534 */
    size_t take = size - off < hd.block_size ? size - off : hd.block_size;
    uint32_t got = 0;
    rc = decode_block_readable(&hd, data + off, take, size - off, pcm + (size_t)progress * hd.num_channels,
                               pcm_frames - progress, &got);
    if (rc != AADO_OK) return rc;
    off += take;
    progress += got;
  }
  return AADO_OK;
}

/* ---- batch helpers (cpu_baseline leg of bench.py) ------------------------------------- */

int aado_encode_batch(const int16_t *pcm, uint32_t num_streams, uint32_t num_samples,
                      uint32_t channels, uint32_t sampling_rate, uint32_t bits,
                      uint32_t max_block_size, uint32_t ch_process_method, uint32_t trials,
                      uint8_t *out, size_t out_stride)
{
  for (uint32_t s = 0; s < num_streams; s++) {
    AadoLane lanes[AADO_MAX_CHANNELS];
    memset(lanes, 0, sizeof(lanes));
    size_t got = 0;
    int rc = aado_encode_stream(pcm + (size_t)s * num_samples * channels, num_samples, channels, sampling_rate,
                                bits, max_block_size, ch_process_method, trials, lanes,
                                out + s * out_stride, out_stride, &got);
    if (rc != AADO_OK) return rc;
  }
  return AADO_OK;
}

int aado_decode_batch(const uint8_t *data, uint32_t num_streams, size_t stride, size_t size,
                      int16_t *pcm, uint32_t pcm_frames)
{
  for (uint32_t s = 0; s < num_streams; s++) {
    AadoHeader hd;
    int rc = aado_get_header(data + s * stride, size, &hd);
    if (rc != AADO_OK) return rc;
    rc = aado_decode_stream(data + s * stride, size, AADO_MAX_CHANNELS,
                            pcm + (size_t)s * pcm_frames * hd.num_channels, pcm_frames, NULL);
    if (rc != AADO_OK) return rc;
  }
  return AADO_OK;
}

/* ---- the CLI's reconstruction statistics (reference src/main.c:396-503) ------------------- */

static int32_t residual32(int16_t x, int16_t y) /* src/main.c:419-423 in the reader's 32-bit domain */
{
  return (int32_t)(((uint32_t)(int32_t)x << 16) - ((uint32_t)(int32_t)y << 16));
}

void aado_residual(const int16_t *x, const int16_t *y, size_t count, int16_t *out)
{
  size_t i;
  for (i = 0; i < count; i++) out[i] = (int16_t)(residual32(x[i], y[i]) >> 16); /* src/wav.c:429 */
}

void aado_error_stats(const int16_t *x, const int16_t *y, uint32_t num_samples, uint32_t channels,
                      double out[3])
{
  double sq = 0.0, ab = 0.0, mx = 0.0;
  uint32_t c, n;
  for (c = 0; c < channels; c++) {
    for (n = 0; n < num_samples; n++) {
      const size_t i = (size_t)n * channels + c;
      const double p1 = (double)residual32(x[i], y[i]) / INT32_MAX; /* src/main.c:483 */
      const double p2 = (double)y[i] / INT32_MAX;                   /* src/main.c:484 */
      sq += pow(p1 - p2, 2);
      ab += fabs(p1 - p2);
      if (mx < fabs(p1 - p2)) mx = fabs(p1 - p2);
    }
  }
  out[0] = sqrt(sq / ((double)channels * num_samples)); /* src/main.c:493-497 */
  out[1] = ab / ((double)channels * num_samples);
  out[2] = mx;
}
