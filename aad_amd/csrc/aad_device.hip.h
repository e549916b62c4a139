/*
 * aad_device.hip.h - CDNA4 (gfx950) device code of the AAD engine.
 *
 * One lane owns one independent recurrence:
 *   encode: lane = (stream, channel) - blocks of a stream are chained through the predictor
 *           state (SURVEY.md finding 2), so a stream is the smallest independent unit;
 *   decode: lane = (block, channel)  - every block header reloads the whole state.
 * The whole per-channel state (4 weights, 4 history samples, step index) lives in VGPRs; the
 * 256-entry step table sits in LDS next to fl32(0.5/step) so that the quantiser's division is a
 * convert-multiply-truncate (proved equal to the reference's integer division over every
 * reachable operand in tests/test_quantiser_equiv.py).  There is no contraction anywhere, hence
 * no MFMA: the kernels are integer VALU work streaming int16 PCM in and packed codes out.
 *
 * Arithmetic widths follow SURVEY.md finding 5: int32 wraparound (done in unsigned), arithmetic
 * right shifts, 24-bit multiplies only where both operands provably fit.
 */
#ifndef AAD_DEVICE_HIP_H
#define AAD_DEVICE_HIP_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "aad_tables_data.h"

namespace aad {

constexpr int kTaps = 4;
constexpr int kBlockHeaderBytesPerCh = 18;
constexpr int kFileHeaderBytes = 31;
constexpr int kIndexMax = AAD_STEP_INDEX_MAX;

struct StepEntry {
  int32_t step;
  float half_recip; /* fl32(0.5 / step) */
};

__constant__ uint16_t c_step_table[AAD_STEP_TABLE_LEN] = {AAD_STEP_TABLE_VALUES};
__constant__ uint32_t c_half_recip_bits[AAD_STEP_TABLE_LEN] = {AAD_HALF_RECIP_BITS};

/* mirrors struct AADHipStreamDesc (include/aad_hip.h) */
struct StreamDesc {
  uint64_t pcm_offset;
  uint64_t data_offset;
  uint64_t data_size;
  uint32_t num_samples;
  uint32_t reserved;
};

/* mirrors struct AADHipLaneState */
struct LaneStateRecord {
  int32_t weight[4];
  int32_t history[4];
  int32_t stepsize_index;
  int32_t quantize_error;
};

struct Lane {
  int32_t w0, w1, w2, w3; /* Q15 LMS weights */
  int32_t h0, h1, h2, h3; /* history, h0 newest */
  int32_t idx;            /* Q4 step index */
};

/* stage the step table into LDS; every thread of the workgroup must call this */
__device__ __forceinline__ void stage_step_table(StepEntry *tab)
{
  for (int i = threadIdx.x; i < AAD_STEP_TABLE_LEN; i += blockDim.x) {
    StepEntry e;
    e.step = c_step_table[i];
    e.half_recip = __uint_as_float(c_half_recip_bits[i]);
    tab[i] = e;
  }
  __syncthreads();
}

__device__ __forceinline__ int32_t clip16(int32_t v) { return min(max(v, -32768), 32767); }

/* (16384 + sum h*w) >> 15 with int32 wraparound - reference src/aad_encoder.c:359-363 */
__device__ __forceinline__ int32_t predict(const Lane &L)
{
  uint32_t acc = 16384u;
  acc += (uint32_t)L.h0 * (uint32_t)L.w0;
  acc += (uint32_t)L.h1 * (uint32_t)L.w1;
  acc += (uint32_t)L.h2 * (uint32_t)L.w2;
  acc += (uint32_t)L.h3 * (uint32_t)L.w3;
  return (int32_t)acc >> 15;
}

/* Q4 step-index delta of a magnitude code - the constants of reference src/aad_tables.c:8-45
 * (AAD_INDEX_DELTA_*BIT in aad_tables_data.h) folded into selects: 4-bit {-18,-17,-14,16,32,64,128,256},
 * 3-bit {-16,-15,32,128}, 2-bit {-14,40}. */
template <int BITS>
__device__ __forceinline__ int32_t index_delta(uint32_t mag)
{
  if (BITS == 4) {
    const int32_t low = mag == 0 ? -18 : (mag == 1 ? -17 : -14);
    return mag >= 3 ? (int32_t)(2u << mag) : low;
  } else if (BITS == 3) {
    return mag >= 2 ? (mag == 2 ? 32 : 128) : (int32_t)mag - 16;
  } else {
    return mag ? 40 : -14;
  }
}

/* step-index, LMS and history update - reference src/aad_encoder.c:386-406, src/aad_decoder.c:303-315.
 * qd*h fits 32 bits (|qd| <= 61438, |h| <= 32768), so the 24-bit multiplier is exact. */
template <int BITS>
__device__ __forceinline__ void advance(Lane &L, uint32_t mag, int32_t qd, int32_t y)
{
  L.idx = min(max(L.idx + index_delta<BITS>(mag), 0), kIndexMax);
  L.w0 = (int32_t)((uint32_t)L.w0 + (uint32_t)((__mul24(qd, L.h0) + 16384) >> 18));
  L.w1 = (int32_t)((uint32_t)L.w1 + (uint32_t)((__mul24(qd, L.h1) + 16384) >> 18));
  L.w2 = (int32_t)((uint32_t)L.w2 + (uint32_t)((__mul24(qd, L.h2) + 16384) >> 18));
  L.w3 = (int32_t)((uint32_t)L.w3 + (uint32_t)((__mul24(qd, L.h3) + 16384) >> 18));
  L.h3 = L.h2;
  L.h2 = L.h1;
  L.h1 = L.h0;
  L.h0 = y;
}

/* one encoder step - reference src/aad_encoder.c:343-410.  Returns the code; qd is the
 * dequantised difference (the reference's quantize_error). */
template <int BITS>
__device__ __forceinline__ uint32_t encode_step(Lane &L, int32_t x, const StepEntry *tab, int32_t &qd)
{
  constexpr uint32_t kSign = 1u << (BITS - 1), kMagMax = kSign - 1u;
  const StepEntry e = tab[(L.idx + 8) >> 4];
  const int32_t p = predict(L);
  const int32_t d = x - p;
  const int32_t m = d >> 31; /* 0 or -1 */
  const uint32_t a = (uint32_t)((d ^ m) - m);
  /* min((a << (BITS-2)) / step, magmax) as trunc(fl32(2*(a << (BITS-2)) + 1) * fl32(0.5/step)) */
  const uint32_t a2 = (a << (BITS - 1)) | 1u;
  const uint32_t mag = min((uint32_t)((float)a2 * e.half_recip), kMagMax);
  const int32_t q = (int32_t)(__umul24((uint32_t)e.step, 2u * mag + 1u) >> (BITS - 1));
  qd = (q ^ m) - m;
  advance<BITS>(L, mag, qd, clip16(qd + p));
  return mag | ((uint32_t)m & kSign);
}

/* one decoder step - reference src/aad_decoder.c:269-318 */
template <int BITS>
__device__ __forceinline__ int32_t decode_step(Lane &L, uint32_t code, const StepEntry *tab)
{
  constexpr uint32_t kSign = 1u << (BITS - 1), kMagMax = kSign - 1u;
  const int32_t step = tab[(L.idx + 8) >> 4].step;
  const uint32_t mag = code & kMagMax;
  const int32_t q = (int32_t)(__umul24((uint32_t)step, 2u * mag + 1u) >> (BITS - 1));
  const int32_t m = -(int32_t)((code >> (BITS - 1)) & 1u);
  const int32_t qd = (q ^ m) - m;
  const int32_t y = clip16(qd + predict(L));
  advance<BITS>(L, mag, qd, y);
  return y;
}

template <int BITS>
struct Pack {
  static constexpr int kUnitSamples = BITS == 3 ? 8 : (BITS == 4 ? 2 : 4);
  static constexpr int kUnitBytes = BITS == 3 ? 3 : 1;
};

/* ================================================================================ decode == */

struct DecodeArgs {
  const StreamDesc *streams;
  const uint64_t *block_prefix; /* [num_streams + 1] exclusive prefix sum of blocks per stream */
  const uint8_t *data;
  int16_t *pcm;
  uint64_t total_blocks;
  uint32_t num_streams;
  uint32_t channels;
  uint32_t block_size;
  uint32_t samples_per_block;
  uint32_t header_bytes; /* 31 (file image) or 0 (bare block) */
  uint32_t mid_side;
  uint32_t bits;
};

/* last stream whose first block index is <= g */
__device__ __forceinline__ uint32_t find_stream(const uint64_t *prefix, uint32_t num_streams, uint64_t g)
{
  uint32_t lo = 0, hi = num_streams;
  while (hi - lo > 1) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (prefix[mid] <= g) lo = mid; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ uint32_t load_be16(const uint8_t *p) { return ((uint32_t)p[0] << 8) | p[1]; }

/*
 * Block-parallel decode (reference src/aad_decoder.c:321-475, looped by :514-534).
 * lane = (global block, channel); adjacent lanes are the channels of one block so the packed
 * units they read are adjacent bytes and the frames they write are adjacent int16.
 */
template <int BITS>
__global__ void __launch_bounds__(64) decode_blocks_kernel(DecodeArgs a)
{
  __shared__ StepEntry tab[AAD_STEP_TABLE_LEN];
  stage_step_table(tab);

  const uint32_t ch = a.channels;
  const uint64_t lane = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = lane < a.total_blocks * ch;
  uint64_t g = active ? lane / ch : 0;
  const uint32_t c = active ? (uint32_t)(lane % ch) : 0;

  const uint32_t s = find_stream(a.block_prefix, a.num_streams, g);
  const StreamDesc sd = a.streams[s];
  const uint64_t b = g - a.block_prefix[s];
  const uint64_t first = b * a.samples_per_block;
  uint32_t n = 0;
  if (active && first < sd.num_samples) {
    const uint64_t left = sd.num_samples - first;
    n = left < a.samples_per_block ? (uint32_t)left : a.samples_per_block;
  }
  /* bytes of this stream still present from the start of this block */
  const uint64_t block_off = a.header_bytes + b * a.block_size;
  const uint64_t avail = sd.data_size > block_off ? sd.data_size - block_off : 0;
  const uint8_t *src = a.data + sd.data_offset + block_off;
  int16_t *dst = a.pcm + sd.pcm_offset + first * ch + c;
  if (avail < (uint64_t)kBlockHeaderBytesPerCh * ch) n = 0; /* DecodeBlock: INSUFFICIENT_DATA (host reports it) */

  Lane L = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (n) {
    const uint8_t *hp = src + c * kBlockHeaderBytesPerCh;
    const uint32_t v = load_be16(hp);
    L.idx = min((int32_t)(v >> 4), kIndexMax);
    const uint32_t shift = v & 0xFu;
    L.w0 = (int32_t)((uint32_t)(int32_t)(int16_t)load_be16(hp + 2) << shift);
    L.h0 = (int16_t)load_be16(hp + 4);
    L.w1 = (int32_t)((uint32_t)(int32_t)(int16_t)load_be16(hp + 6) << shift);
    L.h1 = (int16_t)load_be16(hp + 8);
    L.w2 = (int32_t)((uint32_t)(int32_t)(int16_t)load_be16(hp + 10) << shift);
    L.h2 = (int16_t)load_be16(hp + 12);
    L.w3 = (int32_t)((uint32_t)(int32_t)(int16_t)load_be16(hp + 14) << shift);
    L.h3 = (int16_t)load_be16(hp + 16);
  }

  const bool ms = a.mid_side != 0; /* host guarantees ch == 2 then */
  auto emit = [&](uint32_t i, int32_t y) {
    if (ms) {
      const int32_t other = __shfl_xor(y, 1);
      y = c == 0 ? clip16(y + other) : clip16(other - y);
    }
    if (i < n) dst[(uint64_t)i * ch] = (int16_t)y;
  };
  /* the first four samples are stored verbatim in the header - reference :386-391 */
  emit(0, L.h3);
  emit(1, L.h2);
  emit(2, L.h1);
  emit(3, L.h0);

  constexpr int US = Pack<BITS>::kUnitSamples, UB = Pack<BITS>::kUnitBytes;
  const uint64_t unit_base = (uint64_t)kBlockHeaderBytesPerCh * ch + (uint64_t)c * UB;
  const uint32_t unit_stride = UB * ch;
  for (uint32_t i = kTaps, u = 0; i < n; i += US, u++) {
    const uint64_t o = unit_base + (uint64_t)u * unit_stride;
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < UB; k++) acc = (acc << 8) | (o + k < avail ? (uint32_t)src[o + k] : 0u);
#pragma unroll
    for (int k = 0; k < US; k++) {
      const uint32_t code = (acc >> (BITS * (US - 1 - k))) & ((1u << BITS) - 1u);
      emit(i + k, decode_step<BITS>(L, code, tab));
    }
  }
}

/* ================================================================================ encode == */

struct EncodeArgs {
  const StreamDesc *streams;
  const int16_t *pcm;
  uint8_t *data;
  LaneStateRecord *state; /* may be null */
  uint32_t num_streams;
  uint32_t channels;
  uint32_t block_size;
  uint32_t samples_per_block;
  uint32_t mid_side;
  uint32_t trials;
  uint32_t bits;
  uint8_t header_template[32]; /* 31-byte file header with num_samples = 0 */
};

/* sample i of channel c of a stream, after the optional L/R -> M/S transform
 * (reference src/aad_encoder.c:413-428; the clip there can never trigger for int16 input) */
struct SampleSource {
  const int16_t *x;
  uint32_t ch, c, ms;
  __device__ __forceinline__ int32_t at(uint64_t i) const
  {
    if (ms) {
      const int32_t l = x[i * 2], r = x[i * 2 + 1];
      return c == 0 ? (l + r) >> 1 : (l - r) >> 1;
    }
    return x[i * ch + c];
  }
};

__device__ __forceinline__ void seed_history(Lane &L, const SampleSource &src, uint64_t first, uint32_t n)
{
  L.h3 = n > 0 ? src.at(first + 0) : 0;
  L.h2 = n > 1 ? src.at(first + 1) : 0;
  L.h1 = n > 2 ? src.at(first + 2) : 0;
  L.h0 = n > 3 ? src.at(first + 3) : 0;
}

/* RMSE of the dequantised differences over one block while the lane adapts -
 * reference src/aad_encoder.c:431-467.  The squares wrap in int32 like the reference's
 * (SURVEY.md finding 5); every partial sum is an exact integer < 2^53, so accumulating in
 * int64 and converting once equals the reference's running double sum. */
template <int BITS>
__device__ __forceinline__ double rmse_pass(Lane &L, const SampleSource &src, uint64_t first, uint32_t n,
                                            const StepEntry *tab)
{
  if (n < (uint32_t)kTaps) return 0.0;
  seed_history(L, src, first, n);
  int64_t sum = 0;
  for (uint32_t i = kTaps; i < n; i++) {
    int32_t qd;
    encode_step<BITS>(L, src.at(first + i), tab, qd);
    sum += (int64_t)(int32_t)((uint32_t)qd * (uint32_t)qd);
  }
  return sqrt((double)sum / (double)n);
}

/* trial search - reference src/aad_encoder.c:470-562 (per channel; channels are independent) */
template <int BITS>
__device__ __forceinline__ void search_best_lane(Lane &L, const SampleSource &src, uint64_t first, uint32_t n,
                                                 uint32_t spb, uint32_t trials, const StepEntry *tab)
{
  const bool have_prev = first >= spb;
  Lane best = L, run = L, probe = L;
  double best_rmse = rmse_pass<BITS>(probe, src, first, n, tab);
  for (uint32_t t = 0; t < trials; t++) {
    if (have_prev) (void)rmse_pass<BITS>(run, src, first - spb, spb, tab);
    const Lane cand = run;
    const double r = rmse_pass<BITS>(run, src, first, n, tab);
    if (best_rmse > r) {
      best_rmse = r;
      best = cand;
    }
  }
  L = best;
}

__device__ __forceinline__ void store_be16(uint8_t *p, uint32_t v)
{
  p[0] = (uint8_t)(v >> 8);
  p[1] = (uint8_t)v;
}

/* block header of one channel - reference src/aad_encoder.c:619-655.  Drops the weight bits the
 * 16-bit header fields cannot carry from the lane's own state as well. */
__device__ __forceinline__ void write_block_header(Lane &L, uint8_t *p)
{
  auto wabs = [](int32_t w) { const int32_t m = w >> 31; return (int32_t)(((uint32_t)w ^ (uint32_t)m) - (uint32_t)m); };
  const int32_t maxabs = max(max(max(wabs(L.w0), wabs(L.w1)), max(wabs(L.w2), wabs(L.w3))), 0);
  const int32_t shift = max(17 - (int32_t)__clz(maxabs), 0); /* smallest shift with maxabs >> shift <= 32767 */
  const int32_t mask = (int32_t)~((1u << shift) - 1u);
  L.w0 &= mask;
  L.w1 &= mask;
  L.w2 &= mask;
  L.w3 &= mask;
  store_be16(p, (((uint32_t)L.idx << 4) & 0xFFFFu) | ((uint32_t)shift & 0xFu));
  store_be16(p + 2, (uint32_t)(L.w0 >> shift));
  store_be16(p + 4, (uint32_t)L.h0);
  store_be16(p + 6, (uint32_t)(L.w1 >> shift));
  store_be16(p + 8, (uint32_t)L.h1);
  store_be16(p + 10, (uint32_t)(L.w2 >> shift));
  store_be16(p + 12, (uint32_t)L.h2);
  store_be16(p + 14, (uint32_t)(L.w3 >> shift));
  store_be16(p + 16, (uint32_t)L.h3);
}

/*
 * Stream-parallel encode (reference src/aad_encoder.c:814-891 with EncodeBlock :565-727 and the
 * optional trial search :470-562 inlined).  lane = (stream, channel).
 */
template <int BITS>
__global__ void __launch_bounds__(64) encode_streams_kernel(EncodeArgs a)
{
  __shared__ StepEntry tab[AAD_STEP_TABLE_LEN];
  stage_step_table(tab);

  const uint32_t ch = a.channels;
  const uint64_t lane = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (lane >= (uint64_t)a.num_streams * ch) return;
  const uint32_t s = (uint32_t)(lane / ch), c = (uint32_t)(lane % ch);
  const StreamDesc sd = a.streams[s];
  const SampleSource src = {a.pcm + sd.pcm_offset, ch, c, a.mid_side};
  uint8_t *out = a.data + sd.data_offset;
  const uint32_t total = sd.num_samples, spb = a.samples_per_block;

  Lane L = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (a.state) {
    const LaneStateRecord r = a.state[lane];
    L = {r.weight[0], r.weight[1], r.weight[2], r.weight[3],
         r.history[0], r.history[1], r.history[2], r.history[3], r.stepsize_index};
  }
  int32_t last_qd = a.state ? a.state[lane].quantize_error : 0;

  if (c == 0) { /* file header - reference src/aad_encoder.c:190-214 */
    for (int i = 0; i < kFileHeaderBytes; i++) out[i] = a.header_template[i];
    out[14] = (uint8_t)(total >> 24);
    out[15] = (uint8_t)(total >> 16);
    out[16] = (uint8_t)(total >> 8);
    out[17] = (uint8_t)total;
  }

  constexpr int US = Pack<BITS>::kUnitSamples, UB = Pack<BITS>::kUnitBytes;
  const uint32_t unit_stride = UB * ch;
  uint64_t block_off = kFileHeaderBytes;
  for (uint64_t first = 0; first < total; first += spb, block_off += a.block_size) {
    const uint32_t n = total - first < spb ? (uint32_t)(total - first) : spb;
    if (a.trials) search_best_lane<BITS>(L, src, first, n, spb, a.trials, tab);
    seed_history(L, src, first, n);
    write_block_header(L, out + block_off + (uint64_t)c * kBlockHeaderBytesPerCh);
    uint8_t *up = out + block_off + (uint64_t)kBlockHeaderBytesPerCh * ch + (uint64_t)c * UB;
    for (uint32_t i = kTaps; i < n; i += US, up += unit_stride) {
      uint32_t acc = 0;
#pragma unroll
      for (int k = 0; k < US; k++) { /* samples past n are zero padding - reference :592-593 */
        const int32_t x = i + k < n ? src.at(first + i + k) : 0;
        acc = (acc << BITS) | encode_step<BITS>(L, x, tab, last_qd);
      }
#pragma unroll
      for (int k = 0; k < UB; k++) up[k] = (uint8_t)(acc >> (8 * (UB - 1 - k)));
    }
  }

  if (a.state) {
    LaneStateRecord r;
    r.weight[0] = L.w0; r.weight[1] = L.w1; r.weight[2] = L.w2; r.weight[3] = L.w3;
    r.history[0] = L.h0; r.history[1] = L.h1; r.history[2] = L.h2; r.history[3] = L.h3;
    r.stepsize_index = L.idx;
    r.quantize_error = last_qd;
    a.state[lane] = r;
  }
}

} /* namespace aad */

#endif /* AAD_DEVICE_HIP_H */
