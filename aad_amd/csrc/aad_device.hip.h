/*
 * aad_device.hip.h - CDNA4 (gfx950) device code of the AAD engine.
 *
 * One lane owns one independent recurrence:
 *   encode: lane = (stream, channel) - blocks of a stream are chained through the predictor
 *           state (SURVEY.md finding 2), so a stream is the smallest independent unit;
 *   decode: lane = (block, channel)  - every block header reloads the whole state.
 * Adjacent lanes are the channels of one stream/block, so the bytes they touch are adjacent.
 *
 * What bounds these kernels (measured, tools/microbench, tools/phase_probe.py): every BASELINE
 * workload has far fewer recurrences than the chip has SIMD slots, and a lone gfx950 wave issues
 * one instruction per 4 cycles - one more when it reads the result of the instruction just before
 * it - whatever the instruction (VALU, SALU, LDS; 24- or 32-bit multiplies alike).  The lever is
 * therefore the NUMBER of instruction slots on the per-sample path, s_nop and s_waitcnt included:
 *   - the whole per-channel state (4 weights, 4 history samples, step index) lives in VGPRs,
 *     history rotation is free because samples are processed in unrolled chunks of 16;
 *   - step size, fl32(0.5/step) and fl32(2^(b-1)*0.5/step) sit in LDS, fetched one sample ahead
 *     in hand-pipelined 16-sample chunks;
 *   - the quantiser's integer division is  min(trunc(fma(|d|, hs, hr)), magmax)  - convert, fma
 *     with |.| source modifier, convert, min - proved equal to the reference's division for
 *     every reachable operand (tests/test_quantiser_equiv.py);
 *   - dequantising is a mad + a shift from a 12-byte per-code record (decoder) or one
 *     v_mul_hi_u32 (quad encoder); the encoder's step-index delta is a v_perm_b32 byte lookup;
 *   - predict uses full 32-bit multiplies (exact int32 wraparound for ANY weights), the LMS
 *     products v_mad_i32_i24 (|qd| <= 61438, |h| <= 32768: exact);
 *   - code bytes are read/written in wide unaligned accesses per 16-sample chunk; the stereo
 *     L/R byte interleave is one DPP lane swap plus v_perm_b32 byte permutes;
 *   - with few recurrences the "quad" mapping spreads each one over four lanes (lane t = tap t);
 *     the decoder's step-index walk, which is a scan and not a recurrence, then moves to other
 *     waves altogether (aad_decode_split.hip.h).
 * There is no contraction anywhere, hence no MFMA.
 *
 * Arithmetic widths follow SURVEY.md finding 5: int32 wraparound, arithmetic right shifts.
 */
#ifndef AAD_DEVICE_HIP_H
#define AAD_DEVICE_HIP_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "aad_tables_data.h"

#ifndef AAD_PHASE_TIMING
#define AAD_PHASE_TIMING 0 /* 1: thread 0 of the encode kernel logs s_memtime at phase boundaries (measurement builds only) */
#endif

namespace aad {

#if AAD_PHASE_TIMING
static __device__ uint64_t g_phase_times[512]; /* per translation unit: the reader in aad_hip_engine.hip sees that unit's kernels only */
static __device__ uint32_t g_phase_count;
__device__ __forceinline__ void phase_mark(bool who)
{
  if (who) {
    const uint32_t i = g_phase_count;
    if (i < 512) {
      g_phase_times[i] = __builtin_amdgcn_s_memtime();
      g_phase_count = i + 1;
    }
  }
}
#define AAD_PHASE_MARK(who) phase_mark(who)
#else
#define AAD_PHASE_MARK(who) ((void)0)
#endif

constexpr int kTaps = 4;
constexpr int kBlockHeaderBytesPerCh = 18;
constexpr int kFileHeaderBytes = 31;
constexpr int kChunk = 16; /* samples per unrolled chunk: a multiple of every pack unit (2, 8, 4) and of the tap count */

/*
 * LDS image (one per workgroup, ~3.3 KB).  Dense dword arrays so that a lookup is bank-conflict
 * free unless two lanes hit different entries 32 apart:
 *   step[256]   uint32   step size
 *   hr[256]     float    fl32(0.5 / step)
 *   hs[256]     float    2^(BITS-1) * hr
 *   code[16]    {int32 sm21, int32 bias, int32 delta, -}   decoder: everything a code implies
 *   delta[8]    int16    encoder: index delta by magnitude
 * (A first layout used one 16-byte record per step: only 8 bank groups, SQ_LDS_BANK_CONFLICT was
 * half of all LDS cycles and lookups on the recurrence's critical path took ~100 cycles.)
 * The Q4 step index is kept biased by +8 (idxb), so its table slot is idxb >> 4.
 */
constexpr int kIdxBias = 8;
constexpr int kIdxMin = kIdxBias, kIdxMax = AAD_STEP_INDEX_MAX + kIdxBias;
constexpr int kLdsStepOff = 0, kLdsHrOff = 1024, kLdsHsOff = 2048;
constexpr int kLdsCodeOff = 3072;
constexpr int kLdsDeltaOff = kLdsCodeOff + 16 * 16;
constexpr int kLdsBytes = kLdsDeltaOff + 16;
/* Quad kernels only: the same three values as 16-byte records {step, hr, hs, -}, addressed by
 * idxb & 0xFF0 - one instruction less than slot_addr and one lookup instead of two.  A wave of
 * the quad mapping holds just 16 distinct recurrences, so the 8-bank-group stride that made this
 * layout an 8-way conflict with 64 recurrences per wave is harmless here. */
constexpr int kLdsWideOff = (kLdsBytes + 15) & ~15;
constexpr int kLdsBytesQuad = kLdsWideOff + AAD_STEP_TABLE_LEN * 16;
__device__ __forceinline__ uint32_t wide_addr(int32_t idxb) { return (uint32_t)idxb & 0xFF0u; }

/* byte offset of the step index's slot in the dword arrays */
__device__ __forceinline__ uint32_t slot_addr(int32_t idxb) { return ((uint32_t)idxb >> 2) & 0x3FCu; }

/* internal linkage: the header is compiled into more than one translation unit */
static __constant__ uint16_t c_step_table[AAD_STEP_TABLE_LEN] = {AAD_STEP_TABLE_VALUES};
static __constant__ uint32_t c_half_recip_bits[AAD_STEP_TABLE_LEN] = {AAD_HALF_RECIP_BITS};
static __constant__ int16_t c_delta4[8] = {AAD_INDEX_DELTA_4BIT};
static __constant__ int16_t c_delta3[4] = {AAD_INDEX_DELTA_3BIT};
static __constant__ int16_t c_delta2[2] = {AAD_INDEX_DELTA_2BIT};

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(1))) U32 { uint32_t v; };
struct __attribute__((packed, aligned(1))) U32x2 { u32x2 v; };
struct __attribute__((packed, aligned(1))) U32x3 { u32x3 v; };
struct __attribute__((packed, aligned(1))) U32x4 { u32x4 v; };
struct __attribute__((packed, aligned(1))) U16 { uint16_t v; };

/* mirrors struct AADHipStreamDesc (include/aad_hip.h) */
struct StreamDesc {
  uint64_t pcm_offset;
  uint64_t data_offset;
  uint64_t data_size;
  uint32_t num_samples;
  uint32_t reserved;
};

/* When every stream of a plan has the same length and the offsets form an arithmetic progression
 * (the common batch shape, and the bench's), the kernels derive a lane's stream from its index
 * instead of walking the table: no dependent global loads (a binary search costs ~10 of them,
 * several microseconds for a lone wave) before the first useful byte is fetched. */
struct UniformLayout {
  uint64_t pcm_base, pcm_stride;   /* int16 elements */
  uint64_t data_base, data_stride; /* bytes */
  uint64_t data_size;
  uint32_t num_samples;
  uint32_t blocks_per_stream;      /* decode */
  uint32_t enabled;
  uint32_t reserved;
};

__device__ __forceinline__ StreamDesc uniform_stream(const UniformLayout &u, uint32_t s)
{
  StreamDesc d;
  d.pcm_offset = u.pcm_base + (uint64_t)s * u.pcm_stride;
  d.data_offset = u.data_base + (uint64_t)s * u.data_stride;
  d.data_size = u.data_size;
  d.num_samples = u.num_samples;
  d.reserved = 0;
  return d;
}

/* mirrors struct AADHipLaneState */
struct LaneStateRecord {
  int32_t weight[4];
  int32_t history[4];
  int32_t stepsize_index;
  int32_t quantize_error;
};

struct Lane { /* one lane = one whole recurrence */
  int32_t w0, w1, w2, w3; /* Q15 LMS weights */
  int32_t h0, h1, h2, h3; /* history, h0 newest */
  int32_t idxb;           /* Q4 step index + kIdxBias */
};

/* compile-time unrolled loop: f(std::integral_constant<int, I>) for I in [I0, N) */
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <int BITS>
struct Pack {
  static constexpr int kUnitSamples = BITS == 3 ? 8 : (BITS == 4 ? 2 : 4);
  static constexpr int kUnitBytes = BITS == 3 ? 3 : 1;
  static constexpr int kChunkBytes = kChunk * BITS / 8;                 /* per channel: 8 / 6 / 4 */
  static constexpr int kWords = BITS == 2 ? 1 : 2;                      /* code words per chunk */
  static constexpr int kCodesPerWord = kChunk / kWords;                 /* 8 / 8 / 16 */
  static constexpr uint32_t kSign = 1u << (BITS - 1), kMagMax = kSign - 1u;
  /* bit position of code j (0 = oldest) inside its big-endian code word */
  static constexpr int pos(int j) { return BITS * (kCodesPerWord - 1 - j); }
};

/* stage the tables into LDS; every thread of the workgroup must call this */
template <int BITS, bool QUAD, int WIDE_STEP_SHIFT = 0>
__device__ __forceinline__ void stage_tables(char *lds)
{
  constexpr int kShift = BITS - 1;
  for (int i = threadIdx.x; i < AAD_STEP_TABLE_LEN; i += blockDim.x) {
    const float hr = __uint_as_float(c_half_recip_bits[i]);
    const float hs = hr * (float)(1 << kShift); /* exact power-of-two scaling */
    reinterpret_cast<uint32_t *>(lds + kLdsStepOff)[i] = c_step_table[i];
    reinterpret_cast<float *>(lds + kLdsHrOff)[i] = hr;
    reinterpret_cast<float *>(lds + kLdsHsOff)[i] = hs;
    if (QUAD) {
      u32x4 e;
      e.x = (uint32_t)c_step_table[i] << WIDE_STEP_SHIFT; /* the quad encoder wants 2 * step, see encode_chunk16_quad */
      e.y = __float_as_uint(hr);
      e.z = __float_as_uint(hs);
      e.w = 0;
      *reinterpret_cast<u32x4 *>(lds + kLdsWideOff + (i << 4)) = e;
    }
  }
  const int16_t *dt = BITS == 4 ? c_delta4 : (BITS == 3 ? c_delta3 : c_delta2);
  if (threadIdx.x < (1 << BITS)) {
    /* code -> {signed 2*mag+1, rounding bias, index delta}:
     * qd = (step * sm21 + bias) >> (BITS-1) equals the reference's sign ? -q : q, q = (step * m21) >> (BITS-1) */
    const int code = threadIdx.x, mag = code & ((1 << kShift) - 1), neg = code >> kShift;
    u32x4 e;
    e.x = (uint32_t)(neg ? -(2 * mag + 1) : (2 * mag + 1));
    e.y = neg ? (1u << kShift) - 1u : 0u;
    e.z = (uint32_t)(int32_t)dt[mag];
    e.w = 0;
    *reinterpret_cast<u32x4 *>(lds + kLdsCodeOff + (code << 4)) = e;
  }
  if (threadIdx.x < 8) reinterpret_cast<int16_t *>(lds + kLdsDeltaOff)[threadIdx.x] = dt[threadIdx.x & ((1 << kShift) - 1)];
  __syncthreads();
}

__device__ __forceinline__ int32_t clip16(int32_t v) { return min(max(v, -32768), 32767); }

/* Pin a value's computation between the surrounding scheduling barriers.  sched_barrier only
 * orders instructions with side effects; pure arithmetic is free to sink past it (it did: the
 * compiler ran all sixteen table lookups first and the whole LMS chain afterwards).  A volatile
 * empty asm that "modifies" the value is ordered against the barriers and emits nothing. */
__device__ __forceinline__ void pin(int32_t &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(uint32_t &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(float &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ int32_t clamp_idx(int32_t v) { return min(max(v, kIdxMin), kIdxMax); }

/* a*b + c on the 24-bit multiplier (v_mad_i32_i24).  Both operands MUST fit 24 signed bits; the
 * explicit sign extensions tell the compiler so and cost nothing (the instruction ignores the
 * upper byte).  Native code rather than inline asm: the hazard recogniser pads every asm result
 * with an s_nop, and a wave pays ~4 cycles for each. */
__device__ __forceinline__ int32_t sx24(int32_t v) { return (int32_t)((uint32_t)v << 8) >> 8; }
__device__ __forceinline__ int32_t mad_i24(int32_t a, int32_t b, int32_t c)
{
  return (int32_t)((uint32_t)(sx24(a) * sx24(b)) + (uint32_t)c);
}

/* (16384 + sum h*w) >> 15 with int32 wraparound - reference src/aad_encoder.c:359-363.
 * Full 32-bit multiplies: exact for ANY weights, and on gfx950 v_mul_lo_u32 issues at the same
 * rate as the 24-bit forms (tools/microbench), so a guarded 24-bit fast path buys nothing. */
__device__ __forceinline__ int32_t predict(const Lane &L)
{
  uint32_t acc = 16384u;
  acc += (uint32_t)L.h0 * (uint32_t)L.w0;
  acc += (uint32_t)L.h1 * (uint32_t)L.w1;
  acc += (uint32_t)L.h2 * (uint32_t)L.w2;
  acc += (uint32_t)L.h3 * (uint32_t)L.w3;
  return (int32_t)acc >> 15;
}

/* LMS and history update - reference src/aad_encoder.c:396-406, src/aad_decoder.c:306-315.
 * qd*h fits 32 bits (|qd| <= 61438, |h| <= 32768), so the 24-bit multiplier is exact. */
__device__ __forceinline__ void lms_and_shift(Lane &L, int32_t qd, int32_t y)
{
  L.w0 += mad_i24(qd, L.h0, 16384) >> 18;
  L.w1 += mad_i24(qd, L.h1, 16384) >> 18;
  L.w2 += mad_i24(qd, L.h2, 16384) >> 18;
  L.w3 += mad_i24(qd, L.h3, 16384) >> 18;
  L.h3 = L.h2;
  L.h2 = L.h1;
  L.h1 = L.h0;
  L.h0 = y;
}

/* LMS split in two halves for the hand-pipelined encoder (taps 0-1, then taps 2-3 + shift) */
__device__ __forceinline__ void lms_first(Lane &L, int32_t qd)
{
  L.w0 += mad_i24(qd, L.h0, 16384) >> 18;
  L.w1 += mad_i24(qd, L.h1, 16384) >> 18;
}
__device__ __forceinline__ void lms_rest_and_shift(Lane &L, int32_t qd, int32_t y)
{
  L.w2 += mad_i24(qd, L.h2, 16384) >> 18;
  L.w3 += mad_i24(qd, L.h3, 16384) >> 18;
  L.h3 = L.h2;
  L.h2 = L.h1;
  L.h1 = L.h0;
  L.h0 = y;
}
__device__ __forceinline__ void pin_weights(Lane &L)
{
  pin(L.w0);
  pin(L.w1);
  pin(L.w2);
  pin(L.w3);
}

/*
 * "Quad" mapping: FOUR adjacent lanes share one recurrence, lane t of the quad owning tap t
 * (weight w_t, history sample h_t); everything else is replicated.  The LMS update becomes 3
 * instructions instead of 12, the prediction one multiply-add and two DPP butterfly adds instead
 * of 4 multiplies and 2 three-operand adds, the history shift one DPP move and a select.  A
 * wave pays ~4.7 cycles per instruction whatever it is, and every BASELINE batch leaves most of
 * the chip's lanes idle, so spending 4x the lanes to cut the per-sample instruction count by a
 * third is a pure win there; the host picks this mapping while the batch is small.
 */
template <int CTRL>
__device__ __forceinline__ uint32_t quad_dpp(uint32_t v)
{
  /* old = 0 + bound_ctrl lets the compiler fold the move into the consuming VOP2 (v_add_u32_dpp) */
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}

struct QuadLane {
  int32_t w, h;    /* this lane's tap */
  int32_t idxb;    /* replicated */
  uint32_t round;  /* 16384 in tap 0, 0 elsewhere: the prediction's rounding term enters the sum once */
  uint32_t newest; /* all ones in tap 0, 0 elsewhere: where the reconstructed sample enters the history */
  bool tap0;
};

__device__ __forceinline__ int32_t predict(const QuadLane &Q)
{
  uint32_t s = (uint32_t)Q.h * (uint32_t)Q.w + Q.round;
  s += quad_dpp<0xB1>(s); /* quad_perm [1,0,3,2] */
  s += quad_dpp<0x4E>(s); /* quad_perm [2,3,0,1] */
  return (int32_t)s >> 15;
}
__device__ __forceinline__ void lms_first(QuadLane &Q, int32_t qd) { Q.w += mad_i24(qd, Q.h, 16384) >> 18; }
/* History shift.  SELECT: one v_mov_b32_dpp + v_cndmask on a lane mask the compiler keeps in VCC
 * (decoder: shortest path from the new sample to the next product).  Otherwise a bit-select on a
 * per-lane VGPR mask (v_and_b32_dpp + v_and_or_b32): the encoder bodies have no spare VCC, there
 * the select's mask ended up in an SGPR pair in some instantiations (the trial-search kernels),
 * and those ran every chunk ~24 % slower than the very same code with the mask in VCC. */
template <bool SELECT = false>
__device__ __forceinline__ void lms_rest_and_shift(QuadLane &Q, int32_t, int32_t y)
{
  const uint32_t up = quad_dpp<0x90>((uint32_t)Q.h); /* quad_perm [0,0,1,2]: the next-older tap's sample */
  if (SELECT) Q.h = Q.tap0 ? y : (int32_t)up;
  else Q.h = (int32_t)(((uint32_t)y & Q.newest) | (up & ~Q.newest));
}
template <bool SELECT = false>
__device__ __forceinline__ void lms_and_shift(QuadLane &Q, int32_t qd, int32_t y)
{
  lms_first(Q, qd);
  lms_rest_and_shift<SELECT>(Q, qd, y);
}
__device__ __forceinline__ void pin_weights(QuadLane &Q) { pin(Q.w); }

/* full state <-> quad (block boundaries only) */
template <bool BITMASK = true>
__device__ __forceinline__ QuadLane to_quad(const Lane &L, uint32_t tap)
{
  QuadLane Q;
  Q.w = tap == 0 ? L.w0 : (tap == 1 ? L.w1 : (tap == 2 ? L.w2 : L.w3));
  Q.h = tap == 0 ? L.h0 : (tap == 1 ? L.h1 : (tap == 2 ? L.h2 : L.h3));
  Q.idxb = L.idxb;
  Q.round = tap == 0 ? 16384u : 0u;
  Q.tap0 = tap == 0;
  Q.newest = tap == 0 ? 0xFFFFFFFFu : 0u;
  if (BITMASK) pin(Q.newest); /* opaque: keeps the bit-select from being turned back into a v_cndmask */
  return Q;
}
__device__ __forceinline__ Lane from_quad(const QuadLane &Q)
{
  Lane L;
  L.w0 = (int32_t)quad_dpp<0x00>((uint32_t)Q.w);
  L.w1 = (int32_t)quad_dpp<0x55>((uint32_t)Q.w);
  L.w2 = (int32_t)quad_dpp<0xAA>((uint32_t)Q.w);
  L.w3 = (int32_t)quad_dpp<0xFF>((uint32_t)Q.w);
  L.h0 = (int32_t)quad_dpp<0x00>((uint32_t)Q.h);
  L.h1 = (int32_t)quad_dpp<0x55>((uint32_t)Q.h);
  L.h2 = (int32_t)quad_dpp<0xAA>((uint32_t)Q.h);
  L.h3 = (int32_t)quad_dpp<0xFF>((uint32_t)Q.h);
  L.idxb = Q.idxb;
  return L;
}

/* Q4 step-index delta of a magnitude code as arithmetic (the constants of reference
 * src/aad_tables.c:8-45): 4-bit {-18,-17,-14,16,32,64,128,256}, 3-bit {-16,-15,32,128},
 * 2-bit {-14,40}.  Five instructions instead of an LDS lookup: used where that lookup would sit
 * on the recurrence's critical path and there are instruction slots to spare (quad encoder). */
template <int BITS>
__device__ __forceinline__ int32_t index_delta_arith(uint32_t mag)
{
  if (BITS == 4) {
    /* 2 << mag, minus {20, 21, 22, 0, 0, 0, 0, 0}[mag] picked by one v_perm_b32 byte lookup
     * (selector bytes 1-3 = 0x0c give zero): no compare/select pair, no SGPR hazard */
    const uint32_t corr = __builtin_amdgcn_perm(0u, 0x00161514u, mag | 0x0c0c0c00u);
    return (int32_t)(2u << mag) - (int32_t)corr;
  } else if (BITS == 3) {
    return mag < 2 ? (int32_t)mag - 16 : (int32_t)(2u << (2u * mag));
  } else {
    return mag ? 40 : -14;
  }
}

/* one encoder step - reference src/aad_encoder.c:343-410.  Returns the code; qd is the
 * dequantised difference (the reference's quantize_error).  Plain form, used for tails and the
 * trial search; the bulk goes through encode_chunk16.  S = Lane or QuadLane. */
template <int BITS, typename S>
__device__ __forceinline__ uint32_t encode_step(S &L, int32_t x, const char *lds, int32_t &qd)
{
  const uint32_t sa = slot_addr(L.idxb);
  const uint32_t step = *reinterpret_cast<const uint32_t *>(lds + kLdsStepOff + sa);
  const float hr = *reinterpret_cast<const float *>(lds + kLdsHrOff + sa);
  const float hs = *reinterpret_cast<const float *>(lds + kLdsHsOff + sa);
  const int32_t p = predict(L);
  const int32_t d = x - p;
  const int32_t m = d >> 31; /* 0 or -1 */
  /* min((|d| << (BITS-2)) / step, magmax) == min(trunc(fma(|d|, 2^(BITS-1)*hr, hr)), magmax), hr = fl32(0.5/step) */
  const uint32_t mag = min((uint32_t)__builtin_fmaf(__builtin_fabsf((float)d), hs, hr), Pack<BITS>::kMagMax);
  const uint32_t m21 = (mag << 1) | 1u;
  const int32_t q = (int32_t)(__umul24(step, m21) >> (BITS - 1));
  qd = (q ^ m) - m;
  /* m21 = 2*mag + 1 addresses the int16 delta table: byte offset 2*mag = m21 - 1 */
  const int32_t delta = *reinterpret_cast<const int16_t *>(lds + (kLdsDeltaOff - 1) + m21);
  L.idxb = clamp_idx(L.idxb + delta);
  lms_and_shift(L, qd, clip16(qd + p));
  return mag | ((uint32_t)m & Pack<BITS>::kSign);
}

/* one decoder step - reference src/aad_decoder.c:269-318; `code` in the low BITS bits.  Plain
 * form for tails; the bulk goes through decode_chunk16. */
template <int BITS, typename S>
__device__ __forceinline__ int32_t decode_step(S &L, uint32_t code, const char *lds)
{
  const uint32_t step = *reinterpret_cast<const uint32_t *>(lds + kLdsStepOff + slot_addr(L.idxb));
  const u32x3 t = *reinterpret_cast<const u32x3 *>(lds + kLdsCodeOff + ((code & ((1u << BITS) - 1u)) << 4));
  const int32_t qd = mad_i24((int32_t)step, (int32_t)t.x, (int32_t)t.y) >> (BITS - 1);
  const int32_t y = clip16(qd + predict(L));
  L.idxb = clamp_idx(L.idxb + (int32_t)t.z);
  lms_and_shift(L, qd, y);
  return y;
}

/*
 * Sixteen encoder steps, software-pipelined by hand.  The encoder's recurrence runs through two
 * dependent LDS lookups per sample (code -> index delta -> step record of the NEXT sample), ~55
 * cycles each for a lone wave.  The compiler's schedule waits for both right after issuing
 * them; here every sample is cut into four regions separated by scheduling barriers so that
 * each lookup has ~15 independent instructions (~60 cycles) issued behind it:
 *   A  quantise with the step record fetched during the previous sample; start the delta lookup
 *   B  dequantise, reconstruct, LMS taps 0-1, pack the code            (hides the delta lookup)
 *   C  new step index; start the lookup of the next sample's step record
 *   D  LMS taps 2-3, history shift, predict + difference of the NEXT sample (hides the record lookup)
 * Same arithmetic as encode_step, instruction for instruction.
 */
/* squared dequantised difference as the reference accumulates it (src/aad_encoder.c:461): the
 * product wraps in int32 before it is widened (SURVEY.md finding 5) */
__device__ __forceinline__ int64_t wrapped_square(int32_t qd) { return (int64_t)(int32_t)((uint32_t)qd * (uint32_t)qd); }

/* EMIT: pack the codes into w[] (the real encode pass); otherwise add the wrapped squares of the
 * dequantised differences to sq (an RMSE pass of the trial search - same recurrence, no output) */
/* PACKED: x holds eight dwords of two int16 samples each instead of sixteen widened values (the
 * subtract then reads the halves directly, v_sub_u32_sdwa) */
template <int BITS, bool EMIT, bool PACKED = false, typename S>
__device__ __forceinline__ void encode_chunk16(S &L, const int32_t *x, const char *lds, uint32_t *w, int32_t &qd_out, int64_t &sq)
{
  auto sample = [&](int k) -> int32_t { /* k compile-time after unrolling */
    if (PACKED) return (k & 1) ? x[k >> 1] >> 16 : (int32_t)(int16_t)x[k >> 1];
    return x[k];
  };
  uint32_t sa = slot_addr(L.idxb);
  uint32_t step = *reinterpret_cast<const uint32_t *>(lds + kLdsStepOff + sa);
  float hr = *reinterpret_cast<const float *>(lds + kLdsHrOff + sa);
  float hs = *reinterpret_cast<const float *>(lds + kLdsHsOff + sa);
  int32_t p = predict(L);
  int32_t d = sample(0) - p;
  int32_t m = d >> 31;
  float f = (float)d;
  static_for<0, kChunk>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    /* A */
    const uint32_t mag = min((uint32_t)__builtin_fmaf(__builtin_fabsf(f), hs, hr), Pack<BITS>::kMagMax);
    const uint32_t m21 = (mag << 1) | 1u;
    const int32_t delta = *reinterpret_cast<const int16_t *>(lds + (kLdsDeltaOff - 1) + m21);
    __builtin_amdgcn_sched_barrier(0);
    /* B */
    const int32_t q = (int32_t)(__umul24(step, m21) >> (BITS - 1));
    const int32_t qd = (q ^ m) - m;
    const int32_t y = clip16(qd + p);
    lms_first(L, qd);
    if (EMIT) {
      uint32_t &acc = w[j / Pack<BITS>::kCodesPerWord];
      uint32_t code = ((uint32_t)m & Pack<BITS>::kSign) | mag; /* v_and_or_b32 */
      pin(code);
      acc = (acc << BITS) | code;                               /* v_lshl_or_b32 */
      pin(acc);
    } else {
      sq += wrapped_square(qd);
    }
    pin_weights(L);
    __builtin_amdgcn_sched_barrier(0);
    /* C */
    L.idxb = clamp_idx(L.idxb + delta);
    if (j + 1 < kChunk) {
      sa = slot_addr(L.idxb);
      step = *reinterpret_cast<const uint32_t *>(lds + kLdsStepOff + sa);
      hr = *reinterpret_cast<const float *>(lds + kLdsHrOff + sa);
      hs = *reinterpret_cast<const float *>(lds + kLdsHsOff + sa);
    }
    __builtin_amdgcn_sched_barrier(0);
    /* D */
    lms_rest_and_shift(L, qd, y);
    if (j + 1 < kChunk) {
      p = predict(L);
      d = sample(j + 1) - p;
      m = d >> 31;
      f = (float)d;
      pin(m);
      pin(f);
    } else {
      qd_out = qd;
      pin_weights(L);
    }
    __builtin_amdgcn_sched_barrier(0);
  });
}

/*
 * Sixteen encoder steps for the quad mapping.  With the LMS and the prediction down to a few
 * instructions the two dependent LDS lookups of encode_chunk16 would bound the sample (~190
 * cycles); here the index delta is arithmetic, so the only lookup on the recurrence is the next
 * step record, started right after the quantiser and hidden behind everything else:
 *   A  quantise, delta, new step index; start the lookup of the next sample's step record
 *   B  dequantise, reconstruct, LMS, history shift, pack the code, predict + difference of the next sample
 */
struct EncodeCarry {
  u32x3 e;         /* {step, hr, hs} of the coming sample */
  int32_t p, d, m; /* its prediction, difference and sign mask */
  float f;         /* (float)d */
};

template <int BITS>
__device__ __forceinline__ void encode_prime_quad(QuadLane &L, EncodeCarry &C, int32_t x0, const char *lds)
{
  C.e = *reinterpret_cast<const u32x3 *>(lds + kLdsWideOff + wide_addr(L.idxb));
  C.p = predict(L);
  C.d = x0 - C.p;
  C.m = C.d >> 31;
  C.f = (float)C.d;
}

/* x: this chunk's 16 samples, xn0: the first sample of the next chunk (the pipeline is carried
 * from chunk to chunk like the decoder's, see DecodeCarry) */
/* PACKED: x holds the chunk as eight dwords of two int16 samples each (and xn0 the next chunk's
 * first such dword) instead of sixteen sign-extended values */
template <int BITS, bool EMIT, bool PACKED = false>
__device__ __forceinline__ void encode_chunk16_quad(QuadLane &L, EncodeCarry &C, const int32_t *x, int32_t xn0,
                                                    const char *lds, uint32_t *w, int32_t &qd_out, int64_t &sq)
{
  auto sample = [&](int k) -> int32_t { /* sample k of this chunk (k = 16: first of the next), k compile-time */
    if (PACKED) {
      const int32_t word = k < kChunk ? x[k >> 1] : xn0;
      return (k & 1) ? word >> 16 : (int32_t)(int16_t)word;
    }
    return k < kChunk ? x[k] : xn0;
  };
  u32x3 e = C.e;
  int32_t p = C.p, d = C.d, m = C.m;
  float f = C.f;
  static_for<0, kChunk>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    /* A */
    const uint32_t mag = min((uint32_t)__builtin_fmaf(__builtin_fabsf(f), __uint_as_float(e.z), __uint_as_float(e.y)),
                             Pack<BITS>::kMagMax);
    const uint32_t step2_j = e.x; /* 2 * step (stage_tables<.., WIDE_STEP_SHIFT = 1>) */
    L.idxb = clamp_idx(L.idxb + index_delta_arith<BITS>(mag));
    e = *reinterpret_cast<const u32x3 *>(lds + kLdsWideOff + wide_addr(L.idxb));
    __builtin_amdgcn_sched_barrier(0);
    /* B: q = (step * (2 mag + 1)) >> (BITS-1) as ONE high multiply:
     * (2 step) * ((2 mag + 1) << (32 - BITS)) = step * (2 mag + 1) * 2^(33 - BITS), upper 32 bits.
     * 2 step < 2^16 and (2 mag + 1) < 2^BITS, so neither factor overflows and the result is exact. */
    const uint32_t m21s = (mag << (33 - BITS)) | (1u << (32 - BITS));
    const int32_t q = (int32_t)__umulhi(step2_j, m21s);
    const int32_t qd = (q ^ m) - m;
    const int32_t y = clip16(qd + p);
    lms_and_shift(L, qd, y);
    /* prediction of the next sample, with the two instructions that pack this sample's code
     * placed in the wait states its DPP adds need (see decode_chunk16_quad) */
    uint32_t s = (uint32_t)L.h * (uint32_t)L.w + L.round;
    pin(s);
    uint32_t code = 0, sqw = 0;
    if (EMIT) {
      code = ((uint32_t)m & Pack<BITS>::kSign) | mag; /* v_and_or_b32 */
      pin(code);
    } else {
      sqw = (uint32_t)qd * (uint32_t)qd;
      pin(sqw);
    }
    s += quad_dpp<0xB1>(s);
    pin(s);
    if (EMIT) {
      uint32_t &acc = w[j / Pack<BITS>::kCodesPerWord];
      acc = (acc << BITS) | code; /* v_lshl_or_b32 */
      pin(acc);
    } else {
      sq += (int64_t)(int32_t)sqw;
    }
    s += quad_dpp<0x4E>(s);
    p = (int32_t)s >> 15;
    d = sample(j + 1) - p;
    m = d >> 31;
    f = (float)d;
    pin(m);
    pin(f);
    if (j + 1 == kChunk) qd_out = qd;
    __builtin_amdgcn_sched_barrier(0);
  });
  C.e = e;
  C.p = p;
  C.d = d;
  C.m = m;
  C.f = f;
}

/*
 * Sixteen decoder steps, software-pipelined by hand.  Codes are known a chunk ahead, so the
 * per-code record of sample j+2 is fetched during sample j; the only lookup on the recurrence
 * is the step size of the next sample, started as soon as the new index is known and hidden
 * behind this sample's reconstruction, LMS update and the next prediction (~22 instructions).
 *   A  step index of the next sample; start its step lookup and the record lookup of sample j+2
 *   B  dequantise (mad + shift), reconstruct, LMS, history shift, predict the next sample
 */
template <int BITS, typename S, typename Finish>
__device__ __forceinline__ void decode_chunk16(S &L, const uint32_t *w, const char *lds, int32_t *y, Finish finish)
{
  constexpr int cpw = Pack<BITS>::kCodesPerWord;
  auto code_addr = [&](int j) -> uint32_t { /* (code << 4) for sample j, j compile-time after unrolling */
    const int pos = Pack<BITS>::pos(j % cpw);
    const uint32_t word = w[j / cpw];
    return (pos >= 4 ? word >> (pos >= 4 ? pos - 4 : 0) : word << (4 - pos)) & (((1u << BITS) - 1u) << 4);
  };
  uint32_t step = *reinterpret_cast<const uint32_t *>(lds + kLdsStepOff + slot_addr(L.idxb));
  /* per-code records two samples ahead: they depend on nothing but the code bits */
  u32x3 t0 = *reinterpret_cast<const u32x3 *>(lds + kLdsCodeOff + code_addr(0));
  u32x3 t1 = *reinterpret_cast<const u32x3 *>(lds + kLdsCodeOff + code_addr(1));
  int32_t p = predict(L);
  static_for<0, kChunk>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    /* A */
    const uint32_t step_j = step;
    const u32x3 t_j = t0;
    t0 = t1;
    L.idxb = clamp_idx(L.idxb + (int32_t)t_j.z);
    if (j + 1 < kChunk) step = *reinterpret_cast<const uint32_t *>(lds + kLdsStepOff + slot_addr(L.idxb));
    if (j + 2 < kChunk) t1 = *reinterpret_cast<const u32x3 *>(lds + kLdsCodeOff + code_addr(j + 2 < kChunk ? j + 2 : j));
    __builtin_amdgcn_sched_barrier(0);
    /* B */
    const int32_t qd = mad_i24((int32_t)step_j, (int32_t)t_j.x, (int32_t)t_j.y) >> (BITS - 1);
    const int32_t yy = clip16(qd + p);
    lms_and_shift(L, qd, yy);
    if (j + 1 < kChunk) {
      p = predict(L);
      pin(p);
    } else {
      pin_weights(L);
    }
    y[j] = finish(yy);
    __builtin_amdgcn_sched_barrier(0);
  });
}

/*
 * Sixteen decoder steps for the quad mapping.  The prediction's two DPP butterfly adds each need
 * wait states after the instruction that wrote their operand (the compiler pads with s_nop, ~4-8
 * cycles apiece for a lone wave).  The decoder's step-index chain depends only on the codes, so
 * it is run one sample further ahead than in decode_chunk16 and its instructions are placed
 * exactly in those gaps: index update + record lookup after the product, slot address + step
 * lookup after the first butterfly add.  In flight per lane: the step sizes of samples j+1 and
 * j+2 and the per-code records of samples j+1 .. j+3.
 *
 * The pipeline is carried from chunk to chunk (DecodeCarry) instead of being re-primed every 16
 * samples - a re-prime costs two exposed LDS round trips and a DPP reduction, ~170 cycles.  For
 * that the last three samples of a chunk look their records up in the NEXT chunk's code words
 * (wn), which the kernel unpacks one chunk early.
 */
struct DecodeCarry {
  uint32_t step0, step1; /* step sizes of samples j, j+1 */
  u32x3 t0, t1, t2;      /* per-code records of samples j, j+1, j+2 */
  int32_t p;             /* prediction for sample j */
  int32_t idx_next;      /* step index (biased) of the first sample after the chunk just finished */
};

template <int BITS>
__device__ __forceinline__ uint32_t chunk_code_addr(const uint32_t *w, const uint32_t *wn, int j)
{
  constexpr int cpw = Pack<BITS>::kCodesPerWord;
  const uint32_t word = j < kChunk ? w[j / cpw] : wn[(j - kChunk) / cpw];
  const int pos = Pack<BITS>::pos((j % kChunk) % cpw);
  return (pos >= 4 ? word >> (pos >= 4 ? pos - 4 : 0) : word << (4 - pos)) & (((1u << BITS) - 1u) << 4);
}

template <int BITS>
__device__ __forceinline__ void decode_prime_quad(QuadLane &L, DecodeCarry &C, const uint32_t *w, const char *lds)
{
  auto record = [&](int j) { return *reinterpret_cast<const u32x3 *>(lds + kLdsCodeOff + chunk_code_addr<BITS>(w, w, j)); };
  auto step_at = [&](int32_t idxb) { return *reinterpret_cast<const uint32_t *>(lds + kLdsWideOff + wide_addr(idxb)); };
  C.step0 = step_at(L.idxb);
  C.t0 = record(0);
  C.t1 = record(1);
  C.t2 = record(2);
  L.idxb = clamp_idx(L.idxb + (int32_t)C.t0.z); /* from here on L.idxb runs one sample ahead */
  C.step1 = step_at(L.idxb);
  C.p = predict(L);
  C.idx_next = L.idxb;
}

template <int BITS, typename Finish>
__device__ __forceinline__ void decode_chunk16_quad(QuadLane &L, DecodeCarry &C, const uint32_t *w, const uint32_t *wn,
                                                    const char *lds, int32_t *y, Finish finish)
{
  auto record = [&](int j) { return *reinterpret_cast<const u32x3 *>(lds + kLdsCodeOff + chunk_code_addr<BITS>(w, wn, j)); };
  auto step_at = [&](int32_t idxb) { return *reinterpret_cast<const uint32_t *>(lds + kLdsWideOff + wide_addr(idxb)); };
  uint32_t step0 = C.step0, step1 = C.step1;
  u32x3 t0 = C.t0, t1 = C.t1, t2 = C.t2;
  int32_t p = C.p;
  static_for<0, kChunk>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const int32_t qd = mad_i24((int32_t)step0, (int32_t)t0.x, (int32_t)t0.y) >> (BITS - 1);
    const int32_t yy = clip16(qd + p);
    lms_and_shift<true>(L, qd, yy);
    y[j] = finish(yy);
    uint32_t s = (uint32_t)L.h * (uint32_t)L.w + L.round;
    pin(s);
    /* gap 1: index of sample j+2 and the record lookup of sample j+3.  The record is started
     * BEFORE the step lookup below: LDS results return in order, so the wait for a step size at
     * the top of a sample also covers the record whose delta is needed in the middle of the one
     * before - one s_waitcnt per sample, not two. */
    int32_t idx2 = clamp_idx(L.idxb + (int32_t)t1.z);
    const u32x3 t3 = record(j + 3);
    pin(idx2);
    s += quad_dpp<0xB1>(s);
    pin(s);
    /* gap 2: start the step lookup that hangs on the new index */
    const uint32_t step2 = step_at(idx2);
    s += quad_dpp<0x4E>(s);
    p = (int32_t)s >> 15;
    pin(p);
    if (j == kChunk - 2) C.idx_next = idx2; /* index of sample 16: what a non-pipelined continuation needs */
    step0 = step1;
    step1 = step2;
    t0 = t1;
    t1 = t2;
    t2 = t3;
    L.idxb = idx2;
    __builtin_amdgcn_sched_barrier(0);
  });
  C.step0 = step0;
  C.step1 = step1;
  C.t0 = t0;
  C.t1 = t1;
  C.t2 = t2;
  C.p = p;
}

/* ---- per-lane byte shuffles ------------------------------------------------------------- */

/* v_perm_b32: selector bytes 0-3 pick from `lo`, 4-7 from `hi`, 0x0c yields 0x00 */
__device__ __forceinline__ uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

/* value of the neighbouring lane (lane ^ 1): the other channel of a stereo pair */
template <bool QUAD>
__device__ __forceinline__ uint32_t pair_swap(uint32_t v, uint32_t c)
{
  if (QUAD) {
    /* lane ^ 4: the same tap of the other channel's quad.  Two DPP row shifts and a select
     * (~13 cycles) rather than ds_swizzle (an LDS round trip, ~64 cycles, on the store path) */
    const uint32_t from_hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x104 /* row_shl:4 */, 0xF, 0xF, true);
    const uint32_t from_lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114 /* row_shr:4 */, 0xF, 0xF, true);
    return c ? from_lo : from_hi;
  }
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, false);
}

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
  UniformLayout uni;
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

/* Code words of one 16-sample chunk of channel c.  CHF = 1 or 2 channels (fast paths: one wide
 * unaligned load, per-lane v_perm selectors pull the lane's own bytes out of the L/R interleave
 * and turn them big-endian).  `p` points at the first byte of the chunk's first unit of channel 0. */
template <int BITS, int CHF>
struct ChunkCodes {
  uint32_t r[4]; /* raw dwords as loaded */
  /* bytes the wide load touches, measured from p */
  static constexpr int kLoadBytes = CHF == 1 ? (BITS == 2 ? 4 : 8) : (BITS == 4 ? 16 : (BITS == 3 ? 12 : 8));
  static constexpr int kRaw = kLoadBytes / 4;
  __device__ __forceinline__ void load(const uint8_t *p)
  {
    if (kRaw == 1) {
      r[0] = reinterpret_cast<const U32 *>(p)->v;
    } else if (kRaw == 2) {
      const u32x2 d = reinterpret_cast<const U32x2 *>(p)->v;
      r[0] = d.x; r[1] = d.y;
    } else if (kRaw == 3) {
      const u32x3 d = reinterpret_cast<const U32x3 *>(p)->v;
      r[0] = d.x; r[1] = d.y; r[2] = d.z;
    } else {
      const u32x4 d = reinterpret_cast<const U32x4 *>(p)->v;
      r[0] = d.x; r[1] = d.y; r[2] = d.z; r[3] = d.w;
    }
  }
  /* Claim the loaded registers without emitting an instruction: the compiler has to place the
   * s_waitcnt for the prefetch HERE (a whole chunk of arithmetic after it was issued) instead of
   * at the top of the next iteration behind a burst of fresh stores - gfx950 has one vmcnt for
   * loads and stores, so a wait placed after the stores would also wait for every one of them. */
  __device__ __forceinline__ void touch()
  {
    if (kRaw == 1) asm volatile("" : "+v"(r[0]) :: "memory");
    if (kRaw == 2) asm volatile("" : "+v"(r[0]), "+v"(r[1]) :: "memory");
    if (kRaw == 3) asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]) :: "memory");
    if (kRaw == 4) asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) :: "memory");
  }
  /* big-endian code words of channel c */
  __device__ __forceinline__ void unpack(uint32_t c, uint32_t *w) const
  {
    if (CHF == 1) {
      if (BITS == 2) {
        w[0] = perm(0, r[0], 0x00010203);
      } else if (BITS == 4) {
        w[0] = perm(0, r[0], 0x00010203);
        w[1] = perm(0, r[1], 0x00010203);
      } else { /* two 3-byte units */
        w[0] = perm(r[1], r[0], 0x0c000102);
        w[1] = perm(r[1], r[0], 0x0c030405);
      }
    } else {
      if (BITS == 4) { /* L R L R ...: own bytes c, c+2 of every dword */
        const uint32_t sel = 0x00020406u + c * 0x01010101u;
        w[0] = perm(r[1], r[0], sel);
        w[1] = perm(r[3], r[2], sel);
      } else if (BITS == 2) {
        w[0] = perm(r[1], r[0], 0x00020406u + c * 0x01010101u);
      } else { /* L3 R3 L3 R3 */
        w[0] = perm(r[1], r[0], c ? 0x0c030405u : 0x0c000102u);
        w[1] = perm(r[2], r[1], c ? 0x0c050607u : 0x0c020304u);
      }
    }
  }
};

/* Write 16 decoded samples of channel c (y[], int16 range) as interleaved PCM.  Mono: two 16-byte
 * stores.  Stereo: the two lanes of a pair trade half of their packed samples through DPP and
 * each writes 2 x 16 contiguous bytes of L/R frames.  A vector-memory instruction costs a lone
 * wave ~17 cycles to issue whatever its width, so few wide stores beat one short per sample. */
template <int CHF, bool QUAD>
__device__ __forceinline__ void store_chunk_pcm(int16_t *frame0, const int32_t *y, uint32_t c, uint32_t ch)
{
  if (CHF == 1) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
      u32x4 v;
      v.x = perm((uint32_t)y[8 * h + 1], (uint32_t)y[8 * h + 0], 0x05040100);
      v.y = perm((uint32_t)y[8 * h + 3], (uint32_t)y[8 * h + 2], 0x05040100);
      v.z = perm((uint32_t)y[8 * h + 5], (uint32_t)y[8 * h + 4], 0x05040100);
      v.w = perm((uint32_t)y[8 * h + 7], (uint32_t)y[8 * h + 6], 0x05040100);
      reinterpret_cast<U32x4 *>(frame0 + 8 * h)->v = v;
    }
  } else if (CHF == 2) {
    /* per 8 samples: lane 0 writes frames 0-3 (own samples 0-3 + partner's), lane 1 frames 4-7 */
    const uint32_t sel_lo = c ? 0x05040100u : 0x01000504u, sel_hi = c ? 0x07060302u : 0x03020706u;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const uint32_t p0 = perm((uint32_t)y[8 * h + 1], (uint32_t)y[8 * h + 0], 0x05040100);
      const uint32_t p1 = perm((uint32_t)y[8 * h + 3], (uint32_t)y[8 * h + 2], 0x05040100);
      const uint32_t p2 = perm((uint32_t)y[8 * h + 5], (uint32_t)y[8 * h + 4], 0x05040100);
      const uint32_t p3 = perm((uint32_t)y[8 * h + 7], (uint32_t)y[8 * h + 6], 0x05040100);
      const uint32_t ra = pair_swap<QUAD>(c ? p0 : p2, c), rb = pair_swap<QUAD>(c ? p1 : p3, c);
      const uint32_t ka = c ? p2 : p0, kb = c ? p3 : p1;
      u32x4 v;
      v.x = perm(ka, ra, sel_lo);
      v.y = perm(ka, ra, sel_hi);
      v.z = perm(kb, rb, sel_lo);
      v.w = perm(kb, rb, sel_hi);
      reinterpret_cast<U32x4 *>(frame0 + 16 * h + 8 * c)->v = v;
    }
  } else {
#pragma unroll
    for (int j = 0; j < kChunk; j++) frame0[(uint32_t)j * ch + c] = (int16_t)y[j];
  }
}

/*
 * Block-parallel decode (reference src/aad_decoder.c:321-475, looped by :514-534).
 * CHF: 1 / 2 = specialised channel counts with wide chunk loads, 0 = any channel count (byte loads).
 */
template <int BITS, int CHF, bool MS, bool QUAD>
__global__ void __launch_bounds__(256) decode_blocks_kernel(DecodeArgs a)
{
  static_assert(!QUAD || CHF != 0, "the quad mapping exists for the mono / stereo fast paths");
  __shared__ __attribute__((aligned(16))) char lds[QUAD ? kLdsBytesQuad : kLdsBytes];
  AAD_PHASE_MARK(blockIdx.x == 0 && threadIdx.x == 0);
  stage_tables<BITS, QUAD>(lds);
  AAD_PHASE_MARK(blockIdx.x == 0 && threadIdx.x == 0);

  const uint32_t ch = CHF ? CHF : a.channels;
  const uint64_t thread = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t lane = QUAD ? thread >> 2 : thread; /* index of the (block, channel) recurrence */
  const uint32_t tap = QUAD ? threadIdx.x & 3u : 0u;
  const bool writer = tap == 0;                      /* quad: all four lanes hold the samples, one stores them */
  const bool active = lane < a.total_blocks * ch;
  const uint64_t g = active ? lane / ch : 0;
  const uint32_t c = active ? (uint32_t)(lane % ch) : 0;

  uint32_t s;
  StreamDesc sd;
  uint64_t b;
  if (a.uni.enabled) { /* wave-uniform branch */
    s = (uint32_t)g / a.uni.blocks_per_stream; /* the host only enables this below 2^32 blocks */
    b = (uint32_t)g - s * a.uni.blocks_per_stream;
    sd = uniform_stream(a.uni, s);
  } else {
    s = find_stream(a.block_prefix, a.num_streams, g);
    sd = a.streams[s];
    b = g - a.block_prefix[s];
  }
  const uint64_t first = b * a.samples_per_block;
  uint32_t n = 0;
  if (active && first < sd.num_samples) {
    const uint64_t left = sd.num_samples - first;
    n = left < a.samples_per_block ? (uint32_t)left : a.samples_per_block;
  }
  /* bytes of this stream still present from the start of this block */
  const uint64_t block_off = a.header_bytes + b * a.block_size;
  const uint64_t avail64 = sd.data_size > block_off ? sd.data_size - block_off : 0;
  const uint32_t avail = avail64 > 0x7FFFFFFFu ? 0x7FFFFFFFu : (uint32_t)avail64;
  const uint8_t *src = a.data + sd.data_offset + block_off;
  int16_t *dst = a.pcm + sd.pcm_offset + first * ch + c;
  if (avail < (uint32_t)kBlockHeaderBytesPerCh * ch) n = 0; /* DecodeBlock: INSUFFICIENT_DATA (reported by the host) */

  Lane H = {0, 0, 0, 0, 0, 0, 0, 0, kIdxBias};
  if (n) { /* block header - reference src/aad_decoder.c:364-380 */
    const uint8_t *hp = src + c * kBlockHeaderBytesPerCh;
    const uint32_t v = load_be16(hp);
    H.idxb = min((int32_t)(v >> 4), (int32_t)AAD_STEP_INDEX_MAX) + kIdxBias;
    const uint32_t shift = v & 0xFu;
    H.w0 = (int32_t)((uint32_t)(int32_t)(int16_t)load_be16(hp + 2) << shift);
    H.h0 = (int16_t)load_be16(hp + 4);
    H.w1 = (int32_t)((uint32_t)(int32_t)(int16_t)load_be16(hp + 6) << shift);
    H.h1 = (int16_t)load_be16(hp + 8);
    H.w2 = (int32_t)((uint32_t)(int32_t)(int16_t)load_be16(hp + 10) << shift);
    H.h2 = (int16_t)load_be16(hp + 12);
    H.w3 = (int32_t)((uint32_t)(int32_t)(int16_t)load_be16(hp + 14) << shift);
    H.h3 = (int16_t)load_be16(hp + 16);
  }

  /* inverse mid/side needs the partner channel's sample: the lanes of channels 0/1 of a block
   * are neighbours (4 apart in the quad mapping) and run the same trip counts, so the swap
   * always meets an active lane */
  auto finish = [&](int32_t y) -> int32_t {
    if (MS) {
      const int32_t other = (int32_t)pair_swap<QUAD>((uint32_t)y, c);
      return c == 0 ? clip16(y + other) : clip16(other - y);
    }
    return y;
  };

  /* the first four samples are stored verbatim in the header - reference :386-391 */
  {
    const int32_t y0 = finish(H.h3), y1 = finish(H.h2), y2 = finish(H.h1), y3 = finish(H.h0);
    if (writer) {
      if (n > 0) dst[0] = (int16_t)y0;
      if (n > 1) dst[ch] = (int16_t)y1;
      if (n > 2) dst[2 * ch] = (int16_t)y2;
      if (n > 3) dst[3 * ch] = (int16_t)y3;
    }
  }
  using S = std::conditional_t<QUAD, QuadLane, Lane>;
  S L;
  if constexpr (QUAD) L = to_quad<false>(H, tap); else L = H;
  AAD_PHASE_MARK(blockIdx.x == 0 && threadIdx.x == 0);

  constexpr int US = Pack<BITS>::kUnitSamples, UB = Pack<BITS>::kUnitBytes;
  const uint32_t coded = n > (uint32_t)kTaps ? n - kTaps : 0;
  uint32_t done = 0; /* coded samples finished */

  if (CHF != 0) {
    /* full 16-sample chunks whose wide load stays inside the stream's bytes */
    using CC = ChunkCodes<BITS, (CHF ? CHF : 1)>;
    constexpr uint32_t kStride = Pack<BITS>::kChunkBytes * (CHF ? CHF : 1);
    const uint32_t body = (uint32_t)kBlockHeaderBytesPerCh * ch;
    uint32_t full = coded / kChunk;
    if (avail < body + CC::kLoadBytes) {
      full = 0;
    } else {
      const uint32_t fit = (avail - body - CC::kLoadBytes) / kStride + 1;
      full = full < fit ? full : fit;
    }
    const uint8_t *cp = src + body;
    int16_t *op = a.pcm + sd.pcm_offset + (first + kTaps) * ch; /* frame of this chunk's first sample, channel 0 */
    CC next;
    next.r[0] = next.r[1] = next.r[2] = next.r[3] = 0;
    if (full) next.load(cp);
    next.touch();
    if constexpr (QUAD) {
      /* pipeline carried across chunks: the code words of chunk k+1 are unpacked one chunk early */
      uint32_t w[2] = {0, 0}, wn[2] = {0, 0};
      DecodeCarry C;
      if (full) {
        next.unpack(c, w);
        if (full > 1) cp += kStride;
        next.load(cp);
        next.touch();
        next.unpack(c, wn);
        decode_prime_quad<BITS>(L, C, w, lds);
      }
      for (uint32_t k = 0; k < full; k++) {
        /* prefetch chunk k+2 (clamped to the last full chunk), consumed after this chunk's arithmetic */
        if (k + 2 < full) cp += kStride;
        next.load(cp);
        int32_t y[kChunk];
        decode_chunk16_quad<BITS>(L, C, w, wn, lds, y, finish);
        next.touch();
        w[0] = wn[0];
        w[1] = wn[1];
        next.unpack(c, wn);
        if (writer) store_chunk_pcm<CHF, QUAD>(op, y, c, ch);
        op += (uint64_t)kChunk * ch;
      }
      if (full) L.idxb = C.idx_next; /* drop the run-ahead: the tail below is not pipelined */
    } else {
      for (uint32_t k = 0; k < full; k++) {
        uint32_t w[2] = {0, 0};
        next.unpack(c, w);
        /* prefetch the next chunk (the last iteration re-reads its own: an unconditional load lands
         * straight in `next`'s registers, a conditional one would be copied - and waited for - at once);
         * it is consumed (touch) only after this chunk's arithmetic */
        if (k + 1 < full) cp += kStride;
        next.load(cp);
        int32_t y[kChunk];
        decode_chunk16<BITS>(L, w, lds, y, finish);
        next.touch();
        if (writer) store_chunk_pcm<CHF, QUAD>(op, y, c, ch);
        op += (uint64_t)kChunk * ch;
      }
    }
    done = full * kChunk;
  }
  AAD_PHASE_MARK(blockIdx.x == 0 && threadIdx.x == 0);

  /* remaining units (all of them when CHF == 0): byte loads, bytes past the stream read as zero */
  {
    const uint32_t unit_stride = UB * ch;
    const uint32_t base = (uint32_t)kBlockHeaderBytesPerCh * ch + c * UB;
    for (uint32_t i = done; i < coded; i += US) {
      const uint32_t o = base + (i / US) * unit_stride;
      uint32_t acc = 0;
#pragma unroll
      for (int k = 0; k < UB; k++) acc = (acc << 8) | (o + k < avail ? (uint32_t)src[o + k] : 0u);
      acc <<= 32 - 8 * UB; /* codes to the top of the word */
#pragma unroll
      for (int k = 0; k < US; k++) {
        const int32_t y = finish(decode_step<BITS>(L, acc >> (32 - BITS), lds));
        acc <<= BITS;
        if (writer && i + k < coded) dst[(uint64_t)(kTaps + i + k) * ch] = (int16_t)y;
      }
    }
  }
  AAD_PHASE_MARK(blockIdx.x == 0 && threadIdx.x == 0);
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
  UniformLayout uni;
  uint8_t header_template[32]; /* 31-byte file header with num_samples = 0 */
};

/* sample i of channel c of a stream, after the optional L/R -> M/S transform
 * (reference src/aad_encoder.c:413-428; the clip there can never trigger for int16 input) */
template <bool MS>
struct SampleSource {
  const int16_t *x;
  uint32_t ch, c;
  __device__ __forceinline__ int32_t at(uint64_t i) const
  {
    if (MS) {
      const int32_t l = x[i * 2], r = x[i * 2 + 1];
      return c == 0 ? (l + r) >> 1 : (l - r) >> 1;
    }
    return x[i * ch + c];
  }
};

/* 16 consecutive samples of channel c starting at frame `first`, fetched with wide loads.
 * CHF = 1: 32 contiguous bytes; CHF = 2: 64 bytes of L/R frames (both lanes of the pair read
 * the same bytes and keep their own half); CHF = 0: 16 strided int16 loads. */
template <int CHF, bool MS>
struct ChunkSamples {
  uint32_t d[CHF == 1 ? 8 : 16];
  __device__ __forceinline__ void load(const int16_t *x, uint32_t ch, uint32_t c)
  {
    if (CHF == 1) {
      const u32x4 a = reinterpret_cast<const U32x4 *>(x)->v, b = reinterpret_cast<const U32x4 *>(x + 8)->v;
      d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
    } else if (CHF == 2) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const u32x4 a = reinterpret_cast<const U32x4 *>(x + 8 * k)->v;
        d[4 * k] = a.x; d[4 * k + 1] = a.y; d[4 * k + 2] = a.z; d[4 * k + 3] = a.w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 16; k++) d[k] = (uint32_t)(int32_t)x[(uint32_t)k * ch + c];
    }
  }
  /* see ChunkCodes::touch */
  __device__ __forceinline__ void touch()
  {
    constexpr int n = CHF == 1 ? 8 : 16;
    asm volatile("" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) :: "memory");
    if (n == 16)
      asm volatile("" : "+v"(d[n - 8]), "+v"(d[n - 7]), "+v"(d[n - 6]), "+v"(d[n - 5]), "+v"(d[n - 4]), "+v"(d[n - 3]),
                        "+v"(d[n - 2]), "+v"(d[n - 1]) :: "memory");
  }
  /* samples 2k and 2k+1 of channel c as one dword (low half first): what the quad encoder keeps
   * per chunk - its subtract reads the halves directly (SDWA), no per-sample extraction */
  __device__ __forceinline__ uint32_t pair(int k, uint32_t sel) const
  {
    if (CHF == 1) return d[k];
    return __builtin_amdgcn_perm(d[2 * k + 1], d[2 * k], sel); /* sel = c ? 0x07060302 : 0x05040100 */
  }
  __device__ __forceinline__ int32_t get(int j, uint32_t c) const
  {
    if (CHF == 1) return __builtin_amdgcn_sbfe((int32_t)d[j >> 1], (j & 1) * 16, 16);
    if (CHF == 2) {
      if (MS) {
        const int32_t l = __builtin_amdgcn_sbfe((int32_t)d[j], 0, 16), r = (int32_t)d[j] >> 16;
        return c == 0 ? (l + r) >> 1 : (l - r) >> 1;
      }
      return __builtin_amdgcn_sbfe((int32_t)d[j], c * 16, 16);
    }
    return (int32_t)d[j];
  }
};

template <typename Src>
__device__ __forceinline__ void seed_history(Lane &L, const Src &src, uint64_t first, uint32_t n)
{
  L.h3 = n > 0 ? src.at(first + 0) : 0;
  L.h2 = n > 1 ? src.at(first + 1) : 0;
  L.h1 = n > 2 ? src.at(first + 2) : 0;
  L.h0 = n > 3 ? src.at(first + 3) : 0;
}
template <typename Src>
__device__ __forceinline__ void seed_history(QuadLane &Q, const Src &src, uint64_t first, uint32_t n, uint32_t tap)
{
  const uint32_t k = 3u - tap; /* tap t holds the sample that is t steps old: h_t = x[3 - t] */
  Q.h = k < n ? src.at(first + k) : 0;
}

__device__ __forceinline__ void store_be16(uint8_t *p, uint32_t v)
{
  p[0] = (uint8_t)(v >> 8);
  p[1] = (uint8_t)v;
}

/* block header of one channel - reference src/aad_encoder.c:619-655.  Drops the weight bits the
 * 16-bit header fields cannot carry from the lane's own state as well. */
__device__ __forceinline__ void write_block_header(Lane &L, uint8_t *p, bool do_store)
{
  auto wabs = [](int32_t w) { const int32_t m = w >> 31; return (int32_t)(((uint32_t)w ^ (uint32_t)m) - (uint32_t)m); };
  const int32_t maxabs = max(max(max(wabs(L.w0), wabs(L.w1)), max(wabs(L.w2), wabs(L.w3))), 0);
  const int32_t shift = max(17 - (int32_t)__clz(maxabs), 0); /* smallest shift with maxabs >> shift <= 32767 */
  const int32_t mask = (int32_t)~((1u << shift) - 1u);
  L.w0 &= mask;
  L.w1 &= mask;
  L.w2 &= mask;
  L.w3 &= mask;
  if (!do_store) return;
  store_be16(p, ((((uint32_t)(L.idxb - kIdxBias)) << 4) & 0xFFFFu) | ((uint32_t)shift & 0xFu));
  store_be16(p + 2, (uint32_t)(L.w0 >> shift));
  store_be16(p + 4, (uint32_t)L.h0);
  store_be16(p + 6, (uint32_t)(L.w1 >> shift));
  store_be16(p + 8, (uint32_t)L.h1);
  store_be16(p + 10, (uint32_t)(L.w2 >> shift));
  store_be16(p + 12, (uint32_t)L.h2);
  store_be16(p + 14, (uint32_t)(L.w3 >> shift));
  store_be16(p + 16, (uint32_t)L.h3);
}

/* Write the packed codes of one 16-sample chunk.  w[]: big-endian code words of this lane's
 * channel.  up: first byte of the chunk's first unit of channel 0.  For stereo the two lanes of
 * a pair trade one word through DPP and each writes half of the interleaved bytes. */
template <int BITS, int CHF, bool QUAD>
__device__ __forceinline__ void store_chunk_codes(uint8_t *up, const uint32_t *w, uint32_t c)
{
  if (CHF == 1) {
    if (BITS == 4) {
      u32x2 v;
      v.x = perm(0, w[0], 0x00010203);
      v.y = perm(0, w[1], 0x00010203);
      reinterpret_cast<U32x2 *>(up)->v = v;
    } else if (BITS == 2) {
      reinterpret_cast<U32 *>(up)->v = perm(0, w[0], 0x00010203);
    } else { /* a0 a1 a2 a3 | a4 a5 : w0 = 0 a0 a1 a2, w1 = 0 a3 a4 a5 */
      reinterpret_cast<U32 *>(up)->v = perm(w[1], w[0], 0x06000102);
      reinterpret_cast<U16 *>(up + 4)->v = (uint16_t)perm(0, w[1], 0x0c0c0001);
    }
  } else { /* stereo */
    if (BITS == 2) { /* out: a0 b0 a1 b1 | a2 b2 a3 b3 ; lane c writes dword c */
      const uint32_t other = pair_swap<QUAD>(w[0], c);
      const uint32_t A = c ? other : w[0], B = c ? w[0] : other; /* A = channel 0 word, B = channel 1 word */
      reinterpret_cast<U32 *>(up + 4 * c)->v = perm(A, B, c ? 0x00040105u : 0x02060307u);
    } else {
      /* lane 0 writes the first half (needs word 0 of both channels), lane 1 the second half */
      const uint32_t send = c ? w[0] : w[1], keep = c ? w[1] : w[0];
      const uint32_t recv = pair_swap<QUAD>(send, c);
      const uint32_t A = c ? recv : keep, B = c ? keep : recv;
      if (BITS == 4) { /* a0 b0 a1 b1 | a2 b2 a3 b3 from A = a0 a1 a2 a3, B = b0 b1 b2 b3 (big-endian words) */
        u32x2 v;
        v.x = perm(A, B, 0x02060307);
        v.y = perm(A, B, 0x00040105);
        reinterpret_cast<U32x2 *>(up + 8 * c)->v = v;
      } else { /* a0 a1 a2 b0 | b1 b2 from A = 0 a0 a1 a2, B = 0 b0 b1 b2 */
        reinterpret_cast<U32 *>(up + 6 * c)->v = perm(A, B, 0x02040506);
        reinterpret_cast<U16 *>(up + 6 * c + 4)->v = (uint16_t)perm(A, B, 0x0c0c0001);
      }
    }
  }
}

/*
 * One pass of the recurrence over the coded samples of a block: samples [first+4, first+n) of
 * channel c, history already seeded.  EMIT = the real encode pass (codes packed and stored under
 * `body`); otherwise an RMSE pass of the trial search (reference src/aad_encoder.c:431-467): the
 * same arithmetic, the int32-wrapped squares of the dequantised differences summed instead of any
 * output (every partial sum is an exact integer < 2^53, so an int64 sum converted once equals
 * the reference's running double).  Full 16-sample chunks go through the hand-pipelined bodies
 * with wide prefetched loads, the rest through encode_step.
 */
template <int BITS, int CHF, bool MS, bool QUAD, bool EMIT, typename S>
__device__ __forceinline__ int64_t run_block(S &L, const SampleSource<MS> &src, uint64_t first, uint32_t n, uint32_t ch,
                                             uint32_t c, bool writer, uint8_t *body, const char *lds, int32_t &last_qd)
{
  constexpr int US = Pack<BITS>::kUnitSamples, UB = Pack<BITS>::kUnitBytes;
  const uint32_t unit_stride = UB * ch;
  const uint32_t coded = n > (uint32_t)kTaps ? n - kTaps : 0;
  int64_t sq = 0;
  uint32_t done = 0;
  {
    using CS = ChunkSamples<CHF, MS>;
    const uint32_t full = coded / kChunk;
    const int16_t *xp = src.x + (first + kTaps) * ch;
    constexpr uint32_t kOutStride = Pack<BITS>::kChunkBytes;
    CS next;
    for (auto &v : next.d) v = 0;
    if (full) next.load(xp, ch, c);
    next.touch();
    if constexpr (QUAD) {
      /* pipeline carried across chunks: chunk k+1's samples are extracted one chunk early.
       * Without M/S they stay packed two to a dword (kN = 8 registers per chunk); the M/S
       * transform needs them widened (kN = 16). */
      constexpr bool PK = !MS;
      constexpr int kN = PK ? kChunk / 2 : kChunk;
      const uint32_t pair_sel = c ? 0x07060302u : 0x05040100u;
      auto extract = [&](int32_t(&dst)[kN]) {
#pragma unroll
        for (int j = 0; j < kN; j++) dst[j] = PK ? (int32_t)next.pair(j, pair_sel) : next.get(j, c);
      };
      auto first_sample = [&](const int32_t(&buf)[kN]) -> int32_t { return PK ? (int32_t)(int16_t)buf[0] : buf[0]; };
      int32_t x[kN], xn[kN];
      EncodeCarry C;
      if (full) {
        extract(x);
        if (full > 1) xp += (uint64_t)kChunk * ch;
        next.load(xp, ch, c);
        next.touch();
        extract(xn);
        encode_prime_quad<BITS>(L, C, first_sample(x), lds);
      }
      /* x and xn swap roles every chunk (the loop is unrolled by two) so that the samples of
       * chunk k+2 are extracted straight into the buffer chunk k has just freed - rotating the
       * buffers with moves cost 30 instructions per chunk */
      auto one = [&](uint32_t k, int32_t(&cur)[kN], const int32_t(&ahead)[kN]) {
        if (k + 2 < full) xp += (uint64_t)kChunk * ch; /* prefetch chunk k+2 (clamped to the last full one) */
        next.load(xp, ch, c);
        uint32_t w[2] = {0, 0};
        encode_chunk16_quad<BITS, EMIT, PK>(L, C, cur, ahead[0], lds, w, last_qd, sq);
        next.touch();
        extract(cur);
        if (EMIT && writer) store_chunk_codes<BITS, (CHF ? CHF : 1), QUAD>(body + (uint64_t)k * kOutStride * ch, w, c);
      };
      for (uint32_t k = 0; k < full; k += 2) {
        one(k, x, xn);
        if (k + 1 < full) one(k + 1, xn, x);
      }
    } else {
      /* mono / stereo without M/S: the samples stay packed two to a dword (see encode_chunk16) */
      constexpr bool PK = CHF != 0 && !MS;
      constexpr int kN = PK ? kChunk / 2 : kChunk;
      const uint32_t pair_sel = c ? 0x07060302u : 0x05040100u;
      for (uint32_t k = 0; k < full; k++) {
        int32_t x[kN];
#pragma unroll
        for (int j = 0; j < kN; j++) x[j] = PK ? (int32_t)next.pair(j, pair_sel) : next.get(j, c);
        /* unconditional prefetch (the last iteration re-reads its own chunk), see the decoder */
        if (k + 1 < full) xp += (uint64_t)kChunk * ch;
        next.load(xp, ch, c);
        uint32_t w[2] = {0, 0};
        encode_chunk16<BITS, EMIT, PK>(L, x, lds, w, last_qd, sq);
        next.touch();
        if (EMIT) {
          if (CHF != 0) {
            store_chunk_codes<BITS, (CHF ? CHF : 1), false>(body + (uint64_t)k * kOutStride * ch, w, c);
          } else { /* any channel count: this lane's unit bytes one by one */
            uint8_t *up = body + (uint64_t)k * kOutStride * ch + (uint64_t)c * UB;
#pragma unroll
            for (int u = 0; u < kChunk / US; u++) {
              const int per_word = Pack<BITS>::kCodesPerWord / US; /* units per code word */
              const uint32_t word = w[u / per_word];
              const uint32_t unit = word >> (8 * UB * (per_word - 1 - (u % per_word)));
#pragma unroll
              for (int q = 0; q < UB; q++) up[(uint64_t)u * unit_stride + q] = (uint8_t)(unit >> (8 * (UB - 1 - q)));
            }
          }
        }
      }
    }
    done = full * kChunk;
  }

  if (EMIT) { /* tail units: samples past n are zero padding - reference :592-593 */
    uint8_t *up = body + (uint64_t)(done / US) * unit_stride + (uint64_t)c * UB;
    for (uint32_t i = done; i < coded; i += US, up += unit_stride) {
      uint32_t acc = 0;
#pragma unroll
      for (int k = 0; k < US; k++) {
        const int32_t x = i + k < coded ? src.at(first + kTaps + i + k) : 0;
        acc = (acc << BITS) | encode_step<BITS>(L, x, lds, last_qd);
      }
      if (writer) {
#pragma unroll
        for (int k = 0; k < UB; k++) up[k] = (uint8_t)(acc >> (8 * (UB - 1 - k)));
      }
    }
  } else { /* an RMSE pass stops at the last real sample (reference :457) */
    for (uint32_t i = done; i < coded; i++) {
      int32_t qd;
      encode_step<BITS>(L, src.at(first + kTaps + i), lds, qd);
      sq += wrapped_square(qd);
    }
  }
  return sq;
}

/* RMSE of the dequantised differences over one block while the lane adapts - reference
 * src/aad_encoder.c:431-467 (divisor = the block length, sum over the coded samples only) */
template <int BITS, int CHF, bool MS, bool QUAD, typename S>
__device__ __forceinline__ double rmse_pass(S &L, const SampleSource<MS> &src, uint64_t first, uint32_t n, uint32_t ch,
                                            uint32_t c, uint32_t tap, const char *lds)
{
  if (n < (uint32_t)kTaps) return 0.0;
  if constexpr (QUAD) seed_history(L, src, first, n, tap); else seed_history(L, src, first, n);
  int32_t qd_unused = 0;
  const int64_t sum = run_block<BITS, CHF, MS, QUAD, false>(L, src, first, n, ch, c, false, nullptr, lds, qd_unused);
  return sqrt((double)sum / (double)n);
}

/* trial search - reference src/aad_encoder.c:470-562 (per channel; channels are independent) */
template <int BITS, int CHF, bool MS, bool QUAD, typename S>
__device__ __forceinline__ void search_best_lane(S &L, const SampleSource<MS> &src, uint64_t first, uint32_t n, uint32_t spb,
                                                 uint32_t trials, uint32_t ch, uint32_t c, uint32_t tap, const char *lds)
{
  const bool have_prev = first >= spb;
  S best = L, run = L;
  double best_rmse = 0.0;
  /* One call site for every pass (the pipelined chunk bodies exist once in the kernel):
   * pass 0 is the probe, then per trial [previous block,] current block.  Without a previous
   * block the probe and trial 0 are the same computation from the same state - equal RMSE, no
   * strict improvement - so it is run once (first block of every stream: 1 + t passes, not 2 + t). */
  const uint32_t per_trial = have_prev ? 2u : 1u;
  const uint32_t passes = have_prev ? 1u + 2u * trials : trials;
  for (uint32_t p = 0; p < passes; p++) {
    const bool is_probe = have_prev && p == 0;
    const bool on_prev = have_prev && p != 0 && ((p - 1u) % per_trial) == 0;
    S from = is_probe ? L : run;
    const S before = from;
    const double r = rmse_pass<BITS, CHF, MS, QUAD>(from, src, on_prev ? first - spb : first, on_prev ? spb : n, ch, c, tap, lds);
    if (!is_probe) run = from;
    AAD_PHASE_MARK(blockIdx.x == 0 && threadIdx.x == 0);
    if (is_probe || (!have_prev && p == 0)) {
      best_rmse = r;
    } else if (!on_prev && best_rmse > r) {
      best_rmse = r;
      best = before;
    }
  }
  L = best;
}

/*
 * The same search with its two independent strands on different lanes ("dual" mapping, used
 * with the quad mapping, i.e. when lanes are idle anyway).  The reference evaluates
 *   probe:  RMSE of the current block from the carried state                      (1 pass)
 *   chain:  trials x ([previous block] + current block), each from where the last ended
 * one after the other: 2 + 2t passes per block with the final encode.  The probe does not feed
 * the chain, so a second group of four lanes (role 1) runs it while role 0 runs the chain's
 * first pass; role 0 then finishes the chain alone, picks the winner exactly as the reference
 * does (strict >, probe first) and encodes: 1 + 2t passes of latency.  One call site for every
 * pass keeps the pipelined chunk bodies in the kernel once.
 */
template <int BITS, int CHF, bool MS, typename S>
__device__ __forceinline__ void search_best_lane_dual(S &L, const SampleSource<MS> &src, uint64_t first, uint32_t n, uint32_t spb,
                                                      uint32_t trials, uint32_t ch, uint32_t c, uint32_t tap, uint32_t role,
                                                      const char *lds)
{
  const bool have_prev = first >= spb;
  const uint32_t chain_passes = trials * (have_prev ? 2u : 1u);
  S best = L, run = L;
  double best_rmse = 0.0;
  for (uint32_t p = 0; p < chain_passes; p++) {
    const bool on_prev = role == 0 && have_prev && (p & 1u) == 0;
    const bool active = role == 0 || p == 0;
    double r = 0.0;
    S before = run;
    if (active) r = rmse_pass<BITS, CHF, MS, true>(run, src, on_prev ? first - spb : first, on_prev ? spb : n, ch, c, tap, lds);
    AAD_PHASE_MARK(blockIdx.x == 0 && threadIdx.x == 0);
    if (p == 0) { /* the probe's figure moves over to the chain's lanes (role 1 sits CHF quads above role 0) */
      const int from = (int)((threadIdx.x & 63u) + (role == 0 ? 4u * (CHF ? CHF : 1) : 0u));
      best_rmse = __shfl(r, from, 64);
    }
    if (role == 0 && !on_prev && best_rmse > r) {
      best_rmse = r;
      best = before;
    }
  }
  L = best; /* role 1 is handed role 0's state again after the block's encode pass */
}

/*
 * Stream-parallel encode (reference src/aad_encoder.c:814-891 with EncodeBlock :565-727 and the
 * optional trial search :470-562 inlined).  lane = (stream, channel).
 */
template <int BITS, int CHF, bool MS, bool QUAD, bool TRIALS, bool DUAL = false>
__global__ void __launch_bounds__(256) encode_streams_kernel(EncodeArgs a)
{
  static_assert(!QUAD || CHF != 0, "the quad mapping exists for the mono / stereo fast paths");
  static_assert(!DUAL || (QUAD && TRIALS), "the dual mapping is the trial search on the quad mapping");
  __shared__ __attribute__((aligned(16))) char lds[QUAD ? kLdsBytesQuad : kLdsBytes];
  AAD_PHASE_MARK(blockIdx.x == 0 && threadIdx.x == 0);
  stage_tables<BITS, QUAD, 1>(lds);
  AAD_PHASE_MARK(blockIdx.x == 0 && threadIdx.x == 0);

  const uint32_t ch = CHF ? CHF : a.channels;
  const uint64_t thread = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  /* dual: with trials on the quad mapping every stream owns 2 x CHF quads, laid out
   * [role 0: ch 0 .. CHF-1][role 1: ch 0 .. CHF-1] so that a stereo pair stays 4 lanes apart */
  constexpr uint32_t kQuadsPerStream = DUAL ? 2u * (CHF ? CHF : 1) : 1u;
  const uint32_t role = DUAL ? (uint32_t)((thread >> 2) % kQuadsPerStream) / (CHF ? CHF : 1) : 0u;
  const uint64_t lane = DUAL ? (thread >> 2) / kQuadsPerStream * (CHF ? CHF : 1) + (thread >> 2) % (CHF ? CHF : 1)
                             : (QUAD ? thread >> 2 : thread); /* index of the (stream, channel) recurrence */
  const uint32_t tap = QUAD ? threadIdx.x & 3u : 0u;
  const bool writer = tap == 0 && role == 0;         /* quad: all four lanes hold the codes, one stores them */
  if (lane >= (uint64_t)a.num_streams * ch) return; /* whole quads / stereo pairs / role groups leave together */
  const uint32_t s = (uint32_t)(lane / ch), c = (uint32_t)(lane % ch);
  const StreamDesc sd = a.uni.enabled ? uniform_stream(a.uni, s) : a.streams[s];
  const SampleSource<MS> src = {a.pcm + sd.pcm_offset, ch, c};
  uint8_t *out = a.data + sd.data_offset;
  const uint32_t total = sd.num_samples, spb = a.samples_per_block;

  /* F: the complete per-channel state, the form block headers and the state records need;
   * between block boundaries the quad mapping spreads it over four lanes (S) */
  using S = std::conditional_t<QUAD, QuadLane, Lane>;
  Lane F = {0, 0, 0, 0, 0, 0, 0, 0, kIdxBias};
  int32_t last_qd = 0;
  if (a.state) {
    const LaneStateRecord r = a.state[lane];
    F = {r.weight[0], r.weight[1], r.weight[2], r.weight[3],
         r.history[0], r.history[1], r.history[2], r.history[3],
         min(max(r.stepsize_index, 0), (int32_t)AAD_STEP_INDEX_MAX) + kIdxBias};
    last_qd = r.quantize_error;
  }

  if (c == 0 && writer) { /* file header - reference src/aad_encoder.c:190-214 */
    for (int i = 0; i < kFileHeaderBytes; i++) out[i] = a.header_template[i];
    out[14] = (uint8_t)(total >> 24);
    out[15] = (uint8_t)(total >> 16);
    out[16] = (uint8_t)(total >> 8);
    out[17] = (uint8_t)total;
  }

  uint64_t block_off = kFileHeaderBytes;
  for (uint64_t first = 0; first < total; first += spb, block_off += a.block_size) {
    const uint32_t n = total - first < spb ? (uint32_t)(total - first) : spb;
    S L;
    AAD_PHASE_MARK(blockIdx.x == 0 && threadIdx.x == 0);
    if constexpr (TRIALS) { /* reference src/aad_encoder.c:863-871; a separate instantiation so that the
                             * trial-free kernel does not carry the search's registers */
      if constexpr (QUAD) L = to_quad(F, tap); else L = F;
      if constexpr (DUAL) search_best_lane_dual<BITS, CHF, MS>(L, src, first, n, spb, a.trials, ch, c, tap, role, lds);
      else search_best_lane<BITS, CHF, MS, QUAD>(L, src, first, n, spb, a.trials, ch, c, tap, lds);
      if constexpr (QUAD) F = from_quad(L); else F = L;
    }
    seed_history(F, src, first, n);
    write_block_header(F, out + block_off + (uint64_t)c * kBlockHeaderBytesPerCh, writer);
    if constexpr (QUAD) L = to_quad(F, tap); else L = F;
    uint8_t *body = out + block_off + (uint64_t)kBlockHeaderBytesPerCh * ch;
    AAD_PHASE_MARK(blockIdx.x == 0 && threadIdx.x == 0);
    if (role == 0) {
      (void)run_block<BITS, CHF, MS, QUAD, true>(L, src, first, n, ch, c, writer, body, lds, last_qd);
      if constexpr (QUAD) F = from_quad(L); else F = L;
    }
    AAD_PHASE_MARK(blockIdx.x == 0 && threadIdx.x == 0);
    if constexpr (DUAL) { /* the probe lanes start the next block from the encoder's state */
      const int from = (int)((threadIdx.x & 63u) - role * 4u * (CHF ? CHF : 1));
      F.w0 = __shfl(F.w0, from, 64); F.w1 = __shfl(F.w1, from, 64); F.w2 = __shfl(F.w2, from, 64); F.w3 = __shfl(F.w3, from, 64);
      F.h0 = __shfl(F.h0, from, 64); F.h1 = __shfl(F.h1, from, 64); F.h2 = __shfl(F.h2, from, 64); F.h3 = __shfl(F.h3, from, 64);
      F.idxb = __shfl(F.idxb, from, 64);
      last_qd = __shfl(last_qd, from, 64);
    }
  }

  if (a.state && writer) {
    LaneStateRecord r;
    r.weight[0] = F.w0; r.weight[1] = F.w1; r.weight[2] = F.w2; r.weight[3] = F.w3;
    r.history[0] = F.h0; r.history[1] = F.h1; r.history[2] = F.h2; r.history[3] = F.h3;
    r.stepsize_index = F.idxb - kIdxBias;
    r.quantize_error = last_qd;
    a.state[lane] = r;
  }
}

} /* namespace aad */

#endif /* AAD_DEVICE_HIP_H */
