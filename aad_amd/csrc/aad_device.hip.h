/*
 * aad_device.hip.h - CDNA4 (gfx950) device code of the AAD engine.
 *
 * One lane owns one independent recurrence:
 *   encode: lane = (stream, channel) - blocks of a stream are chained through the predictor
 *           state (SURVEY.md finding 2), so a stream is the smallest independent unit;
 *   decode: lane = (block, channel)  - every block header reloads the whole state.
 * Adjacent lanes are the channels of one stream/block, so the bytes they touch are adjacent.
 *
 * What bounds these kernels (measured, tools/microbench, tools/phase_probe.py,
 * tools/experiments): every BASELINE workload has far fewer recurrences than the chip has SIMD
 * slots, and a lone gfx950 wave issues one instruction per ~4.6 cycles whatever the instruction
 * (VALU, SALU, DPP, LDS, s_nop, s_waitcnt; 24- or 32-bit multiplies alike) and whatever the
 * order.  The lever is therefore the NUMBER of instruction slots on the per-sample path:
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
 *   code[16]    {int32 sm21, int16 delta | bias << 16}   decoder: everything a code implies, 8 bytes -
 *               sixteen records fill exactly one row of the 32 LDS banks, so a ds_read_b64 in which
 *               every lane asks for a different code is conflict-free (the 16-byte {sm21, bias,
 *               delta, -} records of round 1 put codes c and c + 8 - same magnitude, other sign - on
 *               the same banks: 3.8 conflict cycles per lookup in the dense decoder)
 *   delta[8]    int16    encoder: index delta by magnitude
 * (A first layout used one 16-byte record per step: only 8 bank groups, SQ_LDS_BANK_CONFLICT was
 * half of all LDS cycles and lookups on the recurrence's critical path took ~100 cycles.)
 * The Q4 step index is kept biased by +8 (idxb), so its table slot is idxb >> 4.
 */
constexpr int kIdxBias = 8;
constexpr int kIdxMin = kIdxBias, kIdxMax = AAD_STEP_INDEX_MAX + kIdxBias;
/* Largest step index a BLOCK HEADER may carry and still mean something in the reference: the 12-bit field is taken as it is
 * (src/aad_decoder.c:365-366, no clamp), the first sample's step is T[(idx + 8) >> 4] (src/aad_tables.h:15,28) - slot 255 for
 * 4081..4087 as for 4080 - and the index walk goes on from the UNCLAMPED value (clamp(idx + delta, 0, 4080), :31-43), so a
 * header index of 4087 followed by a delta of -18 gives 4069, not 4062.  4088..4095 make the reference read past its
 * 256-entry table (undefined there); they are taken as 4087 here. */
constexpr int kHeaderIdxMax = AAD_STEP_INDEX_MAX + 7;
constexpr int kLdsStepOff = 0, kLdsHrOff = 1024, kLdsHsOff = 2048;
constexpr int kLdsCodeOff = 3072;
constexpr int kLdsCodeShift = 3; /* log2 of the record size */
constexpr int kLdsDeltaOff = kLdsCodeOff + (16 << kLdsCodeShift);
constexpr int kLdsDeltaScaledOff = kLdsDeltaOff + 16; /* the same eight deltas times kIdxScale (encoders' scaled step index) */
constexpr int kLdsBytes = kLdsDeltaScaledOff + 16;
/* Quad kernels only: the same three values as 16-byte records {step, hr, hs, -}, addressed by
 * idxb & 0xFF0 - one instruction less than slot_addr and one lookup instead of two.  A wave of
 * the quad mapping holds just 16 distinct recurrences, so the 8-bank-group stride that made this
 * layout an 8-way conflict with 64 recurrences per wave is harmless here. */
constexpr int kLdsWideOff = (kLdsBytes + 15) & ~15;
constexpr int kLdsBytesQuad = kLdsWideOff + AAD_STEP_TABLE_LEN * 16;
__device__ __forceinline__ uint32_t wide_addr(int32_t idxb) { return (uint32_t)idxb & 0xFF0u; }
/* ENCODERS: kWideCopies copies of the wide records, interleaved - slot i, copy k at 16 (kWideCopies i + k).
 * A ds_read_b96 is served eight lanes per LDS cycle, and the eight are NOT neighbours: the groups are
 * {0-3, 20-23}, {4-7, 16-19}, {8-11, 28-31}, {12-15, 24-27} and the same in the upper half of the wave
 * (MI355X_MICROARCH.md, LDS table), over 32 banks.  A group is therefore four lanes of an even 16-lane row
 * and four of an odd one - in the tap-major quad layout two taps of EIGHT different recurrences, in the
 * dense mapping eight recurrences outright - and every one of the eight may ask for another slot.  Round 2
 * kept four copies (copy = lane & 3) at a 64-byte pitch on the belief that a group was eight adjacent
 * lanes: the two lanes that share a copy collide whenever their slots agree mod 2 - SQ_LDS_BANK_CONFLICT
 * 7.1 cycles per lookup in the headline kernel (profiles/r02_pmc_summary_bench.txt), on the one lookup
 * that sits on the recurrence.  With eight copies at a 128-byte pitch, copy = (lane & 3) | row parity << 2,
 * the lanes of a group own banks 4 copy .. 4 copy + 2 whatever their slots: no two reads of a cycle meet.
 * The encoders keep the step index scaled by kIdxScale = kWideCopies for this (J = kIdxScale * idxb; the
 * slot is J >> 7 and the address (J & 0x7F80) | 16 copy: one v_and_or_b32, where the unscaled form took one
 * v_and_b32); the index deltas come scaled as well.  AAD_WIDE_COPIES=4 rebuilds round 2's layout (A/B). */
#ifndef AAD_WIDE_COPIES
#define AAD_WIDE_COPIES 8
#endif
constexpr int kWideCopies = AAD_WIDE_COPIES;
static_assert(kWideCopies == 4 || kWideCopies == 8, "four (round 2) or eight copies of the encoders' step records");
constexpr int kIdxScale = kWideCopies;
constexpr int kIdxScaleLog2 = kWideCopies == 8 ? 3 : 2;
constexpr int kLdsBytesQuadEnc = kLdsWideOff + AAD_STEP_TABLE_LEN * 16 * kWideCopies;
/* Dense DECODER: step << 2 in four copies per 16-byte slot (copy = lane & 3 spreads a wave's lookups
 * over all banks; the slot's address is idxb & 0xFF0, one v_and_or_b32 with the copy offset) and
 * 16-byte per-code records {bias << 29 | delta & 0xFFFF, 0, sm21 << 27, -} for the one-instruction
 * dequantiser (dense_dequantise). */
constexpr int kLdsDenseStepOff = (kLdsBytes + 15) & ~15;
constexpr int kLdsDenseCodeOff = kLdsDenseStepOff + AAD_STEP_TABLE_LEN * 16;
/* the same records in 8 bytes {bias << 29 | delta & 0xFFFF, sm21 << 27} (the addend's upper word is zero and costs the
 * reader one v_mov_b32): sixteen of them are one row of the 64 banks a ds_read_b64 sees - no two codes share a bank,
 * where the 16-byte records put codes c and c + 8 on the same banks and every ds_read_b96 takes eight LDS cycles.
 * For the kernels whose waves share a busy LDS (the sector-tiled dense decoder) - a lone wave prefers the one wide
 * lookup without the v_mov. */
constexpr int kLdsDenseCode8Off = kLdsDenseCodeOff + 16 * 16;
constexpr int kLdsBytesDenseDec = kLdsDenseCode8Off + 16 * 8;
constexpr int kWideStepShift = 9; /* the encoders' wide records hold step << 9: a 24-bit factor for v_mul_hi_u32_u24 */
__device__ __forceinline__ uint32_t wide4_addr(int32_t j, uint32_t copy_off) { return ((uint32_t)j & (0xFF0u * kIdxScale)) | copy_off; }
/* byte offset of this lane's copy inside a slot of the encoders' wide table */
__device__ __forceinline__ uint32_t wide_copy_offset()
{
  const uint32_t l = threadIdx.x;
  return kWideCopies == 8 ? ((l & 3u) | ((l >> 2) & 4u)) << 4 : (l & 3u) << 4;
}

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
template <int BITS, bool QUAD, int WIDE_STEP_SHIFT = 0, bool WIDE4 = false>
__device__ __forceinline__ void stage_tables(char *lds)
{
  constexpr int kShift = BITS - 1;
  auto put = [&](int i, uint32_t step_i, uint32_t recip_i) {
    const float hr = __uint_as_float(recip_i);
    const float hs = hr * (float)(1 << kShift); /* exact power-of-two scaling */
    reinterpret_cast<uint32_t *>(lds + kLdsStepOff)[i] = step_i;
    reinterpret_cast<float *>(lds + kLdsHrOff)[i] = hr;
    reinterpret_cast<float *>(lds + kLdsHsOff)[i] = hs;
    if (QUAD) {
      u32x4 e;
      e.x = step_i << WIDE_STEP_SHIFT; /* the encoders want step << kWideStepShift, see encode_chunk16_quad */
      e.y = __float_as_uint(hr);
      e.z = __float_as_uint(hs);
      e.w = 0;
      if (WIDE4) {
        /* every copy of slot i, in an order rotated by i: the eight contiguous lanes a ds_write_b128 serves per LDS
         * cycle then write eight DIFFERENT copies - a copy is a bank window - instead of the same copy of eight
         * slots (all on the same four banks: eight-way conflicts on every staging store) */
#pragma unroll
        for (int r = 0; r < kWideCopies; r++)
          *reinterpret_cast<u32x4 *>(lds + kLdsWideOff + i * (16 * kWideCopies) + (((r + i) & (kWideCopies - 1)) << 4)) = e;
      } else {
        *reinterpret_cast<u32x4 *>(lds + kLdsWideOff + (i << 4)) = e;
      }
    }
  };
  static_assert(AAD_STEP_TABLE_LEN == 256, "four rounds of one wave");
  if (blockDim.x == 64) {
    /* one wave (the quad encoder of a lane-starved batch): its four rounds' table words are all requested before the first is
     * used - round by round, each round waited for its own trip to the constant data */
    uint32_t step_w[4], recip_w[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      step_w[k] = c_step_table[threadIdx.x + 64 * k];
      recip_w[k] = c_half_recip_bits[threadIdx.x + 64 * k];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) put((int)threadIdx.x + 64 * k, step_w[k], recip_w[k]);
  } else {
    for (int i = threadIdx.x; i < AAD_STEP_TABLE_LEN; i += blockDim.x) put(i, c_step_table[i], c_half_recip_bits[i]);
  }
  const int16_t *dt = BITS == 4 ? c_delta4 : (BITS == 3 ? c_delta3 : c_delta2);
  if (threadIdx.x < (1 << BITS)) {
    /* code -> {signed 2*mag+1, index delta | rounding bias << 16}:
     * qd = (step * sm21 + bias) >> (BITS-1) equals the reference's sign ? -q : q, q = (step * m21) >> (BITS-1) */
    const int code = threadIdx.x, mag = code & ((1 << kShift) - 1), neg = code >> kShift;
    u32x2 e;
    e.x = (uint32_t)(neg ? -(2 * mag + 1) : (2 * mag + 1));
    e.y = ((uint32_t)(int32_t)dt[mag] & 0xFFFFu) | ((neg ? (1u << kShift) - 1u : 0u) << 16);
    *reinterpret_cast<u32x2 *>(lds + kLdsCodeOff + (code << kLdsCodeShift)) = e;
  }
  if (threadIdx.x < 8) {
    reinterpret_cast<int16_t *>(lds + kLdsDeltaOff)[threadIdx.x] = dt[threadIdx.x & ((1 << kShift) - 1)];
    reinterpret_cast<int16_t *>(lds + kLdsDeltaScaledOff)[threadIdx.x] = (int16_t)(kIdxScale * dt[threadIdx.x & ((1 << kShift) - 1)]);
  }
  __syncthreads();
}

/* the dense decoder's extra tables (after stage_tables, which ends in a barrier) */
template <int BITS>
__device__ __forceinline__ void stage_dense_decode_tables(char *lds)
{
  constexpr int kShift = BITS - 1;
  const int16_t *dt = BITS == 4 ? c_delta4 : (BITS == 3 ? c_delta3 : c_delta2);
  for (int i = threadIdx.x; i < AAD_STEP_TABLE_LEN; i += blockDim.x) {
    const uint32_t v = (uint32_t)c_step_table[i] << 2;
    u32x4 e = {v, v, v, v};
    *reinterpret_cast<u32x4 *>(lds + kLdsDenseStepOff + (i << 4)) = e;
  }
  if (threadIdx.x < (1 << BITS)) {
    const int code = threadIdx.x, mag = code & ((1 << kShift) - 1), neg = code >> kShift;
    const int32_t sm21 = neg ? -(2 * mag + 1) : (2 * mag + 1);
    const uint32_t bias = neg ? (1u << kShift) - 1u : 0u;
    u32x4 e;
    /* (step * sm21 + bias) >> (BITS - 1) as the upper half of (step << 2) * (sm21 << 27) + (bias << 29),
     * scaled for BITS = 4 (>> 3 = 29 - 32); fewer bits shift less: the factor moves up accordingly.
     * Whatever sits in the low 29 bits of the addend (the index delta does) cannot carry into bit 29
     * of a sum of multiples of 2^29 and is dropped with the lower half. */
    e.x = (bias << (32 - kShift)) | ((uint32_t)(int32_t)dt[mag] & 0xFFFFu);
    e.y = 0;
    e.z = (uint32_t)sm21 << (30 - kShift);
    e.w = 0;
    *reinterpret_cast<u32x4 *>(lds + kLdsDenseCodeOff + (code << 4)) = e;
    *reinterpret_cast<u32x2 *>(lds + kLdsDenseCode8Off + (code << 3)) = u32x2{e.x, e.z};
  }
  __syncthreads();
}

/* qd = (step * sm21 + bias) >> (BITS - 1), sign included, in one v_mad_i64_i32: step4 = step << 2,
 * rec = the code's record {addend, 0, factor} */
__device__ __forceinline__ int32_t dense_dequantise(uint32_t step4, const u32x3 &rec)
{
  const uint64_t addend = (uint64_t)rec.x | ((uint64_t)rec.y << 32);
  const int64_t t = (int64_t)(int32_t)rec.z * (int64_t)(int32_t)step4 + (int64_t)addend;
  return (int32_t)(t >> 32);
}

__device__ __forceinline__ int32_t clip16(int32_t v) { return min(max(v, -32768), 32767); }

/* Pin a value's computation between the surrounding scheduling barriers.  sched_barrier only
 * orders instructions with side effects; pure arithmetic is free to sink past it (it did: the
 * compiler ran all sixteen table lookups first and the whole LMS chain afterwards).  A volatile
 * empty asm that "modifies" the value is ordered against the barriers and emits nothing. */
__device__ __forceinline__ void pin(int32_t &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(uint32_t &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(float &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin64(uint64_t &v) { asm volatile("" : "+v"(v)); }
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

/* the decoder's per-code record (stage_tables): the delta is read as the low half by the index add
 * itself (v_add_u32_sdwa), the bias costs one shift.  (Tried: a v_dot2_i32_i16 against {step, 0} or a
 * v_mad_i32_i16 reading packed halves in place - the first is followed by three wait states on
 * gfx950, the second, as inline asm, by one; the compiler's own 16-bit multiply-add sign-extends both
 * factors first.) */
__device__ __forceinline__ u32x2 code_record(const char *lds, uint32_t code_times_8) { return *reinterpret_cast<const u32x2 *>(lds + kLdsCodeOff + code_times_8); }
__device__ __forceinline__ int32_t record_delta(const u32x2 &t) { return (int32_t)(int16_t)t.y; }
/* (step * sm21 + bias) >> (BITS - 1): the dequantised difference, sign included */
template <int BITS>
__device__ __forceinline__ int32_t record_dequantise(uint32_t step, const u32x2 &t)
{
  return mad_i24((int32_t)step, (int32_t)t.x, (int32_t)(t.y >> 16)) >> (BITS - 1);
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

/* LMS and history update - reference src/aad_encoder.c:396-406, src/aad_decoder.c:306-315:
 *   w += (qd * h + 16384) >> 18.
 * One lane holding all four taps (the dense mapping) spends three instructions per tap on that
 * (v_mad_i32_i24, v_ashrrev_i32, v_add_u32).  With q14 = qd << 14, computed once per sample,
 *   (qd * h + 16384) >> 18  =  high 32 bits of  q14 * h + 2^28        (64-bit, arithmetic floor)
 * exactly - (qd * h + 16384) * 2^14 is the 64-bit sum, its upper half the same floor - so a tap is
 * one v_mad_i64_i32 (the 2^28 sits in a scalar register pair) and one add: 9 instructions per sample
 * instead of 12.  |qd| <= 61438 < 2^16, so q14 fits 31 bits. */
__device__ __forceinline__ int32_t lms_q14(int32_t qd) { return (int32_t)((uint32_t)qd << 14); }
__device__ __forceinline__ int32_t lms_tap(int32_t w, int32_t q14, int32_t h)
{
  const int64_t t = (int64_t)q14 * (int64_t)h + (int64_t)(1ll << 28);
  return (int32_t)((uint32_t)w + (uint32_t)(t >> 32));
}
__device__ __forceinline__ void lms_and_shift(Lane &L, int32_t qd, int32_t y)
{
  const int32_t q14 = lms_q14(qd);
  L.w0 = lms_tap(L.w0, q14, L.h0);
  L.w1 = lms_tap(L.w1, q14, L.h1);
  L.w2 = lms_tap(L.w2, q14, L.h2);
  L.w3 = lms_tap(L.w3, q14, L.h3);
  L.h3 = L.h2;
  L.h2 = L.h1;
  L.h1 = L.h0;
  L.h0 = y;
}

/* LMS, history shift and the NEXT sample's prediction in one: tap i of the prediction needs only the
 * updated weight i and the shifted history, so the eight 64-bit multiply-adds alternate - no two
 * dependent ones back to back (a dependent pair costs a wait state) */
__device__ __forceinline__ int32_t lms_shift_predict(Lane &L, int32_t qd, int32_t y)
{
  const int32_t q14 = lms_q14(qd);
  uint64_t acc = 16384u;
  const int32_t w0 = lms_tap(L.w0, q14, L.h0);
  acc += (uint64_t)(uint32_t)y * (uint64_t)(uint32_t)w0;
  pin64(acc);
  const int32_t w1 = lms_tap(L.w1, q14, L.h1);
  acc += (uint64_t)(uint32_t)L.h0 * (uint64_t)(uint32_t)w1;
  pin64(acc);
  const int32_t w2 = lms_tap(L.w2, q14, L.h2);
  acc += (uint64_t)(uint32_t)L.h1 * (uint64_t)(uint32_t)w2;
  /* The last tap's increment is pinned together with the sum and nothing is pinned behind the last
   * multiply-add: an instruction that reads a pinned value right behind the pin gets an s_nop in front of
   * it (the hazard recogniser pads every inline-asm result) - with a pin of the sum alone here and a fourth
   * one at the end, the last multiply-add and the final shift each did, and a latency-bound launch pays
   * an issue slot for each.  (No pin at all: the last two products become v_mul_lo_u32, quarter rate.) */
  int32_t inc3 = (int32_t)(((int64_t)q14 * (int64_t)L.h3 + (int64_t)(1ll << 28)) >> 32);
  asm volatile("" : "+v"(acc), "+v"(inc3));
  const int32_t w3 = (int32_t)((uint32_t)L.w3 + (uint32_t)inc3);
  acc += (uint64_t)(uint32_t)L.h2 * (uint64_t)(uint32_t)w3;
  L.h3 = L.h2;
  L.h2 = L.h1;
  L.h1 = L.h0;
  L.h0 = y;
  L.w0 = w0;
  L.w1 = w1;
  L.w2 = w2;
  L.w3 = w3;
  return (int32_t)(uint32_t)acc >> 15;
}

/* LMS split in two halves for the hand-pipelined encoder (taps 0-1, then taps 2-3 + shift) */
__device__ __forceinline__ void lms_first(Lane &L, int32_t qd)
{
  const int32_t q14 = lms_q14(qd);
  L.w0 = lms_tap(L.w0, q14, L.h0);
  L.w1 = lms_tap(L.w1, q14, L.h1);
}
__device__ __forceinline__ void lms_rest_and_shift(Lane &L, int32_t qd, int32_t y)
{
  const int32_t q14 = lms_q14(qd);
  L.w2 = lms_tap(L.w2, q14, L.h2);
  L.w3 = lms_tap(L.w3, q14, L.h3);
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

/* Two lane layouts of a quad.  Tap-minor (decoders): the four taps of a recurrence are four ADJACENT
 * lanes, cross-tap traffic is quad_perm DPP.  Tap-major (encoder): within every row of sixteen lanes
 * lane = 4 * tap + r, i.e. bank b of the row holds tap b of four recurrences r = 0..3 - cross-tap
 * traffic is row_ror / row_shr by 4, and because a DPP write can be switched off per BANK, the
 * history shift "every tap takes the next-newer tap's sample, tap 0 keeps the new one" is ONE
 * instruction (see kShiftBankMask).  The channels of a stereo stream are then adjacent lanes. */
constexpr int kDppRowShr4 = 0x114, kDppRowRor4 = 0x124, kDppRowRor8 = 0x128;
template <bool TM> __device__ __forceinline__ uint32_t quad_tap() { return TM ? (threadIdx.x >> 2) & 3u : threadIdx.x & 3u; }
/* index of the lane's recurrence among the 16 of its wave */
template <bool TM> __device__ __forceinline__ uint32_t quad_slot()
{
  const uint32_t l = threadIdx.x & 63u;
  return TM ? ((l >> 4) << 2) | (l & 3u) : l >> 2;
}
/* lane (within the wave) of tap t of recurrence slot q */
template <bool TM> __device__ __forceinline__ uint32_t quad_lane_of(uint32_t q, uint32_t t)
{
  return TM ? ((q >> 2) << 4) | (t << 2) | (q & 3u) : (q << 2) | t;
}

template <bool TM = false>
__device__ __forceinline__ int32_t predict(const QuadLane &Q)
{
  uint32_t s = (uint32_t)Q.h * (uint32_t)Q.w + Q.round;
  s += quad_dpp<TM ? kDppRowRor4 : 0xB1>(s); /* quad_perm [1,0,3,2] */
  s += quad_dpp<TM ? kDppRowRor8 : 0x4E>(s); /* quad_perm [2,3,0,1] */
  return (int32_t)s >> 15;
}
__device__ __forceinline__ void lms_first(QuadLane &Q, int32_t qd) { Q.w += mad_i24(qd, Q.h, 16384) >> 18; }
/* History shift: every tap takes the next-newer tap's sample, tap 0 the reconstructed one.
 *   kShiftBankMask  tap-major layout only, ONE instruction: v_mov_b32_dpp row_shr:4 with bank_mask
 *                   0xE - the DPP write is switched off for bank 0 of every row (= the tap-0 lanes),
 *                   which keeps what the destination register held before: y.  (y must not be needed
 *                   afterwards, or the compiler pays a copy; the DPP may not follow the write of y by
 *                   fewer than two instructions - the LMS update sits in between.)
 *   kShiftSelect    v_mov_b32_dpp + v_cndmask on a lane mask the compiler keeps in VCC.
 *   kShiftBitSelect v_and_b32_dpp + v_and_or_b32 on a per-lane VGPR mask: for bodies with no spare
 *                   VCC, where the select's mask ended up in an SGPR pair in some instantiations
 *                   (the trial-search kernels) and those ran every chunk ~24 % slower. */
enum { kShiftBitSelect = 0, kShiftSelect = 1, kShiftBankMask = 2 };
template <int MODE = kShiftBitSelect>
__device__ __forceinline__ void lms_rest_and_shift(QuadLane &Q, int32_t, int32_t y)
{
  if (MODE == kShiftBankMask) {
    Q.h = __builtin_amdgcn_update_dpp(y, Q.h, kDppRowShr4, 0xF, 0xE, false);
    return;
  }
  const uint32_t up = quad_dpp<0x90>((uint32_t)Q.h); /* quad_perm [0,0,1,2]: the next-older tap's sample */
  if (MODE == kShiftSelect) Q.h = Q.tap0 ? y : (int32_t)up;
  else Q.h = (int32_t)(((uint32_t)y & Q.newest) | (up & ~Q.newest));
}
template <int MODE = kShiftBitSelect>
__device__ __forceinline__ void lms_and_shift(QuadLane &Q, int32_t qd, int32_t y)
{
  lms_first(Q, qd);
  lms_rest_and_shift<MODE>(Q, qd, y);
}
__device__ __forceinline__ void pin_weights(QuadLane &Q) { pin(Q.w); }

/* full state <-> quad (block boundaries only) */
template <bool BITMASK = true>
__device__ __forceinline__ QuadLane to_quad(const Lane &L, uint32_t tap)
{
  QuadLane Q;
  /* tap t's weight and history by bit masks, the masks opaque to the compiler: written as a chain of conditionals
   * (tap == 0 ? L.w0 : tap == 1 ? L.w1 : ...) LLVM turned the selection into a table on the STACK - nine scratch stores and two
   * indexed scratch loads per block boundary, and a kernel that uses scratch at all is launched under the queue's scratch-wave
   * limit: the dual trial-search encoder (76 bytes of scratch per lane) ran at most ~540 waves at a time, so batches of
   * 4097-8192 recurrences took up to 1.6x as long as their waves needed (profiles/r04_trial_search_size_sweep.txt). */
  uint32_t m0 = tap == 0 ? 0xFFFFFFFFu : 0u, m1 = tap == 1 ? 0xFFFFFFFFu : 0u, m2 = tap == 2 ? 0xFFFFFFFFu : 0u, m3 = tap == 3 ? 0xFFFFFFFFu : 0u;
  pin(m0);
  pin(m1);
  pin(m2);
  pin(m3);
  Q.w = (int32_t)(((uint32_t)L.w0 & m0) | ((uint32_t)L.w1 & m1) | ((uint32_t)L.w2 & m2) | ((uint32_t)L.w3 & m3));
  Q.h = (int32_t)(((uint32_t)L.h0 & m0) | ((uint32_t)L.h1 & m1) | ((uint32_t)L.h2 & m2) | ((uint32_t)L.h3 & m3));
  Q.idxb = L.idxb;
  Q.round = tap == 0 ? 16384u : 0u;
  Q.tap0 = tap == 0;
  Q.newest = tap == 0 ? 0xFFFFFFFFu : 0u;
  if (BITMASK) pin(Q.newest); /* opaque: keeps the bit-select from being turned back into a v_cndmask */
  return Q;
}
template <bool TM = false>
__device__ __forceinline__ Lane from_quad(const QuadLane &Q)
{
  Lane L;
  if (TM) { /* block boundaries only: eight wave shuffles, one wait */
    const uint32_t q = quad_slot<true>();
    L.w0 = __shfl(Q.w, (int)quad_lane_of<true>(q, 0), 64);
    L.w1 = __shfl(Q.w, (int)quad_lane_of<true>(q, 1), 64);
    L.w2 = __shfl(Q.w, (int)quad_lane_of<true>(q, 2), 64);
    L.w3 = __shfl(Q.w, (int)quad_lane_of<true>(q, 3), 64);
    L.h0 = __shfl(Q.h, (int)quad_lane_of<true>(q, 0), 64);
    L.h1 = __shfl(Q.h, (int)quad_lane_of<true>(q, 1), 64);
    L.h2 = __shfl(Q.h, (int)quad_lane_of<true>(q, 2), 64);
    L.h3 = __shfl(Q.h, (int)quad_lane_of<true>(q, 3), 64);
    L.idxb = Q.idxb;
    return L;
  }
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

/* the compiler sees one thread: tell it that LDS written here is read by OTHER lanes of the wave (no instruction:
 * the LDS serves a wave's accesses in order) */
__device__ __forceinline__ void wave_lds_fence()
{
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

/* A 16-byte store that does not stay in the L2 (sc1: write-through, the line is dropped - MI355X_MICROARCH.md, "stores of
 * each flavour").  For output that is written once, in whole sectors, and never read back (the tiled decoder's PCM, the
 * encoders' image sectors): kept in the L2 it evicts lines that WILL be read again (the tiled mono decoder fetched every
 * code line twice: FETCH_SIZE x 2 = 2.0x the code bytes), and half-written 128-byte lines left the L2 as 1.27x their bytes. */
#ifndef AAD_TILED_STORE_SC1
#define AAD_TILED_STORE_SC1 1
#endif
__device__ __forceinline__ void store_through(uint64_t address, const u32x4 &v)
{
#if AAD_TILED_STORE_SC1
  /* s_nop 1 inside the asm: a VMEM store of more than 64 bits needs two wait states before a VALU instruction overwrites its
   * data registers (gfx940+ ISA hazard).  LLVM's hazard recogniser inserts them for instructions it selected itself, not for
   * inline asm - and the data registers are dead after this statement, so the register allocator is free to hand them to the
   * very next VALU result. */
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(address), "v"(v) : "memory");
#else
  *reinterpret_cast<u32x4 *>(address) = v;
#endif
}

/* ---- per-lane byte shuffles ------------------------------------------------------------- */

/* v_perm_b32: selector bytes 0-3 pick from `lo`, 4-7 from `hi`, 0x0c yields 0x00 */
__device__ __forceinline__ uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

/* value of the lane that holds the other channel of a stereo pair.  QUAD = tap-minor quad layout
 * (the pair is four lanes away); dense mapping and tap-major quad layout: the neighbouring lane. */
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

} /* namespace aad */

#endif /* AAD_DEVICE_HIP_H */
