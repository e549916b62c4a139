/*
 * aad_encode.hip.h - the encoder side of the device code: single step, the hand-pipelined
 * 16-sample chunk bodies (dense and quad mapping), sample fetch, block framing and code stores,
 * the trial search, and encode_streams_kernel (reference src/aad_encoder.c:343-727, 814-891).
 * Shared pieces (tables, lane state, LMS, prediction, shuffles) are in aad_device.hip.h.
 */
#ifndef AAD_ENCODE_HIP_H
#define AAD_ENCODE_HIP_H

#include "aad_device.hip.h"

namespace aad {

/* the encoder's quad kernels use the tap-major lane layout (aad_device.hip.h) */
constexpr bool kEncTM = true;

/* Q4 step-index delta of a magnitude code as arithmetic (the constants of reference
 * src/aad_tables.c:8-45): 4-bit {-18,-17,-14,16,32,64,128,256}, 3-bit {-16,-15,32,128},
 * 2-bit {-14,40}.  Five instructions instead of an LDS lookup: used where that lookup would sit
 * on the recurrence's critical path and there are instruction slots to spare (quad encoder). */
template <int BITS>
__device__ __forceinline__ int32_t index_delta_arith(uint32_t mag)
{
  if (BITS == 4) {
    /* 2 << mag, minus {20, 21, 22, 0, 0, 0, 0, 0}[mag] picked by one v_perm_b32 byte lookup:
     * no compare/select pair, no SGPR hazard */
    /* selector = mag as it is: its upper bytes (zero) replicate table byte 0 into bytes 1-3 of the
     * result, which the subtract then ignores by reading byte 0 only (v_sub_u32_sdwa) */
    const uint32_t corr = __builtin_amdgcn_perm(0u, 0x00161514u, mag) & 0xFFu;
    return (int32_t)(2u << mag) - (int32_t)corr;
  } else if (BITS == 3) {
    return mag < 2 ? (int32_t)mag - 16 : (int32_t)(2u << (2u * mag));
  } else {
    return mag ? 40 : -14;
  }
}

/* the same delta times kIdxScale, for the encoders' scaled step index (4-bit: the correction bytes
 * kIdxScale * {20, 21, 22} still fit a byte at scale 8: 160, 168, 176) */
template <int BITS>
__device__ __forceinline__ int32_t index_delta_arith_scaled(uint32_t mag)
{
  constexpr uint32_t K = kIdxScale;
  if (BITS == 4) {
    const uint32_t corr = __builtin_amdgcn_perm(0u, (22u * K) << 16 | (21u * K) << 8 | (20u * K), mag) & 0xFFu;
    return (int32_t)((2u * K) << mag) - (int32_t)corr;
  } else if (BITS == 3) {
    return mag < 2 ? (int32_t)(K * mag) - (int32_t)(16u * K) : (int32_t)((2u * K) << (2u * mag));
  } else {
    return mag ? (int32_t)(40u * K) : -(int32_t)(14u * K);
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
  int32_t p;
  if constexpr (std::is_same_v<S, QuadLane>) p = predict<kEncTM>(L); else p = predict(L);
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
  if constexpr (std::is_same_v<S, QuadLane>) lms_and_shift<kEncTM ? kShiftBankMask : kShiftBitSelect>(L, qd, clip16(qd + p));
  else lms_and_shift(L, qd, clip16(qd + p));
  return mag | ((uint32_t)m & Pack<BITS>::kSign);
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
  /* The step record {step << 9, hr, hs} comes from the multi-copy wide table with ONE lookup at an address
   * that is one v_and_or_b32 of the scaled step index (see wide4_addr) - the three dense dword arrays
   * cost two address instructions and three lookups per sample.  With eight copies the eight lanes a
   * 96-bit read serves per LDS cycle never share a bank (aad_device.hip.h, kWideCopies). */
  int32_t idxj = L.idxb * kIdxScale;
  const uint32_t copy = wide_copy_offset();
  u32x3 e = *reinterpret_cast<const u32x3 *>(lds + kLdsWideOff + wide4_addr(idxj, copy));
  int32_t p = predict(L);
  int32_t d = sample(0) - p;
  int32_t m = d >> 31;
  float f = (float)d;
  static_for<0, kChunk>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    /* A */
    const uint32_t step9 = e.x; /* step << kWideStepShift (stage_tables), 24 bits at most */
    const uint32_t mag = min((uint32_t)__builtin_fmaf(__builtin_fabsf(f), __uint_as_float(e.z), __uint_as_float(e.y)), Pack<BITS>::kMagMax);
    const int32_t delta = *reinterpret_cast<const int16_t *>(lds + kLdsDeltaScaledOff + (mag << 1)); /* kIdxScale * delta */
    __builtin_amdgcn_sched_barrier(0);
    /* B: (step * (2 mag + 1)) >> (BITS - 1) as ONE high multiply on the 24-bit multiplier, as in the quad body */
    const uint32_t m21s = (mag << (25 - BITS)) | (1u << (24 - BITS));
    uint32_t code = 0;
    if (EMIT) { /* pinned HERE, several instructions in front of its reader: no s_nop behind the pin */
      code = ((uint32_t)m & Pack<BITS>::kSign) | mag; /* v_and_or_b32 */
      pin(code);
    }
    const int32_t q = (int32_t)(uint32_t)(((uint64_t)(step9 & 0xFFFFFFu) * (uint64_t)(m21s & 0xFFFFFFu)) >> 32); /* v_mul_hi_u32_u24 */
    const int32_t qd = (q ^ m) - m;
    const int32_t y = clip16(qd + p);
    if (EMIT) {
      uint32_t &acc = w[j / Pack<BITS>::kCodesPerWord];
      acc = (acc << BITS) | code; /* v_lshl_or_b32 */
      pin(acc);
    } else {
      sq += wrapped_square(qd);
    }
    __builtin_amdgcn_sched_barrier(0);
    /* C */
    idxj = min(max(idxj + delta, kIdxScale * kIdxMin), kIdxScale * kIdxMax);
    if (j + 1 < kChunk) e = *reinterpret_cast<const u32x3 *>(lds + kLdsWideOff + wide4_addr(idxj, copy));
    __builtin_amdgcn_sched_barrier(0);
    /* D: LMS, history shift and the next prediction as alternating 64-bit multiply-adds (lms_shift_predict) */
    if (j + 1 < kChunk) {
      p = lms_shift_predict(L, qd, y);
      d = sample(j + 1) - p;
      m = d >> 31;
      f = (float)d;
      pin(m);
      pin(f);
    } else {
      lms_and_shift(L, qd, y);
      qd_out = qd;
      pin_weights(L);
    }
    __builtin_amdgcn_sched_barrier(0);
  });
  L.idxb = idxj >> kIdxScaleLog2;
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
  int32_t j;       /* kIdxScale * (biased step index): the chunk bodies' own form of L.idxb */
  uint32_t copy;   /* byte offset of this lane's copy of the wide records inside a slot (wide_copy_offset) */
};

template <int BITS>
__device__ __forceinline__ void encode_prime_quad(QuadLane &L, EncodeCarry &C, int32_t x0, const char *lds)
{
  C.j = L.idxb * kIdxScale;
  C.copy = wide_copy_offset();
  C.e = *reinterpret_cast<const u32x3 *>(lds + kLdsWideOff + wide4_addr(C.j, C.copy));
  C.p = predict<kEncTM>(L);
  C.d = x0 - C.p;
  C.m = C.d >> 31;
  C.f = (float)C.d;
}

/* x: this chunk's 16 samples, xn0: the first sample of the next chunk (the pipeline is carried
 * from chunk to chunk like the decoder's, see DecodeCarry) */
/* How the chunk's samples sit in x (and the next chunk's first one in xn0):
 *   kWide    sixteen sign-extended values (M/S: the transform needs them widened)
 *   kPairs   eight dwords of two int16 samples each - mono PCM exactly as it was loaded
 *   kFrames  sixteen dwords whose LOW half is this lane's sample - stereo PCM exactly as it was
 *            loaded, the channel-1 lane having loaded from two bytes further on
 * The subtract reads the halves directly (v_sub_u32_sdwa): no per-sample extraction. */
enum { kWide = 0, kPairs = 1, kFrames = 2 };
/* a filler that emits nothing */
struct NoFill {
  template <class J> __device__ __forceinline__ void operator()(J) const {}
};

/* fill(integral_constant<int, j>) is invoked once per sample j, in the first DPP gap: one independent
 * instruction there replaces the s_nop the gap is otherwise padded with - the caller hands in the
 * chunk-level work (loads of a later chunk, the stores of the previous one, pointer arithmetic) one
 * instruction at a time, see run_block */
/* PASS: what a pass over a block leaves behind - the packed codes (kPassEncode), the sum of the wrapped
 * squares of the dequantised differences (kPassRmse: the trial search's measurement, same recurrence, no
 * output), or both at once (kPassBoth: the dual trial search runs measurement and encode passes of
 * different candidates side by side on different lanes of one wave - one instruction stream) */
enum { kPassRmse = 0, kPassEncode = 1, kPassBoth = 2 };

/* N: samples walked (sixteen; fewer for a block's last samples, see run_block's pipelined tail - their codes end up in the LOW
 * bits of the last code word they reach) */
template <int BITS, int PASS, int FORMAT = kWide, typename Fill = NoFill, int N = kChunk>
__device__ __forceinline__ void encode_chunk16_quad(QuadLane &L, EncodeCarry &C, const int32_t *x, int32_t xn0,
                                                    const char *lds, uint32_t *w, int32_t &qd_out, int64_t &sq, Fill fill = Fill())
{
  static_assert(N >= 1 && N <= kChunk, "at most a chunk");
  constexpr bool EMIT = PASS != kPassRmse, SUM = PASS != kPassEncode;
  auto sample = [&](int k) -> int32_t { /* sample k of this chunk (k = 16: first of the next), k compile-time */
    if (FORMAT == kPairs) {
      const int32_t word = k < kChunk ? x[k >> 1] : xn0;
      return (k & 1) ? word >> 16 : (int32_t)(int16_t)word;
    }
    if (FORMAT == kFrames) return (int32_t)(int16_t)(k < kChunk ? x[k] : xn0);
    return k < kChunk ? x[k] : xn0;
  };
  u32x3 e = C.e;
  int32_t p = C.p, d = C.d, m = C.m, idxj = C.j;
  const uint32_t copy = C.copy;
  float f = C.f;
  static_for<0, N>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    /* A */
    const uint32_t mag = min((uint32_t)__builtin_fmaf(__builtin_fabsf(f), __uint_as_float(e.z), __uint_as_float(e.y)),
                             Pack<BITS>::kMagMax);
    const uint32_t step9_j = e.x; /* step << kWideStepShift (stage_tables) */
    idxj = min(max(idxj + index_delta_arith_scaled<BITS>(mag), kIdxScale * kIdxMin), kIdxScale * kIdxMax);
    e = *reinterpret_cast<const u32x3 *>(lds + kLdsWideOff + wide4_addr(idxj, copy));
    __builtin_amdgcn_sched_barrier(0);
    /* B: q = (step * (2 mag + 1)) >> (BITS-1) as ONE high multiply on the 24-bit multiplier
     * (v_mul_hi_u32_u24 issues like an add, the 32-bit v_mul_hi_u32 costs a lone wave ~1.5 cycles more):
     * (step << 9) * ((2 mag + 1) << (24 - BITS)) = step * (2 mag + 1) * 2^(33 - BITS), upper 32 bits of
     * the 48-bit product.  step < 2^15 and 2 mag + 1 < 2^BITS, so both factors fit 24 bits. */
    const uint32_t m21s = (mag << (25 - BITS)) | (1u << (24 - BITS));
    const int32_t q = (int32_t)(uint32_t)(((uint64_t)(step9_j & 0xFFFFFFu) * (uint64_t)(m21s & 0xFFFFFFu)) >> 32); /* v_mul_hi_u32_u24 */
    const int32_t qd = (q ^ m) - m;
    const int32_t y = clip16(qd + p);
    lms_and_shift<kEncTM ? kShiftBankMask : kShiftBitSelect>(L, qd, y);
    /* prediction of the next sample, with the two instructions that pack this sample's code
     * placed in the wait states its DPP adds need (see decode_chunk16_quad) */
    uint32_t s = (uint32_t)L.h * (uint32_t)L.w + L.round;
    pin(s);
    uint32_t code = 0, sqw = 0;
    if (EMIT) {
      code = ((uint32_t)m & Pack<BITS>::kSign) | mag; /* v_and_or_b32 */
      pin(code);
    }
    if (SUM) {
      sqw = (uint32_t)qd * (uint32_t)qd;
      pin(sqw);
    }
    fill(jc);
    s += quad_dpp<kEncTM ? kDppRowRor4 : 0xB1>(s);
    pin(s);
    if (EMIT) {
      uint32_t &acc = w[j / Pack<BITS>::kCodesPerWord];
      acc = (acc << BITS) | code; /* v_lshl_or_b32 */
      pin(acc);
    }
    if (SUM) sq += (int64_t)(int32_t)sqw;
    /* The wait for the step record asked for in A goes HERE.  It has to be somewhere before the next
     * sample's quantiser, and in this slot it doubles as the second wait state the DPP add below
     * needs (where the compiler put it, in front of the quantiser, it cost a slot of its own and
     * this gap was padded with an s_nop).  Not earlier: in the first gap, 13 instructions after the
     * lookup was issued, the record is still on its way and the wave stalls (measured: 72.6 us per
     * launch of the bench batch with the wait in the first gap, 70.4 in front of the quantiser,
     * 68.6 here). */
    __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0) */
    s += quad_dpp<kEncTM ? kDppRowRor8 : 0x4E>(s);
    p = (int32_t)s >> 15;
    d = sample(j + 1) - p;
    /* only ONE of the two is pinned behind the scheduling barrier: with both, the hazard recogniser
     * put an s_nop in front of the quantiser that opens the next sample */
    f = (float)d;
    m = d >> 31;
    pin(m);
    if (j + 1 == N) qd_out = qd;
    __builtin_amdgcn_sched_barrier(0);
  });
  C.e = e;
  C.p = p;
  C.d = d;
  C.m = m;
  C.f = f;
  C.j = idxj;
  L.idxb = idxj >> kIdxScaleLog2; /* the unscaled form everything outside the chunk bodies uses; dead code unless read */
}

/* ================================================================================ encode == */

struct EncodeArgs {
  const StreamDesc *streams;
  const int16_t *pcm;
  uint8_t *data;
  const LaneStateRecord *state; /* carried state read at the start; may be null (fresh encoders) */
  LaneStateRecord *state_out;   /* where the state is left at the end; may be null, may equal `state` */
  uint32_t num_streams;
  uint32_t channels;
  uint32_t block_size;
  uint32_t samples_per_block;
  uint32_t mid_side;
  uint32_t trials;
  uint32_t bits;
  /* Frames at the head of every stream that are context only (0, or one block): encoding starts
   * behind them, the image holds the rest.  The trial search looks one block back in the INPUT
   * (reference src/aad_encoder.c:503-512), so a stream continued from carried state has to bring
   * that block along. */
  uint32_t lead_frames;
  uint32_t ring_ok; /* every image starts on a 64-byte boundary relative to `data` (host-checked): the dense 4- / 2-bit encoders may use ByteRing */
  /* dual trial search (encode_block_dual): three block-sized slots per stream, device memory of the context */
  uint8_t *trial_scratch;
  uint32_t trial_slot_bytes;
  UniformLayout uni;
  alignas(4) uint8_t header_template[32]; /* 31-byte file header with num_samples = 0 */
};

/* sample i of channel c of a stream, after the optional L/R -> M/S transform
 * (reference src/aad_encoder.c:413-428; the clip there can never trigger for int16 input) */
template <bool MS>
struct SampleSource {
  const int16_t *x;
  uint32_t ch, c;
  uint32_t total; /* frames in the stream */
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
/* defer3: leave the last three bytes (low byte of weight 3, history 3) unwritten and return them, lowest
 * address in the low byte - the dense stereo encoder writes them with the first code bytes, which
 * share their 64-byte granule (see run_block) */
__device__ __forceinline__ uint32_t write_block_header(Lane &L, uint8_t *p, bool do_store, bool defer3 = false, uint32_t *words = nullptr)
{
  auto wabs = [](int32_t w) { const int32_t m = w >> 31; return (int32_t)(((uint32_t)w ^ (uint32_t)m) - (uint32_t)m); };
  const int32_t maxabs = max(max(max(wabs(L.w0), wabs(L.w1)), max(wabs(L.w2), wabs(L.w3))), 0);
  const int32_t shift = max(17 - (int32_t)__clz(maxabs), 0); /* smallest shift with maxabs >> shift <= 32767 */
  const int32_t mask = (int32_t)~((1u << shift) - 1u);
  L.w0 &= mask;
  L.w1 &= mask;
  L.w2 &= mask;
  L.w3 &= mask;
  const uint32_t w3 = (uint32_t)(L.w3 >> shift), h3 = (uint32_t)L.h3;
  const uint32_t last3 = (w3 & 0xFFu) | (((h3 >> 8) & 0xFFu) << 8) | ((h3 & 0xFFu) << 16);
  if (words != nullptr) { /* the eighteen bytes in memory order, for the byte ring */
    auto two_ = [](uint32_t first, uint32_t second) { return perm(second, first, 0x04050001); };
    const uint32_t f0_ = ((((uint32_t)(L.idxb - kIdxBias)) << 4) & 0xFFFFu) | ((uint32_t)shift & 0xFu);
    words[0] = two_(f0_, (uint32_t)(L.w0 >> shift));
    words[1] = two_((uint32_t)L.h0, (uint32_t)(L.w1 >> shift));
    words[2] = two_((uint32_t)L.h1, (uint32_t)(L.w2 >> shift));
    words[3] = two_((uint32_t)L.h2, w3);
    words[4] = ((h3 >> 8) & 0xFFu) | ((h3 & 0xFFu) << 8);
  }
  if (!do_store) return last3;
  /* nine big-endian 16-bit fields, stored two to a dword (any alignment: images start on 16-byte
   * boundaries at best, headers never do) instead of byte by byte */
  auto two = [](uint32_t first, uint32_t second) { return perm(second, first, 0x04050001); };
  const uint32_t f0 = ((((uint32_t)(L.idxb - kIdxBias)) << 4) & 0xFFFFu) | ((uint32_t)shift & 0xFu);
  reinterpret_cast<U32 *>(p)->v = two(f0, (uint32_t)(L.w0 >> shift));
  reinterpret_cast<U32 *>(p + 4)->v = two((uint32_t)L.h0, (uint32_t)(L.w1 >> shift));
  reinterpret_cast<U32 *>(p + 8)->v = two((uint32_t)L.h1, (uint32_t)(L.w2 >> shift));
  const uint32_t d3 = two((uint32_t)L.h2, w3);
  if (defer3) {
    reinterpret_cast<U16 *>(p + 12)->v = (uint16_t)d3;
    p[14] = (uint8_t)(w3 >> 8);
  } else {
    reinterpret_cast<U32 *>(p + 12)->v = d3;
    store_be16(p + 16, h3);
  }
  return last3;
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
 * Mono, dense mapping: the code bytes of eight chunks leave together.  A lane that stores its 8 (6, 4) bytes
 * chunk by chunk comes back to the same 64-byte sector eight times, microseconds apart on a full chip - the L2
 * cannot hold a sector per lane that long, every piece reached memory as a write of its own and the saturated
 * mono encoder wrote 5.4x its code bytes (tools/saturated_traffic.sh).  The pieces wait in LDS instead of in
 * registers (the chunk loop stays one body; a register version unrolls it eight times): piece k & 7 of lane l
 * at [k & 7][l] - one conflict-free ds_write per chunk - and after every eighth chunk the lane reads its eight
 * pieces back and stores 64 (48, 32) contiguous bytes with wide stores.  16 KB per workgroup of four waves.
 * Stereo 3- and 2-bit codes the same way, the two lanes of a pair sharing the stores (4-bit stereo keeps its four
 * chunks in registers, see kBurstStores).
 */
template <int BITS, int CHF>
struct CodeStage {
  static_assert(CHF == 1 || (CHF == 2 && BITS != 4), "mono, and the stereo shapes without the register burst");
  static constexpr int kChunks = 8;
  static constexpr int kPiece = BITS == 2 ? 4 : 8;         /* bytes per staged piece (3-bit: six used) */
  static constexpr int kBytesPerWave = kChunks * 64 * kPiece;
  static constexpr int kPieceOut = Pack<BITS>::kChunkBytes; /* bytes of a piece that go to memory */
  char *base;  /* this lane's column */
  char *pair0; /* stereo: the column of the pair's channel-0 lane */
  __device__ __forceinline__ void init(char *stage)
  {
    char *wave = stage + (threadIdx.x >> 6) * kBytesPerWave;
    base = wave + (threadIdx.x & 63u) * kPiece;
    pair0 = wave + (threadIdx.x & 62u) * kPiece;
  }
  /* w: the chunk's big-endian code words of this lane's channel -> the lane's bytes as they go to memory
   * (stereo: its half of the pair's interleaved bytes, as store_chunk_codes<BITS, 2> builds it) */
  __device__ __forceinline__ void put(uint32_t k, const uint32_t *w, uint32_t c)
  {
    char *at = base + (k & (kChunks - 1)) * (64 * kPiece);
    if (CHF == 1) {
      if (BITS == 2) {
        *reinterpret_cast<uint32_t *>(at) = perm(0, w[0], 0x00010203);
      } else if (BITS == 4) {
        *reinterpret_cast<u32x2 *>(at) = u32x2{perm(0, w[0], 0x00010203), perm(0, w[1], 0x00010203)};
      } else { /* a0 a1 a2 a3 | a4 a5 - - : w0 = 0 a0 a1 a2, w1 = 0 a3 a4 a5 */
        *reinterpret_cast<u32x2 *>(at) = u32x2{perm(w[1], w[0], 0x06000102), perm(0, w[1], 0x0c0c0001)};
      }
    } else if (BITS == 2) { /* pair: a0 b0 a1 b1 | a2 b2 a3 b3 ; lane c holds dword c */
      const uint32_t other = pair_swap<false>(w[0], c);
      const uint32_t A = c ? other : w[0], B = c ? w[0] : other;
      *reinterpret_cast<uint32_t *>(at) = perm(A, B, c ? 0x00040105u : 0x02060307u);
    } else { /* pair: a0 a1 a2 b0 b1 b2 | a3 a4 a5 b3 b4 b5 ; lane c holds six bytes c */
      const uint32_t send = c ? w[0] : w[1], keep = c ? w[1] : w[0];
      const uint32_t recv = pair_swap<false>(send, c);
      const uint32_t A = c ? recv : keep, B = c ? keep : recv;
      *reinterpret_cast<u32x2 *>(at) = u32x2{perm(A, B, 0x02040506), perm(A, B, 0x0c0c0001)};
    }
  }
  __device__ __forceinline__ u32x2 piece(const char *col, int j) const
  {
    if (BITS == 2) return u32x2{*reinterpret_cast<const uint32_t *>(col + j * (64 * kPiece)), 0u};
    return *reinterpret_cast<const u32x2 *>(col + j * (64 * kPiece));
  }
  /* two six-byte pieces -> three dwords */
  __device__ __forceinline__ void six2(const u32x2 &a, const u32x2 &b, uint32_t *d) const
  {
    d[0] = a.x;
    d[1] = perm(b.x, a.y, 0x05040100);
    d[2] = perm(b.y, b.x, 0x05040302);
  }
  /* the pieces of a full group (chunks k0 .. k0 + 7, `group` = where chunk k0's bytes go) with wide stores.
   * Mono: the lane's own eight pieces, 64 (48, 32) contiguous bytes.  Stereo: the pair's sixteen pieces
   * alternate in memory - lane c stores the pair's chunks 4c .. 4c + 3, reading both columns. */
  __device__ __forceinline__ void flush(uint8_t *group, uint32_t c) const
  {
    if (CHF == 1) {
      if (BITS == 4) {
#pragma unroll
        for (int v = 0; v < 4; v++) {
          const u32x2 a = piece(base, 2 * v), b = piece(base, 2 * v + 1);
          reinterpret_cast<U32x4 *>(group + 16 * v)->v = u32x4{a.x, a.y, b.x, b.y};
        }
      } else if (BITS == 2) {
#pragma unroll
        for (int v = 0; v < 2; v++)
          reinterpret_cast<U32x4 *>(group + 16 * v)->v =
              u32x4{piece(base, 4 * v).x, piece(base, 4 * v + 1).x, piece(base, 4 * v + 2).x, piece(base, 4 * v + 3).x};
      } else {
        uint32_t d[12];
#pragma unroll
        for (int j = 0; j < 4; j++) six2(piece(base, 2 * j), piece(base, 2 * j + 1), d + 3 * j);
#pragma unroll
        for (int v = 0; v < 3; v++) reinterpret_cast<U32x4 *>(group + 16 * v)->v = u32x4{d[4 * v], d[4 * v + 1], d[4 * v + 2], d[4 * v + 3]};
      }
    } else {
      const char *col0 = pair0 + (4 * c) * (64 * kPiece), *col1 = col0 + kPiece; /* chunk 4c of either channel */
      uint8_t *out = group + (uint64_t)c * (4 * 2 * kPieceOut);
      if (BITS == 2) {
#pragma unroll
        for (int v = 0; v < 2; v++)
          reinterpret_cast<U32x4 *>(out + 16 * v)->v =
              u32x4{piece(col0, 2 * v).x, piece(col1, 2 * v).x, piece(col0, 2 * v + 1).x, piece(col1, 2 * v + 1).x};
      } else {
        uint32_t d[12];
#pragma unroll
        for (int j = 0; j < 4; j++) six2(piece(col0, j), piece(col1, j), d + 3 * j);
#pragma unroll
        for (int v = 0; v < 3; v++) reinterpret_cast<U32x4 *>(out + 16 * v)->v = u32x4{d[4 * v], d[4 * v + 1], d[4 * v + 2], d[4 * v + 3]};
      }
    }
  }
  /* the lane's piece of chunk j of an incomplete group alone, as store_chunk_codes writes it; `at` = where it goes */
  __device__ __forceinline__ void flush_one(int j, uint8_t *at) const
  {
    const u32x2 a = piece(base, j);
    if (BITS == 2) {
      reinterpret_cast<U32 *>(at)->v = a.x;
    } else if (BITS == 4) {
      reinterpret_cast<U32x2 *>(at)->v = a;
    } else {
      reinterpret_cast<U32 *>(at)->v = a.x;
      reinterpret_cast<U16 *>(at + 4)->v = (uint16_t)a.y;
    }
  }
};
/*
 * Per-row byte ring (round 3; dense encoders, the encode pass): the image of a row - a mono lane's stream, or a stereo pair's - is
 * ONE byte stream (file header, then per block: header, packed codes, tail units), and every 64-byte sector of it is stored
 * exactly once, whole, the moment the stream crosses into the next sector.  Before, a lane stored its codes in 64-byte bursts that
 * start where the codes start (49 / 67 bytes into an image): every sector was written in two pieces eight chunks apart, and the
 * encoders wrote 1.3-2.1x their code bytes (profiles/r03_saturated_geometries_pmc.txt); headers and tails went out in small
 * stores of their own.
 *   The ring is 128 bytes of LDS per row in IMAGE alignment (ring offset = image offset & 127; images start on 64-byte boundaries:
 * host-checked).  A chunk's bytes arrive as one or two dwords in memory order at a byte position that is not dword aligned, so
 * they are merged with a carry of up to three bytes (three v_perm_b32 whose selectors depend on the position's low two bits only -
 * constant within a block for 8- and 4-byte pieces) and written as ALIGNED dwords (unaligned LDS accesses cost 65 cycles per
 * instruction on gfx950: tools/microbench/ubench_lds_unaligned.hip).  A stereo pair shares a ring: lane c appends its half of
 * the pair's interleaved bytes behind lane c - 1's, the carry travels from lane 0 to lane 1 within a chunk and back to lane 0 for
 * the next (one DPP swap).  Headers and tail units (a few bytes per block) are written byte by byte around a hand-over of the
 * carry through the ring.  When an append moves the stream into the next sector the row's lanes read the finished one back
 * (ds_read_b128) and store it with 16-byte stores at its own address: mono 4 per lane, stereo 2 per lane.
 */
struct RingSelectors { uint32_t take, shift, keep; }; /* v_perm_b32 selectors for a byte phase (position & 3) */
__device__ __forceinline__ RingSelectors ring_selectors(uint32_t s)
{
  RingSelectors r;
  const uint32_t low = (1u << (8u * s)) - 1u; /* the low s bytes */
  r.shift = 0x07060504u - 0x01010101u * s;   /* bytes 4 - s .. 7 - s of {hi : lo}: the next aligned dword of a stream that sits s bytes in */
  r.take = (r.shift & ~low) | (0x03020100u & low); /* s bytes of the carry (lo), then 4 - s bytes of the new dword (hi) */
  r.keep = ((0x03020100u + 0x01010101u * (4u - s)) & low) | (0x0c0c0c0cu & ~low); /* the last s bytes of a dword, rest zero: the new carry */
  return r;
}

template <int CHF>
struct ByteRing {
  static constexpr uint32_t kPitch = 144; /* 128 bytes + 16: rows a power of two apart would put every row on the same banks */
  char *row;      /* this row's ring */
  uint8_t *image; /* the stream's image in memory (64-byte aligned) */
  uint32_t pos;   /* bytes of the image appended so far, by the whole row */
  uint32_t carry; /* the bytes in front of this lane's next append that do not fill a dword yet (low bytes), see append() */
  uint32_t limit; /* bytes of the image that may be written (data_size) */
  uint32_t c;     /* this lane's channel */

  __device__ __forceinline__ void init(char *area, uint32_t row_index, uint8_t *img, uint32_t data_size, uint32_t channel)
  {
    row = area + row_index * kPitch;
    image = img;
    pos = 0;
    carry = 0;
    limit = data_size;
    c = channel;
  }
  __device__ __forceinline__ char *at(uint32_t p) const { return row + (p & 127u); }

  /* the sector that ends where the stream now stands: whole, to memory (or byte by byte where the image ends inside it) */
  __device__ __forceinline__ void flush_sector(uint32_t sector_start)
  {
    wave_lds_fence();
    const char *from = row + (sector_start & 64u);
    uint8_t *to = image + sector_start;
    if (sector_start + 64u <= limit) {
      constexpr int kPieces = 4 / CHF; /* 16-byte pieces per lane */
#pragma unroll
      for (int i = 0; i < kPieces; i++) {
        /* stereo: the two lanes of a pair store adjacent pieces in one instruction (32 contiguous bytes), the other half next */
        const uint32_t o = 16u * (CHF == 2 ? 2u * i + c : (uint32_t)i);
        /* stereo: the pair's two lanes store 32 contiguous bytes per instruction - written through (sc1), the sector reaches
         * memory as its 64 bytes (1.00x; 1.28x with plain stores: half-written lines).  Mono: one lane, four 16-byte stores -
         * written through they are four partial writes (1.97x); plain stores merge in the L2 (1.06x). */
        if (CHF == 2) store_through(reinterpret_cast<uint64_t>(to + o), *reinterpret_cast<const u32x4 *>(from + o));
        else *reinterpret_cast<u32x4 *>(to + o) = *reinterpret_cast<const u32x4 *>(from + o);
      }
    } else if (c == 0) { /* the caller's buffer ends inside this sector: only the bytes it holds */
      for (uint32_t o = 0; sector_start + o < limit && o < 64u; o++) to[o] = (uint8_t)from[o];
    }
    wave_lds_fence();
  }
  /* `bytes` more bytes have been written behind pos by the row's lanes: move on, store what got complete */
  __device__ __forceinline__ void advance(uint32_t bytes)
  {
    const uint32_t before = pos;
    pos += bytes;
    if (((before ^ pos) & ~63u) != 0) flush_sector(before & ~63u); /* appends are shorter than a sector: at most one boundary */
  }

  /* n_bytes (8 or 4, the same on both lanes of a pair) of this lane in memory order in d0 (and d1): behind the bytes of the
   * lanes before it.  sel: ring_selectors of (pos + c * n_bytes) & 3 - the caller keeps them per block. */
  template <int N>
  __device__ __forceinline__ void append(uint32_t d0, uint32_t d1, const RingSelectors &sel, uint32_t d2 = 0)
  {
    static_assert(N == 12 || N == 8 || N == 4, "whole dwords");
    /* this lane's carry-in: mono - its own carry; stereo - lane 0 takes lane 1's carry of the previous chunk, lane 1 takes
     * lane 0's of this one (same byte phase on both lanes: their pieces are whole dwords long) */
    const uint32_t out = perm(0u, N == 12 ? d2 : (N == 8 ? d1 : d0), sel.keep);
    uint32_t in = carry;
    if (CHF == 2) in = pair_swap<false>(c ? carry : out, c);
    const uint32_t p = pos + c * N;
    const uint32_t m0 = perm(d0, in, sel.take);
    *reinterpret_cast<uint32_t *>(at(p & ~3u)) = m0;
    if (N >= 8) *reinterpret_cast<uint32_t *>(at((p & ~3u) + 4u)) = perm(d1, d0, sel.shift);
    if (N == 12) *reinterpret_cast<uint32_t *>(at((p & ~3u) + 8u)) = perm(d2, d1, sel.shift);
    carry = out;
    advance(N * CHF);
  }

  /* hand the carry over to the ring (before bytes are written one by one) ... */
  __device__ __forceinline__ void carry_to_ring()
  {
    /* the partial dword in front of pos: in a pair it is lane 1's carry (the last piece of a chunk is lane 1's) */
    if ((pos & 3u) != 0 && c == (uint32_t)(CHF - 1)) *reinterpret_cast<uint32_t *>(at(pos & ~3u)) = carry;
    wave_lds_fence();
  }
  /* ... and take it back (before the next append): the lane whose carry the next append reads first */
  __device__ __forceinline__ void carry_from_ring()
  {
    wave_lds_fence();
    const uint32_t s = pos & 3u;
    const uint32_t word = *reinterpret_cast<const uint32_t *>(at(pos & ~3u));
    carry = s ? word & ((1u << (8u * s)) - 1u) : 0u;
  }
  __device__ __forceinline__ void put_byte(uint32_t p, uint32_t value) { *reinterpret_cast<uint8_t *>(at(p)) = (uint8_t)value; }

  /* the end of the stream: what the last sector holds, in the fewest naturally aligned stores (16, 8, 4, 2, 1 bytes: thirty-one
   * single bytes were thirty-one requests the L2 did not always merge) */
  __device__ __forceinline__ void finish()
  {
    carry_to_ring();
    if (c != 0) return;
    const uint32_t start = pos & ~63u;
    uint32_t end = pos < limit ? pos : limit;
    uint32_t o = start;
    for (; o + 16u <= end; o += 16u) *reinterpret_cast<u32x4 *>(image + o) = *reinterpret_cast<const u32x4 *>(at(o));
    if (o + 8u <= end) {
      *reinterpret_cast<u32x2 *>(image + o) = *reinterpret_cast<const u32x2 *>(at(o));
      o += 8u;
    }
    if (o + 4u <= end) {
      *reinterpret_cast<uint32_t *>(image + o) = *reinterpret_cast<const uint32_t *>(at(o));
      o += 4u;
    }
    if (o + 2u <= end) {
      *reinterpret_cast<uint16_t *>(image + o) = *reinterpret_cast<const uint16_t *>(at(o));
      o += 2u;
    }
    if (o < end) image[o] = (uint8_t)*at(o);
  }
};

/* which dense encoders stage their codes, where the staging area starts in their LDS block, and its size */
template <int BITS, int CHF, bool QUAD>
constexpr bool kStagedCodes = !QUAD && (CHF == 1 || (CHF == 2 && BITS != 4));
constexpr int kLdsCodeStageOff = (kLdsBytesQuadEnc + 15) & ~15;
template <int BITS, int CHF, bool QUAD>
constexpr int kLdsBytesEncoder = kStagedCodes<BITS, CHF, QUAD> ? kLdsCodeStageOff + 4 * 8 * 64 * (BITS == 2 ? 4 : 8) : kLdsBytesQuadEnc;
/* the instantiations that move their output through ByteRing: dense, mono / stereo, 4- and 2-bit codes (pieces of whole dwords) */
template <int BITS, int CHF, bool QUAD>
constexpr bool kRingable = !QUAD && (CHF == 1 || CHF == 2);
/* The rows' byte rings live in DYNAMIC LDS, one wave's worth per wave of the workgroup (the launch asks for blockDim.x / 64 of
 * them): a static area for four waves cost the one-wave workgroups of 16 385 .. 65 536-lane batches a resident wave per CU
 * (72 864 B mono / 54 432 B stereo 4-bit per workgroup instead of 45 216 / 40 608). */
template <int CHF>
constexpr int kLdsRingBytesPerWave = (64 / (CHF ? CHF : 1)) * 144;
template <int BITS, int CHF, bool QUAD, bool RING>
constexpr int kLdsBytesEncoderStatic = RING ? kLdsCodeStageOff : kLdsBytesEncoder<BITS, CHF, QUAD>;

/* the lane's bytes of a chunk in memory order (mono: its 8 / 4 code bytes; stereo: its half of the pair's interleaved bytes,
 * as store_chunk_codes / CodeStage::put build them): 4-bit -> d0, d1; 2-bit -> d0 */
template <int BITS, int CHF>
__device__ __forceinline__ void chunk_bytes(const uint32_t *w, uint32_t c, uint32_t &d0, uint32_t &d1)
{
  static_assert(BITS == 4 || BITS == 2, "pieces of whole dwords");
  d1 = 0;
  if (CHF == 1) {
    d0 = perm(0, w[0], 0x00010203);
    if (BITS == 4) d1 = perm(0, w[1], 0x00010203);
  } else if (BITS == 2) { /* pair: a0 b0 a1 b1 | a2 b2 a3 b3 ; lane c holds dword c */
    const uint32_t other = pair_swap<false>(w[0], c);
    const uint32_t A = c ? other : w[0], B = c ? w[0] : other;
    d0 = perm(A, B, c ? 0x00040105u : 0x02060307u);
  } else { /* lane 0 holds the first half of the pair's sixteen bytes (word 0 of both channels), lane 1 the second */
    const uint32_t send = c ? w[0] : w[1], keep = c ? w[1] : w[0];
    const uint32_t recv = pair_swap<false>(send, c);
    const uint32_t A = c ? recv : keep, B = c ? keep : recv;
    d0 = perm(A, B, 0x02060307);
    d1 = perm(A, B, 0x00040105);
  }
}

/* 3-bit codes: four units (24 bits each, the low three bytes of a code word) in memory order - twelve bytes, the whole dwords a
 * ring append wants.  Mono: the units of two chunks; stereo: the L R L R units of one chunk of the pair. */
__device__ __forceinline__ void units12(uint32_t u0, uint32_t u1, uint32_t u2, uint32_t u3, uint32_t &d0, uint32_t &d1, uint32_t &d2)
{
  d0 = perm(u1, u0, 0x06000102);
  d1 = perm(u2, u1, 0x05060001);
  d2 = perm(u3, u2, 0x04050600);
}
/* stereo 3-bit, two chunks of the pair (this lane's code words wa of chunk k, wb of chunk k + 1): lane 0 appends chunk k's twelve
 * bytes, lane 1 chunk k + 1's - each gets the other channel's units of ITS chunk */
__device__ __forceinline__ void pair_units12(const uint32_t *wa, const uint32_t *wb, uint32_t c, uint32_t &d0, uint32_t &d1, uint32_t &d2)
{
  const uint32_t r0 = pair_swap<false>(c ? wa[0] : wb[0], c), r1 = pair_swap<false>(c ? wa[1] : wb[1], c);
  const uint32_t l0 = c ? r0 : wa[0], l1 = c ? r1 : wa[1]; /* channel 0's units of this lane's chunk */
  const uint32_t q0 = c ? wb[0] : r0, q1 = c ? wb[1] : r1; /* channel 1's */
  units12(l0, q0, l1, q1, d0, d1, d2);
}

/* the dense stereo 4-bit encode pass stores its codes four chunks at a time (run_block) */
template <int BITS, int CHF, bool EMIT>
constexpr bool kBurstStores = EMIT && CHF == 2 && BITS == 4;

/*
 * One pass of the recurrence over the coded samples of a block: samples [first+4, first+n) of
 * channel c, history already seeded.  EMIT = the real encode pass (codes packed and stored under
 * `body`); otherwise an RMSE pass of the trial search (reference src/aad_encoder.c:431-467): the
 * same arithmetic, the int32-wrapped squares of the dequantised differences summed instead of any
 * output (every partial sum is an exact integer < 2^53, so an int64 sum converted once equals
 * the reference's running double).  Full 16-sample chunks go through the hand-pipelined bodies
 * with wide prefetched loads, the rest through encode_step.
 */
/* PASS (kPassRmse / kPassEncode / kPassBoth): see encode_chunk16_quad.  kPassBoth exists for the quad
 * mapping only; `pad` then says per lane whether the pass ends like an encode pass (zero padding up to
 * the last whole unit) or like a measurement (at the last real sample) - the sum covers real samples
 * either way. */
/* RING: the encode pass appends its bytes to the row's ByteRing (`ring`) instead of storing them under `body` */
template <int BITS, int CHF, bool MS, bool QUAD, int PASS, bool RING = false, typename S>
__device__ __forceinline__ int64_t run_block(S &L, const SampleSource<MS> &src, uint64_t first, uint32_t n, uint32_t ch,
                                             uint32_t c, bool writer, uint8_t *body, const char *lds, int32_t &last_qd,
                                             bool defer3 = false, uint32_t deferred = 0, bool pad = true,
                                             ByteRing<(CHF ? CHF : 1)> *ring = nullptr)
{
  constexpr bool EMIT = PASS != kPassRmse;
  static_assert(!RING || (EMIT && kRingable<BITS, CHF, QUAD>), "the byte ring serves the dense mono / stereo encode passes");
  constexpr int kPiece = BITS == 4 ? 8 : (BITS == 3 ? 12 : 4); /* RING: bytes per lane and append (3-bit: of two chunks) */
  RingSelectors ring_sel = {0, 0, 0};
  if constexpr (RING) ring_sel = ring_selectors((ring->pos + c * kPiece) & 3u);
  static_assert(PASS != kPassBoth || QUAD, "measurement and encode in one pass: quad mapping only");
  constexpr int US = Pack<BITS>::kUnitSamples, UB = Pack<BITS>::kUnitBytes;
  const uint32_t unit_stride = UB * ch;
  const uint32_t coded = n > (uint32_t)kTaps ? n - kTaps : 0;
  int64_t sq = 0;
  uint32_t done = 0;
  /* kBurstStores: what the chunk loop leaves for the end of the block */
  u32x2 burst_r0 = {0, 0}, burst_r1 = {0, 0}, burst_r2 = {0, 0};
  uint32_t burst_rest = 0, burst_carry = deferred;
  uint8_t *burst_gp = body + 8u * c;
  auto put = [](uint8_t *at, u32x2 v) { reinterpret_cast<U32x2 *>(at)->v = v; };
  auto put_carry = [](uint8_t *at, uint32_t three) { /* at: the granule's first byte */
    reinterpret_cast<U16 *>(at)->v = (uint16_t)three;
    at[2] = (uint8_t)(three >> 16);
  };
  {
    using CS = ChunkSamples<CHF, MS>;
    const uint32_t full = coded / kChunk;
    const int16_t *xp = src.x + (first + kTaps) * ch;
    constexpr uint32_t kOutStride = Pack<BITS>::kChunkBytes;
    CS next;
    for (auto &v : next.d) v = 0;
    if constexpr (!(QUAD && !MS)) {
      if (full) next.load(xp, ch, c);
      next.touch();
    }
    if constexpr (QUAD && !MS) {
      /* Without M/S the loaded dwords ARE what the chunk body reads (kPairs / kFrames above), so
       * three register sets rotate (the loop is unrolled by three, as in the split decoder): chunk k
       * is consumed from one, chunk k+1 - whose first sample the last step of chunk k looks ahead
       * to - sits in the second, chunk k+2 is in flight into the third.  No extraction, no copies;
       * the compiler's own vmcnt waits land where a set is first read, two chunks after its loads
       * were issued (loads older than the code stores in between: nothing waits for a store).
       * Stereo: the channel-1 lane loads from one int16 further on, so that its sample is the low
       * half of every dword as well; the last full chunk of a stream that ends on a chunk boundary
       * is left to the tail loop - its channel-1 load would read two bytes past the stream. */
      constexpr int FMT = CHF == 1 ? kPairs : kFrames;
      constexpr int kParts = CHF == 1 ? 2 : 4; /* 16-byte loads per chunk */
      struct Raw { /* one chunk as loaded: 32 (mono) / 64 (stereo) bytes */
        uint32_t d[CHF == 1 ? 8 : 16];
        __device__ __forceinline__ void load_part(const int16_t *x, int k) /* k compile-time after unrolling */
        {
          const u32x4 a = reinterpret_cast<const U32x4 *>(x + 8 * k)->v;
          d[4 * k] = a.x; d[4 * k + 1] = a.y; d[4 * k + 2] = a.z; d[4 * k + 3] = a.w;
        }
        __device__ __forceinline__ void load(const int16_t *x)
        {
#pragma unroll
          for (int k = 0; k < kParts; k++) load_part(x, k);
        }
      };
      uint32_t chunks = full;
      if (CHF == 2 && chunks && chunks * kChunk == coded && first + n >= (uint64_t)src.total) chunks--;
      const int16_t *rp = xp + (CHF == 2 ? c : 0);
      Raw b0, b1, b2;
      for (auto &v : b0.d) v = 0;
      for (auto &v : b1.d) v = 0;
      for (auto &v : b2.d) v = 0;
      EncodeCarry C;
      if (chunks) {
        b0.load(rp);
        if (chunks > 1) rp += (uint64_t)kChunk * ch;
        b1.load(rp);
        if (chunks > 2) rp += (uint64_t)kChunk * ch; /* rp: where chunk min(2, chunks - 1) starts - what the first chunk prefetches */
        encode_prime_quad<BITS>(L, C, (int32_t)(int16_t)b0.d[0], lds); /* both formats: sample 0 is the low half of dword 0 */
      }
      /* The chunk-level work rides in the first DPP gap of the samples (see encode_chunk16_quad): the
       * loads of chunk k+2 in samples 0-3, then - stereo 4-bit, the BASELINE shape - the store of chunk
       * k-1's codes taken apart into its seven instructions, then the prefetch pointer.  Every tap of
       * a quad (and both roles of the dual mapping) holds the same codes and stores the same bytes to
       * the same address: no exec-masked block, no branch.  Other shapes keep the one-piece store. */
      constexpr bool kStaged = EMIT && kEncTM && CHF == 2 && BITS == 4;
      uint32_t wp0 = 0, wp1 = 0;          /* the previous chunk's code words */
      uint32_t st_send = 0, st_keep = 0, st_recv = 0, st_x = 0, st_y = 0;
      uint8_t *sp = body + 8u * c;        /* where this lane's half of the previous chunk's 16 bytes goes */
      /* A = channel 0's word, B = channel 1's: lane c holds (keep, recv) = c ? (B.., A..) : (A.., B..), so the
       * byte selectors of store_chunk_codes get their source halves swapped on the channel-1 lane */
      const uint32_t sel_x = c ? (0x02060307u ^ 0x04040404u) : 0x02060307u, sel_y = c ? (0x00040105u ^ 0x04040404u) : 0x00040105u;
      auto one = [&](uint32_t k, const Raw &cur, const Raw &ahead, Raw &incoming) {
        uint32_t w[2] = {0, 0};
        const bool pending = kStaged && k != 0;
        if constexpr (!kStaged) { /* (the other shapes ran 1-3 % slower with their loads in the gaps: measured, same box) */
          incoming.load(rp); /* prefetch chunk k+2 (clamped to the last full one) */
          if (k + 3 < chunks) rp += (uint64_t)kChunk * ch;
        }
        auto fill = [&](auto jc) {
          constexpr int j = decltype(jc)::value;
          if constexpr (kStaged) {
            if constexpr (j < kParts) incoming.load_part(rp, j); /* prefetch chunk k+2 (clamped to the last full one) */
            /* lane c writes bytes 8c..8c+7 of the pair's sixteen: a0 b0 a1 b1 | a2 b2 a3 b3 of word c of both
             * channels; it has its own word c and needs the partner's */
            if constexpr (j == 4) { st_send = c ? wp0 : wp1; pin(st_send); }
            if constexpr (j == 5) { st_keep = c ? wp1 : wp0; pin(st_keep); }
            if constexpr (j == 6) { st_recv = pair_swap<false>(st_send, c); pin(st_recv); }
            if constexpr (j == 7) { st_x = perm(st_keep, st_recv, sel_x); pin(st_x); }
            if constexpr (j == 8) { st_y = perm(st_keep, st_recv, sel_y); pin(st_y); }
            if constexpr (j == 9) { /* unconditional: in chunk 0 it puts zeros where chunk 0's own codes land one chunk later */
              u32x2 v;
              v.x = st_x;
              v.y = st_y;
              reinterpret_cast<U32x2 *>(sp)->v = v;
            }
            if constexpr (j == 10) sp += pending ? kOutStride * 2 : 0u;
            if constexpr (j == 12) {
              if (k + 3 < chunks) rp += (uint64_t)kChunk * ch; /* for the next chunk's prefetch */
            }
          }
        };
        encode_chunk16_quad<BITS, PASS, FMT>(L, C, reinterpret_cast<const int32_t *>(cur.d), (int32_t)ahead.d[0], lds, w, last_qd, sq, fill);
        if constexpr (kStaged) {
          wp0 = w[0];
          wp1 = w[1];
        } else {
          if (EMIT && writer) store_chunk_codes<BITS, CHF, QUAD && !kEncTM>(body + (uint64_t)k * kOutStride * ch, w, c);
        }
      };
      for (uint32_t k = 0; k < chunks; k += 3) {
        one(k, b0, b1, b2);
        if (k + 1 < chunks) one(k + 1, b1, b2, b0);
        if (k + 2 < chunks) one(k + 2, b2, b0, b1);
      }
      if constexpr (kStaged) { /* the last chunk's codes */
        if (chunks) {
          const uint32_t w[2] = {wp0, wp1};
          store_chunk_codes<BITS, CHF, false>(body + (uint64_t)(chunks - 1) * kOutStride * ch, w, c);
        }
      }
      done = chunks * kChunk;
    } else if constexpr (QUAD) {
      /* M/S: the transform needs the samples widened (kWide); they are extracted one chunk early */
      constexpr bool PK = false;
      constexpr int kN = kChunk;
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
        encode_chunk16_quad<BITS, PASS, kWide>(L, C, cur, ahead[0], lds, w, last_qd, sq);
        next.touch();
        extract(cur);
        if (EMIT && writer) store_chunk_codes<BITS, (CHF ? CHF : 1), QUAD && !kEncTM>(body + (uint64_t)k * kOutStride * ch, w, c);
      };
      for (uint32_t k = 0; k < full; k += 2) {
        one(k, x, xn);
        if (k + 1 < full) one(k + 1, xn, x);
      }
      done = full * kChunk;
    } else if constexpr (RING && CHF == 2 && BITS == 4) {
      /* dense stereo 4-bit through the byte ring: the same chunk body as the burst path below, the lane's eight bytes appended */
      constexpr bool PK = !MS;
      constexpr int kN = PK ? kChunk / 2 : kChunk;
      const uint32_t pair_sel = c ? 0x07060302u : 0x05040100u;
      for (uint32_t k = 0; k < full; k++) {
        int32_t x[kN];
#pragma unroll
        for (int j = 0; j < kN; j++) x[j] = PK ? (int32_t)next.pair(j, pair_sel) : next.get(j, c);
        if (k + 1 < full) xp += (uint64_t)kChunk * ch;
        next.load(xp, ch, c);
        uint32_t w[2] = {0, 0};
        encode_chunk16<BITS, EMIT, PK>(L, x, lds, w, last_qd, sq);
        next.touch();
        uint32_t d0, d1;
        chunk_bytes<BITS, CHF>(w, c, d0, d1);
        ring->template append<8>(d0, d1, ring_sel);
      }
      done = full * kChunk;
    } else if constexpr (kBurstStores<BITS, CHF, EMIT>) {
      /* Dense stereo 4-bit, the saturated BASELINE shape.  A pair of lanes produces 16 code bytes per
       * chunk; stored chunk by chunk, the four stores that fill a 64-byte granule are a chunk's worth of
       * time apart (microseconds on a full chip), the granule leaves the L2 in between and every store
       * reaches memory as a write of its own: WRITE_SIZE was 4.0x the code bytes.  So the pieces of FOUR
       * chunks are kept in registers and stored back to back (the loop is unrolled by four).  Code
       * bytes start 67 bytes into an image (31 file + 36 block header): with 64-byte aligned images and
       * block sizes every group of 64 code bytes sits 3 bytes into its granule and its last 3 bytes
       * belong to the next one - those (and, at the start of a block, the last 3 bytes of the block
       * header, which share the first code granule) wait in `carry` on the channel-1 lane and are
       * stored with the next group.  defer3 says the block is laid out like that (per lane). */
      constexpr bool PK = !MS;
      constexpr int kN = PK ? kChunk / 2 : kChunk;
      const uint32_t pair_sel = c ? 0x07060302u : 0x05040100u;
      uint8_t *gp = body + 8u * c; /* this lane's half of the group's first chunk */
      uint32_t carry = deferred;
      auto chunk = [&](uint32_t k) -> u32x2 {
        int32_t x[kN];
#pragma unroll
        for (int j = 0; j < kN; j++) x[j] = PK ? (int32_t)next.pair(j, pair_sel) : next.get(j, c);
        if (k + 1 < full) xp += (uint64_t)kChunk * ch; /* unconditional prefetch, see below */
        next.load(xp, ch, c);
        uint32_t w[2] = {0, 0};
        encode_chunk16<BITS, EMIT, PK>(L, x, lds, w, last_qd, sq);
        next.touch();
        /* lane 0 keeps the first half of the pair's sixteen bytes (word 0 of both channels), lane 1 the second */
        const uint32_t send = c ? w[0] : w[1], keep = c ? w[1] : w[0];
        const uint32_t recv = pair_swap<false>(send, c);
        const uint32_t A = c ? recv : keep, B = c ? keep : recv;
        u32x2 v;
        v.x = perm(A, B, 0x02060307);
        v.y = perm(A, B, 0x00040105);
        return v;
      };
      uint32_t k = 0;
      for (; k + 4 <= full; k += 4, gp += 4 * kOutStride * 2) {
        const u32x2 p0 = chunk(k), p1 = chunk(k + 1), p2 = chunk(k + 2), p3 = chunk(k + 3);
        if (defer3) put_carry(gp - 11, carry); /* channel-1 lane: its piece starts 8 + 3 bytes into the granule */
        put(gp, p0);
        put(gp + 16, p1);
        put(gp + 32, p2);
        if (defer3) { /* bytes 59..63 of the granule now, the last three with the next group */
          reinterpret_cast<U32 *>(gp + 48)->v = p3.x;
          gp[52] = (uint8_t)p3.y;
          carry = p3.y >> 8;
        } else {
          put(gp + 48, p3);
        }
      }
      /* the one to three chunks left over: encoded now, stored together with the tail's bytes at the
       * end of the block (below) - r2 is the last of them */
      burst_rest = full - k;
      for (uint32_t j = 0; j < burst_rest; j++) {
        burst_r0 = burst_r1;
        burst_r1 = burst_r2;
        burst_r2 = chunk(k + j);
      }
      burst_gp = gp;
      burst_carry = carry;
      done = full * kChunk;
    } else {
      /* mono / stereo without M/S: the samples stay packed two to a dword (see encode_chunk16) */
      constexpr bool PK = CHF != 0 && !MS;
      constexpr int kN = PK ? kChunk / 2 : kChunk;
      const uint32_t pair_sel = c ? 0x07060302u : 0x05040100u;
      constexpr bool kStage = EMIT && !RING && kStagedCodes<BITS, CHF, false>; /* the codes of eight chunks leave together (CodeStage) */
      CodeStage<BITS, kStage ? CHF : 1> stage;
      if constexpr (kStage) stage.init(const_cast<char *>(lds) + kLdsCodeStageOff);
      uint32_t k0 = 0;
      if constexpr (CHF == 1) {
        /* mono: the samples of TWO chunks (64 bytes) with one group of loads, a pair ahead - a lane that reads
         * 32 bytes per chunk visits every 64-byte sector of its stream three times, a chunk's worth of time apart
         * (the saturated mono encoder fetched 2.2x its PCM), in pairs twice.  The 64-byte window starts at the
         * BLOCK's first sample, not at its first coded one (the four verbatim samples ride in front): on a stream
         * whose PCM starts on a 64-byte boundary every window is then one whole sector - mono blocks are 63 sectors
         * long - where a window that starts 8 bytes in straddles two and a 128-byte line is visited three times
         * instead of twice.  A pair's second chunk ends 8 bytes into the next window: those two dwords always come
         * from the true next window (load_head: they exist whenever the chunk does), the rest of the prefetch is
         * clamped to the last window as before. */
        struct PairSamples {
          uint32_t d[16];
          __device__ __forceinline__ void load_head(const int16_t *x)
          {
            const u32x2 q = reinterpret_cast<const U32x2 *>(x)->v;
            d[0] = q.x; d[1] = q.y;
          }
          __device__ __forceinline__ void load_rest(const int16_t *x)
          {
            const u32x2 h = reinterpret_cast<const U32x2 *>(x + 4)->v;
            d[2] = h.x; d[3] = h.y;
#pragma unroll
            for (int v = 1; v < 4; v++) {
              const u32x4 q = reinterpret_cast<const U32x4 *>(x + 8 * v)->v;
              d[4 * v] = q.x; d[4 * v + 1] = q.y; d[4 * v + 2] = q.z; d[4 * v + 3] = q.w;
            }
          }
          __device__ __forceinline__ void touch()
          {
#pragma unroll
            for (int v = 0; v < 4; v++) asm volatile("" : "+v"(d[4 * v]), "+v"(d[4 * v + 1]), "+v"(d[4 * v + 2]), "+v"(d[4 * v + 3]) :: "memory");
          }
        };
        const uint32_t pairs = full / 2;
        if (pairs) {
          const int16_t *wp = xp - kTaps; /* the window of pair 0: the block's samples 0 .. 31 */
          PairSamples cur, np;
          cur.load_head(wp);
          cur.load_rest(wp);
          for (uint32_t p = 0; p < pairs; p++, k0 += 2) {
            np.load_head(wp + 2 * kChunk); /* samples 32 .. 35 of this window's start: the end of the pair's second chunk */
            if (p + 1 < pairs) wp += (uint64_t)2 * kChunk; /* unconditional prefetch: the last pair re-reads its own window */
            np.load_rest(wp);
            int32_t xa[kChunk / 2], xb[kChunk / 2];
#pragma unroll
            for (int j = 0; j < kChunk / 2; j++) {
              xa[j] = (int32_t)cur.d[2 + j];
              xb[j] = (int32_t)(j < 6 ? cur.d[10 + j] : np.d[j - 6]);
            }
            uint32_t wa[2] = {0, 0}, wb[2] = {0, 0};
            encode_chunk16<BITS, EMIT, true>(L, xa, lds, wa, last_qd, sq);
            encode_chunk16<BITS, EMIT, true>(L, xb, lds, wb, last_qd, sq);
            np.touch();
#pragma unroll
            for (int j = 0; j < 16; j++) cur.d[j] = np.d[j];
            if constexpr (RING && BITS == 2) { /* the pair's two dwords in one append: one carry, one boundary test */
              ring->template append<8>(perm(0, wa[0], 0x00010203), perm(0, wb[0], 0x00010203), ring_sel);
            } else if constexpr (RING && BITS == 3) { /* two chunks are twelve bytes */
              uint32_t d0, d1, d2;
              units12(wa[0], wa[1], wb[0], wb[1], d0, d1, d2);
              ring->template append<12>(d0, d1, ring_sel, d2);
            } else if constexpr (RING) {
              uint32_t d0, d1;
              chunk_bytes<BITS, CHF>(wa, c, d0, d1);
              ring->template append<kPiece>(d0, d1, ring_sel);
              chunk_bytes<BITS, CHF>(wb, c, d0, d1);
              ring->template append<kPiece>(d0, d1, ring_sel);
            }
            if constexpr (kStage) {
              stage.put(k0, wa, c);
              stage.put(k0 + 1, wb, c);
              if ((k0 & 7u) == 6u) stage.flush(body + (uint64_t)(k0 - 6u) * kOutStride, c);
            }
          }
          xp += (uint64_t)pairs * 2 * kChunk; /* the chunk behind the last pair */
          if (k0 < full) next.load(xp, ch, c);
          next.touch();
        }
      }
      if constexpr (RING && CHF == 2 && BITS == 3) {
        /* stereo 3-bit through the ring: two chunks per append, twelve bytes per lane (pair_units12) */
        for (; k0 + 2 <= full; k0 += 2) {
          uint32_t w2[2][2];
#pragma unroll
          for (int h = 0; h < 2; h++) {
            int32_t x[kN];
#pragma unroll
            for (int j = 0; j < kN; j++) x[j] = PK ? (int32_t)next.pair(j, pair_sel) : next.get(j, c);
            if (k0 + h + 1 < full) xp += (uint64_t)kChunk * ch;
            next.load(xp, ch, c);
            w2[h][0] = w2[h][1] = 0;
            encode_chunk16<BITS, EMIT, PK>(L, x, lds, w2[h], last_qd, sq);
            next.touch();
          }
          uint32_t d0, d1, d2;
          pair_units12(w2[0], w2[1], c, d0, d1, d2);
          ring->template append<12>(d0, d1, ring_sel, d2);
        }
      }
      if constexpr (RING && CHF == 2 && BITS == 2) {
        /* stereo 2-bit through the ring: two chunks per append.  A pair's sixteen bytes of two chunks are chunk k's eight
         * (lane 0 appends them) and chunk k + 1's (lane 1): each lane needs the other channel's word of ITS chunk - one swap,
         * then the same interleave as a 4-bit chunk (chunk_bytes<4, 2> with w = {word of chunk k, word of chunk k + 1}). */
        for (; k0 + 2 <= full; k0 += 2) {
          uint32_t w2[2];
#pragma unroll
          for (int h = 0; h < 2; h++) {
            int32_t x[kN];
#pragma unroll
            for (int j = 0; j < kN; j++) x[j] = PK ? (int32_t)next.pair(j, pair_sel) : next.get(j, c);
            if (k0 + h + 1 < full) xp += (uint64_t)kChunk * ch;
            next.load(xp, ch, c);
            uint32_t w[2] = {0, 0};
            encode_chunk16<BITS, EMIT, PK>(L, x, lds, w, last_qd, sq);
            next.touch();
            w2[h] = w[0];
          }
          uint32_t d0, d1;
          chunk_bytes<4, 2>(w2, c, d0, d1);
          ring->template append<8>(d0, d1, ring_sel);
        }
      }
      for (uint32_t k = k0; k < full; k++) {
        int32_t x[kN];
#pragma unroll
        for (int j = 0; j < kN; j++) x[j] = PK ? (int32_t)next.pair(j, pair_sel) : next.get(j, c);
        /* unconditional prefetch (the last iteration re-reads its own chunk), see the decoder */
        if (k + 1 < full) xp += (uint64_t)kChunk * ch;
        next.load(xp, ch, c);
        uint32_t w[2] = {0, 0};
        encode_chunk16<BITS, EMIT, PK>(L, x, lds, w, last_qd, sq);
        next.touch();
        if constexpr (RING && BITS == 3) {
          /* the one chunk two-chunk appends leave over (the last of its block): six bytes per lane, byte by byte */
          ring->carry_to_ring();
#pragma unroll
          for (int u = 0; u < 2; u++)
#pragma unroll
            for (int q = 0; q < 3; q++) ring->put_byte(ring->pos + (u * (CHF ? CHF : 1) + c) * 3u + q, w[u] >> (8 * (2 - q)));
          ring->advance(6u * (CHF ? CHF : 1));
          ring->carry_from_ring();
        } else if constexpr (RING) {
          uint32_t d0, d1;
          chunk_bytes<BITS, CHF>(w, c, d0, d1);
          ring->template append<kPiece>(d0, d1, ring_sel);
        } else if constexpr (kStage) {
          stage.put(k, w, c);
          if ((k & 7u) == 7u) stage.flush(body + (uint64_t)(k - 7u) * kOutStride * ch, c);
        } else if (EMIT) {
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
      if constexpr (kStage) { /* the chunks of the last, incomplete group */
        const uint32_t rem = full & 7u;
        for (uint32_t j = 0; j < rem; j++)
          stage.flush_one((int)j, body + (uint64_t)(full - rem + j) * kOutStride * ch + (uint64_t)c * Pack<BITS>::kChunkBytes);
      }
      done = full * kChunk;
    }
  }

  if constexpr (QUAD && PASS != kPassRmse) { /* (kPassBoth: a whole number of units needs no padding, so measuring and encoding lanes walk the same samples) */
    /* Pipelined tail (round 3).  A 1024-byte block's coded samples are 12 (4-bit: 988, 1980) or 8 (2-bit; 3-bit: one unit) past
     * a multiple of sixteen, and encode_step - one sample at a time, its two table lookups in series, a load per sample - walked
     * them at about a third of the chunk body's pace: ~3000 of a one-block kernel's 149 000 cycles.  A whole number of pack
     * units goes through the chunk body instantiated for that many samples instead: the samples are fetched together, the
     * recurrence is primed as at a block's start, the codes leave as unit bytes at the addresses the slow path writes. */
    const uint32_t rest = coded - done;
    auto tail = [&](auto nc) {
      constexpr int N = decltype(nc)::value;
      constexpr int cpw = Pack<BITS>::kCodesPerWord;
      int32_t xt[kChunk];
#pragma unroll
      for (int j = 0; j < kChunk; j++) xt[j] = j < N ? src.at(first + kTaps + done + j) : 0;
      EncodeCarry C;
      encode_prime_quad<BITS>(L, C, xt[0], lds);
      uint32_t w[2] = {0, 0};
      encode_chunk16_quad<BITS, PASS, kWide, NoFill, N>(L, C, xt, 0, lds, w, last_qd, sq);
      if (writer) {
        uint8_t *up = body + (uint64_t)(done / US) * unit_stride + (uint64_t)c * UB;
        static_for<0, N / US>([&](auto uc) {
          constexpr int u = decltype(uc)::value;
          constexpr int wi = (u * US) / cpw;                                   /* the code word the unit sits in */
          constexpr int filled = (N - wi * cpw < cpw ? N - wi * cpw : cpw) * BITS; /* bits of that word that hold codes (low bits) */
          constexpr int off = ((u * US) % cpw) * BITS;                         /* the unit's first bit, from the top of the filled part */
#pragma unroll
          for (int q = 0; q < UB; q++) up[(uint64_t)u * unit_stride + q] = (uint8_t)(w[wi] >> (filled - off - 8 * (q + 1)));
        });
      }
      done += N;
    };
    if (rest == 12u && 12 % US == 0) tail(std::integral_constant<int, 12>());
    else if (rest == 8u && 8 % US == 0) tail(std::integral_constant<int, 8>());
  }

  if constexpr (RING) {
    /* tail units through the ring, byte by byte: samples past n are zero padding - reference :592-593 */
    ring->carry_to_ring();
    uint32_t units = 0;
    for (uint32_t i = done; i < coded; i += US, units++) {
      uint32_t acc = 0;
#pragma unroll
      for (int k = 0; k < US; k++) {
        const int32_t x = i + k < coded ? src.at(first + kTaps + i + k) : 0;
        acc = (acc << BITS) | encode_step<BITS>(L, x, lds, last_qd);
      }
#pragma unroll
      for (int k = 0; k < UB; k++) ring->put_byte(ring->pos + units * unit_stride + c * UB + k, acc >> (8 * (UB - 1 - k)));
    }
    ring->advance(units * unit_stride);
    ring->carry_from_ring();
  } else if constexpr (kBurstStores<BITS, CHF, EMIT> && !QUAD) {
    /* tail units (one byte per lane and unit, at most seven) are collected and stored with what the
     * chunk loop left over, back to back: the block's last granule is written once, not unit by unit */
    uint64_t tail = 0;
    uint32_t units = 0;
    for (uint32_t i = done; i < coded; i += US, units++) {
      uint32_t acc = 0;
#pragma unroll
      for (int k = 0; k < US; k++) {
        const int32_t x = i + k < coded ? src.at(first + kTaps + i + k) : 0;
        acc = (acc << BITS) | encode_step<BITS>(L, x, lds, last_qd);
      }
      tail |= (uint64_t)(acc & 0xFFu) << (8 * units);
    }
    if (defer3) put_carry(burst_gp - 11, burst_carry); /* what the last whole group (or the header) left */
    if (burst_rest == 3) put(burst_gp, burst_r0);
    if (burst_rest >= 2) put(burst_gp + 16 * (burst_rest - 2), burst_r1);
    if (burst_rest >= 1) put(burst_gp + 16 * (burst_rest - 1), burst_r2);
    uint8_t *up = body + (uint64_t)(done / US) * unit_stride + (uint64_t)c * UB;
    for (uint32_t u = 0; u < units; u++, up += unit_stride, tail >>= 8) up[0] = (uint8_t)tail;
  } else if constexpr (PASS == kPassBoth) {
    /* both at once: lanes that measure stop at the last real sample, lanes that encode pad the last unit
     * with zero samples (the four taps of a recurrence agree, so the DPP traffic of a step stays whole) */
    uint8_t *up = body + (uint64_t)(done / US) * unit_stride + (uint64_t)c * UB;
    for (uint32_t i = done; i < coded; i += US, up += unit_stride) {
      uint32_t acc = 0;
#pragma unroll
      for (int k = 0; k < US; k++) {
        const bool real = i + k < coded;
        uint32_t code = 0;
        if (real || pad) {
          int32_t qd;
          code = encode_step<BITS>(L, real ? src.at(first + kTaps + i + k) : 0, lds, qd);
          last_qd = qd;
          if (real) sq += wrapped_square(qd);
        }
        acc = (acc << BITS) | code;
      }
      if (writer) {
#pragma unroll
        for (int k = 0; k < UB; k++) up[k] = (uint8_t)(acc >> (8 * (UB - 1 - k)));
      }
    }
  } else if (EMIT) { /* tail units: samples past n are zero padding - reference :592-593 */
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
 * Trial search AND encode of one block on the "dual" mapping (quad mapping with a second group of four
 * lanes per channel, used when lanes are idle anyway) - reference src/aad_encoder.c:470-562, 565-727.
 * The reference evaluates, per channel and block,
 *   probe:  RMSE of the current block from the carried state S                       (1 pass)
 *   chain:  trials x ([previous block] + current block), each from where the last ended;
 *           the state C_i in front of the i-th pass over the current block is candidate i
 * one after the other, then encodes the block from the best of {S, C_1 .. C_t}: 2 + 2t passes of
 * latency (1 + t for a first block, where probe and first trial coincide).  An encode pass is the same
 * recurrence as a measuring pass, but from the candidate with its weights cut to what the 16-bit header
 * fields carry (write_block_header) and with zero padding at the end - so it cannot be taken from the
 * measurement, yet it needs nothing but the candidate: it can run BESIDE the chain instead of behind it.
 *   role 0 runs the chain (measuring passes);
 *   role 1 runs the probe, then encodes S and every candidate as soon as it exists - C_t while role 0
 *          is still measuring it - each into a place no channel of the stream currently has its best
 *          encode in: the image or one of two scratch slots (the channels of a stream share their
 *          interleaved bytes, so a place is per stream, while the winner is per channel);
 * every pass is ONE call of the both-at-once pass (kPassBoth), the roles differ in state, window and
 * pointers only.  When the chain ends, so has every encode: 2t passes (3 for t = 1; t for a first block),
 * then the winners' bytes are moved from their slots into the image, channel by channel (12-byte
 * pieces under byte masks), which costs a few microseconds per block.  The winner is picked exactly as
 * the reference does (strict >, probe first, trials in order).
 */
template <int CHF>
__device__ __forceinline__ uint32_t role_swap(uint32_t v)
{ /* the same tap of the same channel in the other role: CHF lanes away inside an aligned group of four */
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CHF == 2 ? 0x4E /* quad_perm [2,3,0,1] */ : 0xB1 /* [1,0,3,2] */, 0xF, 0xF, false);
}
template <int CHF>
__device__ __forceinline__ double role_swap_f64(double v)
{
  const uint64_t u = (uint64_t)__double_as_longlong(v);
  const uint64_t r = ((uint64_t)role_swap<CHF>((uint32_t)(u >> 32)) << 32) | role_swap<CHF>((uint32_t)u);
  return __longlong_as_double((long long)r);
}

template <int BITS, int CHF, bool MS>
__device__ __forceinline__ void encode_block_dual(Lane &F, int32_t &last_qd, const SampleSource<MS> &src, uint64_t first, uint32_t n,
                                                  uint32_t spb, uint32_t trials, uint32_t c, uint32_t tap, uint32_t role,
                                                  uint8_t *img, uint8_t *slots, uint32_t slot_bytes, const char *lds)
{
  static_assert(kEncTM && (CHF == 1 || CHF == 2), "tap-major quads, mono or stereo");
  constexpr uint32_t ch = CHF;
  constexpr int US = Pack<BITS>::kUnitSamples, UB = Pack<BITS>::kUnitBytes;
  enum { kLocImage = 0, kLocA = 1, kLocB = 2, kLocNone = 3 };
  uint8_t *const slot_a = slots, *const slot_b = slots + slot_bytes, *const slot_t = slots + 2u * (uint64_t)slot_bytes;
  auto place = [&](uint32_t loc) -> uint8_t * { return loc == kLocImage ? img : (loc == kLocA ? slot_a : slot_b); };
  const bool have_prev = first >= spb; /* the same for every lane of the launch */
  const uint32_t chain_passes = trials * (have_prev ? 2u : 1u);
  const uint32_t passes = have_prev && chain_passes < 3u ? 3u : chain_passes; /* probe, S, C_1 need three turns of role 1 */

  QuadLane run = to_quad(F, tap); /* the chain's state; every lane of both roles keeps the same books */
  QuadLane held = run;            /* with a previous block: S for pass 1, C_1 for pass 2 */
  QuadLane best_end = run;
  int32_t best_qd = last_qd;
  double best_rmse = 0.0, held_rmse = 0.0;
  uint32_t best_loc = kLocNone;

  /* the first four samples of the two windows the passes start from, fetched once (every pass seeds its
   * history from them: one round trip to memory per block instead of one per pass) */
  Lane seed_cur = {0, 0, 0, 0, 0, 0, 0, 0, 0}, seed_prev = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  seed_history(seed_cur, src, first, n);
  if (have_prev) seed_history(seed_prev, src, first - spb, spb);

  for (uint32_t p = 0; p < passes; p++) {
    /* ---- this lane's part in pass p */
    const bool chain_active = p < chain_passes;
    const bool chain_on_prev = have_prev && (p & 1u) == 0;
    const bool enc_pass = !have_prev || p == 1 || p == 2 || (p >= 3 && (p & 1u)); /* role 1 encodes (else: probe / nothing due) */
    QuadLane from1 = run;
    if (have_prev && p == 1) {
      from1 = held; /* S; C_1 - the chain's state now - waits for pass 2 */
      held = run;
    } else if (have_prev && p == 2) {
      from1 = held;
    }
    const bool enc = role != 0 && enc_pass;
    const bool on_prev = role == 0 && chain_on_prev && chain_active;
    /* a place no channel of the stream has its best encode in (the same answer on both channels' lanes) */
    const uint32_t other_loc = CHF == 2 ? pair_swap<false>(best_loc, c) : best_loc;
    uint32_t target = kLocImage;
    if (best_loc == kLocImage || other_loc == kLocImage) target = (best_loc == kLocA || other_loc == kLocA) ? kLocB : kLocA;
    uint8_t *const base = enc ? place(target) : slot_t;
    const uint64_t wfirst = on_prev ? first - spb : first;
    const uint32_t wn = on_prev ? spb : n;

    QuadLane Q = role != 0 ? from1 : run;
    Lane Ff = from_quad<kEncTM>(Q);
    Ff.h0 = on_prev ? seed_prev.h0 : seed_cur.h0;
    Ff.h1 = on_prev ? seed_prev.h1 : seed_cur.h1;
    Ff.h2 = on_prev ? seed_prev.h2 : seed_cur.h2;
    Ff.h3 = on_prev ? seed_prev.h3 : seed_cur.h3;
    Lane Fm = Ff;
    write_block_header(Fm, base + (uint64_t)c * kBlockHeaderBytesPerCh, enc && tap == 0);
    if (enc) Ff = Fm; /* an encode starts from the weights the header carries */
    Q = to_quad(Ff, tap);
    int32_t qd = last_qd;
    /* What the wave needs from this pass (the same for all of its lanes): with a previous block, pass 0 is
     * probe + previous block - sums only - and the even passes from 2 on are previous block + an encode -
     * codes only (a whole block has no padding, so the chain's pass ends where a measuring one does); the
     * leaner bodies save 3.2 and 1.6 issue slots per sample there. */
    const bool sums_only = have_prev && p == 0, codes_only = have_prev && p >= 2 && (p & 1u) == 0;
    uint8_t *const codes_at = base + (uint64_t)kBlockHeaderBytesPerCh * ch;
    int64_t sum = 0;
    if (sums_only) sum = run_block<BITS, CHF, MS, true, kPassRmse>(Q, src, wfirst, wn, ch, c, false, nullptr, lds, qd);
    else if (codes_only) (void)run_block<BITS, CHF, MS, true, kPassEncode>(Q, src, wfirst, wn, ch, c, tap == 0, codes_at, lds, qd);
    else sum = run_block<BITS, CHF, MS, true, kPassBoth>(Q, src, wfirst, wn, ch, c, tap == 0, codes_at, lds, qd, false, 0, enc);
    const double r = wn < (uint32_t)kTaps ? 0.0 : sqrt((double)sum / (double)wn);
    AAD_PHASE_MARK(blockIdx.x == 0 && threadIdx.x == 0);

    /* ---- both roles' results to every lane */
    const uint32_t ow = role_swap<CHF>((uint32_t)Q.w), oh = role_swap<CHF>((uint32_t)Q.h), oi = role_swap<CHF>((uint32_t)Q.idxb);
    const uint32_t oq = role_swap<CHF>((uint32_t)qd);
    const double orr = role_swap_f64<CHF>(r);
    const double r_chain = role != 0 ? orr : r, r_probe = role != 0 ? r : orr;
    QuadLane enc_end = Q;
    enc_end.w = role != 0 ? Q.w : (int32_t)ow;
    enc_end.h = role != 0 ? Q.h : (int32_t)oh;
    enc_end.idxb = role != 0 ? Q.idxb : (int32_t)oi;
    const int32_t enc_qd = role != 0 ? qd : (int32_t)oq;
    if (chain_active) {
      run.w = role != 0 ? (int32_t)ow : Q.w;
      run.h = role != 0 ? (int32_t)oh : Q.h;
      run.idxb = role != 0 ? (int32_t)oi : Q.idxb;
    }

    /* ---- the books: strict >, probe first, trials in order (reference :520-552) */
    bool take = false;
    if (!have_prev) { /* pass p measured candidate p + 1 and encoded it; the first one doubles as the probe */
      take = p == 0 || best_rmse > r_chain;
      if (take) best_rmse = r_chain;
    } else if (p == 0) {
      best_rmse = r_probe;
    } else if (p == 1) { /* S is encoded (the image); C_1 is measured, its encode comes next */
      take = true;
      held_rmse = r_chain;
    } else if (p == 2) {
      take = best_rmse > held_rmse;
      if (take) best_rmse = held_rmse;
    } else if (p & 1u) { /* measured and encoded in this pass */
      take = best_rmse > r_chain;
      if (take) best_rmse = r_chain;
    }
    if (take) {
      best_loc = target;
      best_end.w = enc_end.w;
      best_end.h = enc_end.h;
      best_end.idxb = enc_end.idxb;
      best_qd = enc_qd;
    }
  }

  /* ---- winners that are not in the image move there, channel by channel */
  /* the passes' stores, made by other lanes of this wave, are read below: wave (workgroup) scope is
   * enough - an agent-scope fence writes the L2 back, ~55 us per block on a 1000-stream batch */
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  const uint32_t other_loc = CHF == 2 ? pair_swap<false>(best_loc, c) : best_loc;
  const uint32_t loc0 = c == 0 ? best_loc : other_loc, loc1 = c == 0 ? other_loc : best_loc;
  if (loc0 != kLocImage || loc1 != kLocImage) {
    constexpr uint32_t kQ = 2u * CHF, kGroup = 4u * kQ; /* quads and lanes of one stream */
    const uint32_t id = tap * kQ + (threadIdx.x & (kQ - 1u));
    const uint32_t coded = n > (uint32_t)kTaps ? n - kTaps : 0;
    const uint32_t used = (kBlockHeaderBytesPerCh + (coded + US - 1) / US * UB) * ch;
    const uint8_t *const s0 = place(loc0), *const s1 = place(loc1);
    const uint32_t pieces = used / 12u;
    /* channel 0's bytes of piece q: 36 header bytes = pieces 0-2 (18 per channel), then whole units alternate
     * (1 byte, or 3 for 3-bit codes: a 6-byte pattern, twice per piece) */
    auto mask0 = [&](uint32_t q) -> u32x3 {
      if (CHF == 1 || q == 0) return u32x3{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
      if (q == 1) return u32x3{0xFFFFFFFFu, 0x0000FFFFu, 0u};
      if (q == 2) return u32x3{0u, 0u, 0u};
      if (UB == 1) return u32x3{0x00FF00FFu, 0x00FF00FFu, 0x00FF00FFu};
      return u32x3{0x00FFFFFFu, 0xFFFF0000u, 0x000000FFu};
    };
    auto fetch = [&](const uint8_t *from, uint32_t q) { return reinterpret_cast<const U32x3 *>(from + 12u * q)->v; };
    auto merged = [&](const u32x3 &v0, const u32x3 &v1, uint32_t q) {
      const u32x3 m = mask0(q);
      return u32x3{(v0.x & m.x) | (v1.x & ~m.x), (v0.y & m.y) | (v1.y & ~m.y), (v0.z & m.z) | (v1.z & ~m.z)};
    };
    /* three pieces per round, their loads issued together: a round costs one trip to memory, not three */
    for (uint32_t q = id; q < pieces; q += 3u * kGroup) {
      const uint32_t qb = q + kGroup < pieces ? q + kGroup : q, qc = q + 2u * kGroup < pieces ? q + 2u * kGroup : q;
      const u32x3 a0 = fetch(s0, q), a1 = fetch(s1, q), b0 = fetch(s0, qb), b1 = fetch(s1, qb), c0 = fetch(s0, qc), c1 = fetch(s1, qc);
      const u32x3 oa = merged(a0, a1, q), ob = merged(b0, b1, qb), oc = merged(c0, c1, qc);
      reinterpret_cast<U32x3 *>(img + 12u * q)->v = oa;
      if (qb != q) reinterpret_cast<U32x3 *>(img + 12u * qb)->v = ob;
      if (qc != q) reinterpret_cast<U32x3 *>(img + 12u * qc)->v = oc;
    }
    for (uint32_t o = pieces * 12u + id; o < used; o += kGroup) { /* at most 11 bytes, all of them code bytes for stereo */
      const uint32_t owner = CHF == 2 ? ((o - kBlockHeaderBytesPerCh * ch) / UB) & 1u : 0u;
      img[o] = (owner ? s1 : s0)[o];
    }
  }
  F = from_quad<kEncTM>(best_end);
  last_qd = best_qd;
}

/*
 * Stream-parallel encode (reference src/aad_encoder.c:814-891 with EncodeBlock :565-727 and the
 * optional trial search :470-562 inlined).  lane = (stream, channel).
 */
template <int BITS, int CHF, bool MS, bool QUAD, bool TRIALS, bool DUAL = false, bool RING = false>
__global__ void __launch_bounds__(256) encode_streams_kernel(EncodeArgs a)
{
  static_assert(!QUAD || CHF != 0, "the quad mapping exists for the mono / stereo fast paths");
  static_assert(!DUAL || (QUAD && TRIALS), "the dual mapping is the trial search on the quad mapping");
  static_assert(!RING || kRingable<BITS, CHF, QUAD>, "the byte ring: dense mono / stereo encoders");
  __shared__ __attribute__((aligned(16))) char lds[kLdsBytesEncoderStatic<BITS, CHF, QUAD, RING>]; /* dense and quad encoders share the wide table; dense: + code staging */
  extern __shared__ __attribute__((aligned(16))) char ring_lds[]; /* RING: the rows' byte rings, kLdsRingBytesPerWave per wave (then the occupancy pad, unused) */
  AAD_PHASE_MARK(blockIdx.x == 0 && threadIdx.x == 0);
  stage_tables<BITS, true, kWideStepShift, true>(lds);
  AAD_PHASE_MARK(blockIdx.x == 0 && threadIdx.x == 0);

  const uint32_t ch = CHF ? CHF : a.channels;
  const uint64_t thread = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  /* dual: with trials on the quad mapping every stream owns 2 x CHF recurrence slots, laid out
   * [role 0: ch 0 .. CHF-1][role 1: ch 0 .. CHF-1] so that a stereo pair stays adjacent */
  constexpr uint32_t kQuadsPerStream = DUAL ? 2u * (CHF ? CHF : 1) : 1u;
  /* quad: sixteen recurrence slots per wave, four lanes each (lane layout: aad_device.hip.h) */
  const uint64_t slot = QUAD ? (thread >> 6) * 16u + quad_slot<kEncTM>() : thread;
  const uint32_t role = DUAL ? (uint32_t)(slot % kQuadsPerStream) / (CHF ? CHF : 1) : 0u;
  const uint64_t lane = DUAL ? slot / kQuadsPerStream * (CHF ? CHF : 1) + slot % (CHF ? CHF : 1)
                             : slot; /* index of the (stream, channel) recurrence */
  const uint32_t tap = QUAD ? quad_tap<kEncTM>() : 0u;
  const bool writer = tap == 0 && role == 0;         /* quad: all four lanes hold the codes, one stores them */
  if (lane >= (uint64_t)a.num_streams * ch) return; /* whole quads / stereo pairs / role groups leave together */
  const uint32_t s = (uint32_t)(lane / ch), c = (uint32_t)(lane % ch);
  const StreamDesc sd = a.uni.enabled ? uniform_stream(a.uni, s) : a.streams[s];
  const SampleSource<MS> src = {a.pcm + sd.pcm_offset, ch, c, sd.num_samples};
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

  ByteRing<(CHF ? CHF : 1)> ring;
  if constexpr (RING) {
    ring.init(ring_lds + (threadIdx.x >> 6) * kLdsRingBytesPerWave<CHF>, (threadIdx.x & 63u) / CHF, out,
              sd.data_size > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)sd.data_size, c);
    /* file header - reference src/aad_encoder.c:190-214 - through the ring, byte by byte (once per stream) */
    const uint32_t encoded = total - a.lead_frames;
    if (c == 0) {
#pragma unroll
      for (uint32_t i = 0; i < (uint32_t)kFileHeaderBytes; i++) {
        uint32_t b = a.header_template[i];
        if (i >= 14 && i < 18) b = (encoded >> (8u * (17u - i))) & 0xFFu; /* the sample count, big-endian in bytes 14..17 */
        ring.put_byte(i, b);
      }
    }
    ring.advance(kFileHeaderBytes);
    ring.carry_from_ring();
  }
  if (!RING && c == 0 && writer) { /* file header - reference src/aad_encoder.c:190-214 */
    const uint32_t encoded = total - a.lead_frames;
    /* 31 bytes as seven dwords and three bytes; the sample count is big-endian in bytes 14..17 */
    const uint32_t *t = reinterpret_cast<const uint32_t *>(a.header_template);
#pragma unroll
    for (int i = 0; i < 7; i++) {
      uint32_t d = t[i];
      if (i == 3) d = (d & 0x0000FFFFu) | ((encoded >> 24) << 16) | (((encoded >> 16) & 0xFFu) << 24);
      if (i == 4) d = (d & 0xFFFF0000u) | ((encoded >> 8) & 0xFFu) | ((encoded & 0xFFu) << 8);
      reinterpret_cast<U32 *>(out + 4 * i)->v = d;
    }
    reinterpret_cast<U16 *>(out + 28)->v = (uint16_t)t[7];
    out[30] = (uint8_t)(t[7] >> 16);
  }

  uint64_t block_off = kFileHeaderBytes;
  for (uint64_t first = a.lead_frames; first < total; first += spb, block_off += a.block_size) {
    const uint32_t n = total - first < spb ? (uint32_t)(total - first) : spb;
    S L;
    AAD_PHASE_MARK(blockIdx.x == 0 && threadIdx.x == 0);
    if constexpr (DUAL) { /* search and encode side by side, see encode_block_dual */
      encode_block_dual<BITS, CHF, MS>(F, last_qd, src, first, n, spb, a.trials, c, tap, role, out + block_off,
                                       a.trial_scratch + (uint64_t)s * 3u * a.trial_slot_bytes, a.trial_slot_bytes, lds);
    } else {
    if constexpr (TRIALS) { /* reference src/aad_encoder.c:863-871; a separate instantiation so that the
                             * trial-free kernel does not carry the search's registers */
      if constexpr (QUAD) L = to_quad(F, tap); else L = F;
      search_best_lane<BITS, CHF, MS, QUAD>(L, src, first, n, spb, a.trials, ch, c, tap, lds);
      if constexpr (QUAD) F = from_quad<kEncTM>(L); else F = L;
    }
    seed_history(F, src, first, n);
    uint8_t *body = out + block_off + (uint64_t)kBlockHeaderBytesPerCh * ch;
    /* dense stereo 4-bit: code bytes 3 bytes into a 64-byte granule - the channel-1 lane holds the
     * header's last three bytes back for the first burst of code bytes (run_block) */
    if constexpr (RING) {
      /* the block header's eighteen bytes per channel behind what the row has written so far, then the codes */
      uint32_t words[5];
      (void)write_block_header(F, nullptr, false, false, words);
      ring.carry_to_ring();
#pragma unroll
      for (uint32_t i = 0; i < (uint32_t)kBlockHeaderBytesPerCh; i++)
        ring.put_byte(ring.pos + c * kBlockHeaderBytesPerCh + i, words[i >> 2] >> (8u * (i & 3u)));
      ring.advance(kBlockHeaderBytesPerCh * (CHF ? CHF : 1));
      ring.carry_from_ring();
      L = F;
      (void)run_block<BITS, CHF, MS, QUAD, kPassEncode, true>(L, src, first, n, ch, c, writer, body, lds, last_qd, false, 0, true, &ring);
      F = L;
    } else {
    const bool defer3 = !QUAD && kBurstStores<BITS, CHF, true> && c == 1 && (reinterpret_cast<uintptr_t>(body) & 63u) == 3u;
    const uint32_t deferred = write_block_header(F, out + block_off + (uint64_t)c * kBlockHeaderBytesPerCh, writer, defer3);
    if constexpr (QUAD) L = to_quad(F, tap); else L = F;
    AAD_PHASE_MARK(blockIdx.x == 0 && threadIdx.x == 0);
    (void)run_block<BITS, CHF, MS, QUAD, true>(L, src, first, n, ch, c, writer, body, lds, last_qd, defer3, deferred);
    if constexpr (QUAD) F = from_quad<kEncTM>(L); else F = L;
    }
    }
    AAD_PHASE_MARK(blockIdx.x == 0 && threadIdx.x == 0);
  }
  if constexpr (RING) ring.finish(); /* the stream's last, incomplete sector */

  if (a.state_out && writer) {
    LaneStateRecord r;
    r.weight[0] = F.w0; r.weight[1] = F.w1; r.weight[2] = F.w2; r.weight[3] = F.w3;
    r.history[0] = F.h0; r.history[1] = F.h1; r.history[2] = F.h2; r.history[3] = F.h3;
    r.stepsize_index = F.idxb - kIdxBias;
    r.quantize_error = last_qd;
    a.state_out[lane] = r;
  }
}

} /* namespace aad */

#endif /* AAD_ENCODE_HIP_H */
