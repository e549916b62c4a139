/*
 * aad_encode_x2.hip.h - stereo encoder for lane-starved batches: ONE quad (four lanes, lane t = tap t,
 * tap-major layout) runs BOTH channels of a stream, their instruction streams interleaved by hand.
 *
 * Why.  A lone gfx950 wave issues one instruction every four cycles at best, and the encoder's
 * recurrence (reference src/aad_encoder.c:343-410) is one long dependent chain per sample:
 * quantise -> step index -> LDS lookup of the next step record -> ... and, in parallel, dequantise ->
 * reconstruct -> LMS -> history shift -> product -> two DPP adds -> next difference.  With one
 * recurrence per quad the counters (profiles/r02_*) showed a wave issuing for 77 % of its cycles and
 * sitting in s_waitcnt / s_nop for the rest: the ~100-cycle LDS round trip of the step record and the
 * DPP wait states are latencies of ONE chain with nothing else to issue.  The two channels of a
 * stereo stream are independent chains (no M/S), so a quad that carries both always has the other
 * channel's instructions to issue into those holes: per sample pair 56 arithmetic slots + one
 * s_waitcnt and no s_nop at all, against 2 x (30 + ~9 slots of waiting).
 *
 * It also removes every cross-lane step of the stereo framing: the 64 bytes a chunk of 16 frames
 * occupies are loaded once and the subtract reads channel 0 / channel 1 straight out of the low /
 * high half of each frame dword (v_sub_u32_sdwa WORD_0 / WORD_1); both channels' code words are in
 * the same lane, so the L/R byte interleave is two v_perm_b32 per 8 bytes and one 16-byte store.
 *
 * Used by the host for the quad mapping when channels == 2 and M/S is off (the BASELINE shape);
 * mono, M/S and >2 channels keep encode_streams_kernel.  Same arithmetic, instruction for
 * instruction, as encode_chunk16_quad; the trial search (src/aad_encoder.c:470-562) is the same
 * schedule as search_best_lane / search_best_lane_dual with both channels in every pass.
 */
#ifndef AAD_ENCODE_X2_HIP_H
#define AAD_ENCODE_X2_HIP_H

#include "aad_encode.hip.h"

namespace aad {

constexpr bool kX2TM = true; /* tap-major quads (aad_device.hip.h) */

/* sixteen frames of both channels as loaded: dword k = {L_k (low half), R_k (high half)} */
struct FrameChunk {
  uint32_t d[kChunk];
  __device__ __forceinline__ void load(const int16_t *x)
  {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const u32x4 a = reinterpret_cast<const U32x4 *>(x + 8 * k)->v;
      d[4 * k] = a.x; d[4 * k + 1] = a.y; d[4 * k + 2] = a.z; d[4 * k + 3] = a.w;
    }
  }
  __device__ __forceinline__ void clear()
  {
#pragma unroll
    for (int k = 0; k < kChunk; k++) d[k] = 0;
  }
};

template <int CH>
__device__ __forceinline__ int32_t frame_sample(uint32_t frame) { return CH == 0 ? (int32_t)(int16_t)frame : (int32_t)frame >> 16; }

template <int BITS>
__device__ __forceinline__ void encode_prime_x2(QuadLane &A, QuadLane &B, EncodeCarry &CA, EncodeCarry &CB, uint32_t frame0, const char *lds)
{
  CA.e = *reinterpret_cast<const u32x3 *>(lds + kLdsWideOff + wide_addr(A.idxb));
  CB.e = *reinterpret_cast<const u32x3 *>(lds + kLdsWideOff + wide_addr(B.idxb));
  CA.p = predict<kX2TM>(A);
  CB.p = predict<kX2TM>(B);
  CA.d = frame_sample<0>(frame0) - CA.p;
  CB.d = frame_sample<1>(frame0) - CB.p;
  CA.m = CA.d >> 31;
  CB.m = CB.d >> 31;
  CA.f = (float)CA.d;
  CB.f = (float)CB.d;
}

/*
 * Sixteen encoder steps of both channels.  x: this chunk's frames, xn0: the first frame of the next
 * chunk.  Per sample the two chains are issued as
 *     [A: quantise, index, start lookup] [B: same] [A: dequantise .. product] [B: same]
 *     [A and B alternating: pack the code, DPP add, DPP add, next difference]
 * so that a DPP add always has at least two instructions of the other channel between it and the
 * write of its operand, and a step record has ~45 instructions (~190 cycles) to arrive.
 */
template <int BITS, bool EMIT>
__device__ __forceinline__ void encode_chunk16_x2(QuadLane &A, QuadLane &B, EncodeCarry &CA, EncodeCarry &CB, const uint32_t *x, uint32_t xn0,
                                                  const char *lds, uint32_t *wa, uint32_t *wb, int32_t &qda_out, int32_t &qdb_out,
                                                  int64_t &sqa, int64_t &sqb)
{
  u32x3 ea = CA.e, eb = CB.e;
  int32_t pa = CA.p, pb = CB.p, ma = CA.m, mb = CB.m;
  float fa = CA.f, fb = CB.f;
  static_for<0, kChunk>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    /* both step records were asked for a whole sample ago: one wait covers the pair */
    __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0) */
    /* A: quantise, new step index, start the lookup of the next record */
    const uint32_t mag_a = min((uint32_t)__builtin_fmaf(__builtin_fabsf(fa), __uint_as_float(ea.z), __uint_as_float(ea.y)), Pack<BITS>::kMagMax);
    const uint32_t step2_a = ea.x;
    A.idxb = clamp_idx(A.idxb + index_delta_arith<BITS>(mag_a));
    ea = *reinterpret_cast<const u32x3 *>(lds + kLdsWideOff + wide_addr(A.idxb));
    __builtin_amdgcn_sched_barrier(0);
    /* B: the same */
    const uint32_t mag_b = min((uint32_t)__builtin_fmaf(__builtin_fabsf(fb), __uint_as_float(eb.z), __uint_as_float(eb.y)), Pack<BITS>::kMagMax);
    const uint32_t step2_b = eb.x;
    B.idxb = clamp_idx(B.idxb + index_delta_arith<BITS>(mag_b));
    eb = *reinterpret_cast<const u32x3 *>(lds + kLdsWideOff + wide_addr(B.idxb));
    __builtin_amdgcn_sched_barrier(0);
    /* A: dequantise (one high multiply, see encode_chunk16_quad), reconstruct, LMS, history shift, product */
    const int32_t q_a = (int32_t)__umulhi(step2_a, (mag_a << (33 - BITS)) | (1u << (32 - BITS)));
    const int32_t qd_a = (q_a ^ ma) - ma;
    const int32_t y_a = clip16(qd_a + pa);
    lms_and_shift<kShiftBankMask>(A, qd_a, y_a);
    uint32_t sa = (uint32_t)A.h * (uint32_t)A.w + A.round;
    pin(sa);
    __builtin_amdgcn_sched_barrier(0);
    /* B: the same */
    const int32_t q_b = (int32_t)__umulhi(step2_b, (mag_b << (33 - BITS)) | (1u << (32 - BITS)));
    const int32_t qd_b = (q_b ^ mb) - mb;
    const int32_t y_b = clip16(qd_b + pb);
    lms_and_shift<kShiftBankMask>(B, qd_b, y_b);
    uint32_t sb = (uint32_t)B.h * (uint32_t)B.w + B.round;
    pin(sb);
    __builtin_amdgcn_sched_barrier(0);
    /* A and B alternating: code / square, DPP add, accumulate, DPP add */
    uint32_t ta, tb;
    if (EMIT) {
      ta = ((uint32_t)ma & Pack<BITS>::kSign) | mag_a; /* v_and_or_b32 */
      pin(ta);
      tb = ((uint32_t)mb & Pack<BITS>::kSign) | mag_b;
      pin(tb);
    } else {
      ta = (uint32_t)qd_a * (uint32_t)qd_a; /* wraps in int32 like the reference's product (src/aad_encoder.c:461) */
      pin(ta);
      tb = (uint32_t)qd_b * (uint32_t)qd_b;
      pin(tb);
    }
    sa += quad_dpp<kDppRowRor4>(sa);
    pin(sa);
    sb += quad_dpp<kDppRowRor4>(sb);
    pin(sb);
    if (EMIT) {
      uint32_t &acc_a = wa[j / Pack<BITS>::kCodesPerWord], &acc_b = wb[j / Pack<BITS>::kCodesPerWord];
      acc_a = (acc_a << BITS) | ta; /* v_lshl_or_b32 */
      pin(acc_a);
      acc_b = (acc_b << BITS) | tb;
      pin(acc_b);
    } else {
      sqa += (int64_t)(int32_t)ta;
      sqb += (int64_t)(int32_t)tb;
    }
    sa += quad_dpp<kDppRowRor8>(sa);
    pin(sa);
    sb += quad_dpp<kDppRowRor8>(sb);
    pa = (int32_t)sa >> 15;
    pb = (int32_t)sb >> 15;
    const uint32_t frame = j + 1 < kChunk ? x[j + 1 < kChunk ? j + 1 : j] : xn0;
    const int32_t da = frame_sample<0>(frame) - pa;
    const int32_t db = frame_sample<1>(frame) - pb;
    fa = (float)da;
    fb = (float)db;
    ma = da >> 31;
    mb = db >> 31;
    pin(ma);
    pin(mb);
    if (j + 1 == kChunk) {
      qda_out = qd_a;
      qdb_out = qd_b;
    }
    __builtin_amdgcn_sched_barrier(0);
  });
  CA.e = ea;
  CB.e = eb;
  CA.p = pa;
  CB.p = pb;
  CA.m = ma;
  CB.m = mb;
  CA.f = fa;
  CB.f = fb;
}

/* the packed codes of one chunk of both channels, interleaved per pack unit (reference
 * src/aad_encoder.c:663-718): wa / wb = big-endian code words of channel 0 / 1 */
template <int BITS>
__device__ __forceinline__ void store_chunk_codes_x2(uint8_t *up, const uint32_t *wa, const uint32_t *wb)
{
  if (BITS == 4) { /* a0 b0 a1 b1 a2 b2 a3 b3 | a4 b4 .. from A = a0 a1 a2 a3, B = b0 b1 b2 b3 (big-endian words) */
    u32x4 v;
    v.x = perm(wa[0], wb[0], 0x02060307);
    v.y = perm(wa[0], wb[0], 0x00040105);
    v.z = perm(wa[1], wb[1], 0x02060307);
    v.w = perm(wa[1], wb[1], 0x00040105);
    reinterpret_cast<U32x4 *>(up)->v = v;
  } else if (BITS == 2) { /* one word of sixteen codes per channel: a0 b0 a1 b1 | a2 b2 a3 b3 */
    u32x2 v;
    v.x = perm(wa[0], wb[0], 0x02060307);
    v.y = perm(wa[0], wb[0], 0x00040105);
    reinterpret_cast<U32x2 *>(up)->v = v;
  } else { /* two 3-byte units per channel: a0 a1 a2 b0 b1 b2 | a3 a4 a5 b3 b4 b5 from A = 0 a0 a1 a2, B = 0 b0 b1 b2 */
    u32x3 v;
    v.x = perm(wa[0], wb[0], 0x02040506);                                   /* a0 a1 a2 b0 */
    v.y = perm(perm(wa[0], wb[0], 0x0c0c0001), wa[1], 0x01020504);          /* b1 b2 a3 a4 */
    v.z = perm(wa[1], wb[1], 0x00010204);                                   /* a5 b3 b4 b5 */
    reinterpret_cast<U32x3 *>(up)->v = v;
  }
}

struct StereoSource {
  const int16_t *x; /* the stream's frames */
  __device__ __forceinline__ int32_t at(uint32_t c, uint64_t i) const { return x[i * 2 + c]; }
};

/* history seeded with the block's first four samples - reference src/aad_encoder.c:606-615 */
__device__ __forceinline__ void seed_history_x2(QuadLane &A, QuadLane &B, const StereoSource &src, uint64_t first, uint32_t n, uint32_t tap)
{
  const uint32_t k = 3u - tap; /* tap t holds the sample that is t steps old */
  A.h = k < n ? src.at(0, first + k) : 0;
  B.h = k < n ? src.at(1, first + k) : 0;
}

/* one pass of both recurrences over the coded samples of a block, see run_block */
template <int BITS, bool EMIT>
__device__ __forceinline__ void run_block_x2(QuadLane &A, QuadLane &B, const StereoSource &src, uint64_t first, uint32_t n, bool writer,
                                             uint8_t *body, const char *lds, int32_t &last_qd_a, int32_t &last_qd_b, int64_t &sqa, int64_t &sqb)
{
  constexpr int US = Pack<BITS>::kUnitSamples, UB = Pack<BITS>::kUnitBytes;
  constexpr uint32_t kOutStride = Pack<BITS>::kChunkBytes * 2;
  const uint32_t coded = n > (uint32_t)kTaps ? n - kTaps : 0;
  const uint32_t full = coded / kChunk;
  sqa = sqb = 0;
  {
    /* three frame sets rotate (cf. the split decoder): chunk k is consumed from one, chunk k+1 - whose
     * first frame the last step of chunk k looks ahead to - sits in the second, chunk k+2 is in
     * flight into the third */
    const int16_t *rp = src.x + (first + kTaps) * 2;
    FrameChunk b0, b1, b2;
    b0.clear();
    b1.clear();
    b2.clear();
    EncodeCarry CA, CB;
    if (full) {
      b0.load(rp);
      if (full > 1) rp += kChunk * 2;
      b1.load(rp);
      encode_prime_x2<BITS>(A, B, CA, CB, b0.d[0], lds);
    }
    auto one = [&](uint32_t k, const FrameChunk &cur, const FrameChunk &ahead, FrameChunk &incoming) {
      if (k + 2 < full) rp += kChunk * 2; /* prefetch chunk k+2 (clamped to the last full one) */
      incoming.load(rp);
      uint32_t wa[2] = {0, 0}, wb[2] = {0, 0};
      encode_chunk16_x2<BITS, EMIT>(A, B, CA, CB, cur.d, ahead.d[0], lds, wa, wb, last_qd_a, last_qd_b, sqa, sqb);
      if (EMIT && writer) store_chunk_codes_x2<BITS>(body + (uint64_t)k * kOutStride, wa, wb);
    };
    for (uint32_t k = 0; k < full; k += 3) {
      one(k, b0, b1, b2);
      if (k + 1 < full) one(k + 1, b1, b2, b0);
      if (k + 2 < full) one(k + 2, b2, b0, b1);
    }
  }
  const uint32_t done = full * kChunk;
  if (EMIT) { /* tail units: samples past n are zero padding - reference :592-593 */
    uint8_t *up = body + (uint64_t)(done / US) * (UB * 2);
    for (uint32_t i = done; i < coded; i += US, up += UB * 2) {
      uint32_t acc_a = 0, acc_b = 0;
#pragma unroll
      for (int k = 0; k < US; k++) {
        const bool real = i + k < coded;
        acc_a = (acc_a << BITS) | encode_step<BITS>(A, real ? src.at(0, first + kTaps + i + k) : 0, lds, last_qd_a);
        acc_b = (acc_b << BITS) | encode_step<BITS>(B, real ? src.at(1, first + kTaps + i + k) : 0, lds, last_qd_b);
      }
      if (writer) {
#pragma unroll
        for (int k = 0; k < UB; k++) {
          up[k] = (uint8_t)(acc_a >> (8 * (UB - 1 - k)));
          up[UB + k] = (uint8_t)(acc_b >> (8 * (UB - 1 - k)));
        }
      }
    }
  } else { /* an RMSE pass stops at the last real sample (reference :457) */
    for (uint32_t i = done; i < coded; i++) {
      int32_t qa, qb;
      encode_step<BITS>(A, src.at(0, first + kTaps + i), lds, qa);
      encode_step<BITS>(B, src.at(1, first + kTaps + i), lds, qb);
      sqa += wrapped_square(qa);
      sqb += wrapped_square(qb);
    }
  }
}

/* RMSE pass of both channels - reference src/aad_encoder.c:431-467 */
template <int BITS>
__device__ __forceinline__ void rmse_pass_x2(QuadLane &A, QuadLane &B, const StereoSource &src, uint64_t first, uint32_t n, uint32_t tap,
                                             const char *lds, double &ra, double &rb)
{
  ra = rb = 0.0;
  if (n < (uint32_t)kTaps) return;
  seed_history_x2(A, B, src, first, n, tap);
  int32_t qa = 0, qb = 0;
  int64_t sa, sb;
  run_block_x2<BITS, false>(A, B, src, first, n, false, nullptr, lds, qa, qb, sa, sb);
  ra = sqrt((double)sa / (double)n);
  rb = sqrt((double)sb / (double)n);
}

/* trial search of both channels - the schedule of search_best_lane (reference src/aad_encoder.c:470-562);
 * the two channels decide independently */
template <int BITS>
__device__ __forceinline__ void search_best_x2(QuadLane &A, QuadLane &B, const StereoSource &src, uint64_t first, uint32_t n, uint32_t spb,
                                               uint32_t trials, uint32_t tap, const char *lds)
{
  const bool have_prev = first >= spb;
  QuadLane best_a = A, best_b = B, run_a = A, run_b = B;
  double best_ra = 0.0, best_rb = 0.0;
  const uint32_t per_trial = have_prev ? 2u : 1u;
  const uint32_t passes = have_prev ? 1u + 2u * trials : trials;
  for (uint32_t p = 0; p < passes; p++) {
    const bool is_probe = have_prev && p == 0;
    const bool on_prev = have_prev && p != 0 && ((p - 1u) % per_trial) == 0;
    QuadLane from_a = is_probe ? A : run_a, from_b = is_probe ? B : run_b;
    const QuadLane before_a = from_a, before_b = from_b;
    double ra, rb;
    rmse_pass_x2<BITS>(from_a, from_b, src, on_prev ? first - spb : first, on_prev ? spb : n, tap, lds, ra, rb);
    if (!is_probe) {
      run_a = from_a;
      run_b = from_b;
    }
    if (is_probe || (!have_prev && p == 0)) {
      best_ra = ra;
      best_rb = rb;
    } else if (!on_prev) {
      if (best_ra > ra) {
        best_ra = ra;
        best_a = before_a;
      }
      if (best_rb > rb) {
        best_rb = rb;
        best_b = before_b;
      }
    }
  }
  A = best_a;
  B = best_b;
}

/* the same with the probe strand on lanes of its own - the schedule of search_best_lane_dual.
 * role 0 / role 1 quads of a stream are neighbouring recurrence slots (adjacent lanes, tap-major). */
template <int BITS>
__device__ __forceinline__ void search_best_dual_x2(QuadLane &A, QuadLane &B, const StereoSource &src, uint64_t first, uint32_t n, uint32_t spb,
                                                    uint32_t trials, uint32_t tap, uint32_t role, const char *lds)
{
  const bool have_prev = first >= spb;
  const uint32_t chain_passes = trials * (have_prev ? 2u : 1u);
  const int chain_lane = (int)((threadIdx.x & 63u) - role); /* role 0's lane of the same tap */
  QuadLane best_a = A, best_b = B, run_a = A, run_b = B;
  double best_ra = 0.0, best_rb = 0.0;
  for (uint32_t p = 0; p < chain_passes; p++) {
    const bool chain_on_prev = have_prev && (p & 1u) == 0;
    const bool on_prev = chain_on_prev && !(p == 0 && role != 0);
    const QuadLane before_a = run_a, before_b = run_b;
    double ra, rb;
    rmse_pass_x2<BITS>(run_a, run_b, src, on_prev ? first - spb : first, on_prev ? spb : n, tap, lds, ra, rb);
    if (p == 0) {
      /* the probe's figures go to both roles; role 1 takes over role 0's chain state and result */
      const double probe_a = __shfl(ra, chain_lane + 1, 64), probe_b = __shfl(rb, chain_lane + 1, 64);
      best_ra = role == 0 ? probe_a : ra;
      best_rb = role == 0 ? probe_b : rb;
      ra = __shfl(ra, chain_lane, 64);
      rb = __shfl(rb, chain_lane, 64);
      run_a.w = __shfl(run_a.w, chain_lane, 64);
      run_a.h = __shfl(run_a.h, chain_lane, 64);
      run_a.idxb = __shfl(run_a.idxb, chain_lane, 64);
      run_b.w = __shfl(run_b.w, chain_lane, 64);
      run_b.h = __shfl(run_b.h, chain_lane, 64);
      run_b.idxb = __shfl(run_b.idxb, chain_lane, 64);
    }
    if (!chain_on_prev) {
      if (best_ra > ra) {
        best_ra = ra;
        best_a = before_a; /* in pass 0 both roles started from the carried state, so `before` is the chain's as well */
      }
      if (best_rb > rb) {
        best_rb = rb;
        best_b = before_b;
      }
    }
  }
  A = best_a;
  B = best_b;
}

/*
 * Stream-parallel stereo encode, one quad per stream (two with DUAL: role 0 / role 1 of the trial
 * search).  Reference src/aad_encoder.c:814-891 with EncodeBlock :565-727 and the optional trial
 * search inlined; byte-identical to encode_streams_kernel<BITS, 2, false, ...>.
 */
template <int BITS, bool TRIALS, bool DUAL>
__global__ void __launch_bounds__(256) encode_stereo_x2_kernel(EncodeArgs a)
{
  static_assert(!DUAL || TRIALS, "the dual mapping is the trial search with the probe on lanes of its own");
  __shared__ __attribute__((aligned(16))) char lds[kLdsBytesQuad];
  stage_tables<BITS, true, 1>(lds);

  const uint64_t thread = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t slot = (thread >> 6) * 16u + quad_slot<kX2TM>();
  const uint32_t role = DUAL ? (uint32_t)(slot & 1u) : 0u;
  const uint64_t stream = DUAL ? slot >> 1 : slot;
  const uint32_t tap = quad_tap<kX2TM>();
  const bool writer = tap == 0 && role == 0;
  if (stream >= a.num_streams) return; /* whole quads / role pairs leave together */
  const uint32_t s = (uint32_t)stream;
  const StreamDesc sd = a.uni.enabled ? uniform_stream(a.uni, s) : a.streams[s];
  const StereoSource src = {a.pcm + sd.pcm_offset};
  const SampleSource<false> src0 = {src.x, 2, 0, sd.num_samples}, src1 = {src.x, 2, 1, sd.num_samples};
  uint8_t *out = a.data + sd.data_offset;
  const uint32_t total = sd.num_samples, spb = a.samples_per_block;

  Lane F0 = {0, 0, 0, 0, 0, 0, 0, 0, kIdxBias}, F1 = F0;
  int32_t last_qd_a = 0, last_qd_b = 0;
  if (a.state) {
    const LaneStateRecord r0 = a.state[(uint64_t)s * 2], r1 = a.state[(uint64_t)s * 2 + 1];
    F0 = {r0.weight[0], r0.weight[1], r0.weight[2], r0.weight[3], r0.history[0], r0.history[1], r0.history[2], r0.history[3],
          min(max(r0.stepsize_index, 0), (int32_t)AAD_STEP_INDEX_MAX) + kIdxBias};
    F1 = {r1.weight[0], r1.weight[1], r1.weight[2], r1.weight[3], r1.history[0], r1.history[1], r1.history[2], r1.history[3],
          min(max(r1.stepsize_index, 0), (int32_t)AAD_STEP_INDEX_MAX) + kIdxBias};
    last_qd_a = r0.quantize_error;
    last_qd_b = r1.quantize_error;
  }

  if (writer) { /* file header - reference src/aad_encoder.c:190-214 */
    for (int i = 0; i < kFileHeaderBytes; i++) out[i] = a.header_template[i];
    out[14] = (uint8_t)(total >> 24);
    out[15] = (uint8_t)(total >> 16);
    out[16] = (uint8_t)(total >> 8);
    out[17] = (uint8_t)total;
  }

  uint64_t block_off = kFileHeaderBytes;
  for (uint64_t first = 0; first < total; first += spb, block_off += a.block_size) {
    const uint32_t n = total - first < spb ? (uint32_t)(total - first) : spb;
    QuadLane A, B;
    if constexpr (TRIALS) { /* reference src/aad_encoder.c:863-871 */
      A = to_quad(F0, tap);
      B = to_quad(F1, tap);
      if constexpr (DUAL) search_best_dual_x2<BITS>(A, B, src, first, n, spb, a.trials, tap, role, lds);
      else search_best_x2<BITS>(A, B, src, first, n, spb, a.trials, tap, lds);
      F0 = from_quad<kX2TM>(A);
      F1 = from_quad<kX2TM>(B);
    }
    seed_history(F0, src0, first, n);
    seed_history(F1, src1, first, n);
    write_block_header(F0, out + block_off, writer);
    write_block_header(F1, out + block_off + kBlockHeaderBytesPerCh, writer);
    A = to_quad(F0, tap);
    B = to_quad(F1, tap);
    int64_t sqa, sqb;
    /* dual: role 1 runs the encode pass as well (it holds the same state; only role 0 stores) */
    run_block_x2<BITS, true>(A, B, src, first, n, writer, out + block_off + 2u * kBlockHeaderBytesPerCh, lds, last_qd_a, last_qd_b, sqa, sqb);
    F0 = from_quad<kX2TM>(A);
    F1 = from_quad<kX2TM>(B);
  }

  if (a.state_out && writer) {
    LaneStateRecord r;
    r.weight[0] = F0.w0; r.weight[1] = F0.w1; r.weight[2] = F0.w2; r.weight[3] = F0.w3;
    r.history[0] = F0.h0; r.history[1] = F0.h1; r.history[2] = F0.h2; r.history[3] = F0.h3;
    r.stepsize_index = F0.idxb - kIdxBias;
    r.quantize_error = last_qd_a;
    a.state_out[(uint64_t)s * 2] = r;
    r.weight[0] = F1.w0; r.weight[1] = F1.w1; r.weight[2] = F1.w2; r.weight[3] = F1.w3;
    r.history[0] = F1.h0; r.history[1] = F1.h1; r.history[2] = F1.h2; r.history[3] = F1.h3;
    r.stepsize_index = F1.idxb - kIdxBias;
    r.quantize_error = last_qd_b;
    a.state_out[(uint64_t)s * 2 + 1] = r;
  }
}

} /* namespace aad */

#endif /* AAD_ENCODE_X2_HIP_H */
