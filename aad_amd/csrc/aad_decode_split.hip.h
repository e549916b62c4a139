/*
 * aad_decode_split.hip.h - two-kernel decoder for lane-starved batches (the quad mapping).
 *
 * The decoder's per-sample work has two strands (reference src/aad_decoder.c:269-318):
 *
 *   (1) code -> step index -> step size -> dequantised difference qd   (:280-296, tables)
 *   (2) qd -> sample = clip(qd + prediction) -> LMS update -> next prediction   (:298-315)
 *
 * Strand (1) never looks at a decoded sample: the index walk  idx' = clamp(idx + delta[code])
 * depends on the code stream alone, and a clamped add is a function x -> min(max(x + a, lo), hi)
 * whose compositions stay in that family.  So (1) is a PARALLEL SCAN over the samples of a block,
 * not a recurrence.  The fused kernel (decode_blocks_kernel) nevertheless runs both strands in
 * every lane's serial loop - ~23 instructions per sample for a lone wave at ~4.5-5.3 cycles each
 * - while a lane-starved batch leaves most of the chip idle.  Here:
 *
 *   decode_residuals_kernel   one WAVE per (block, channel); lane i owns samples [16i, 16i+16):
 *                             composes its chunk's clamp function, a 6-step wave scan yields
 *                             every chunk's starting index, each lane then walks its 16 samples
 *                             and writes qd (int32) to a scratch buffer.  Thousands of
 *                             independent waves, a few microseconds.
 *   decode_predict_kernel     the true recurrence (2) alone on the quad mapping (four lanes per
 *                             recurrence, lane t = tap t): ~12.5 instructions per sample, no
 *                             LDS, qd prefetched a chunk ahead with wide loads.
 *
 * Extra HBM/L2 traffic: 4 bytes written and 4 x 4 read per sample (the four lanes of a quad
 * read the same words) - irrelevant where this path is used (a batch that cannot fill the chip
 * anyway); saturating batches keep the fused dense kernel, whose total instruction count is lower.
 */
#ifndef AAD_DECODE_SPLIT_HIP_H
#define AAD_DECODE_SPLIT_HIP_H

#include "aad_decode.hip.h"

namespace aad {

struct SplitDecodeArgs {
  DecodeArgs d;
  int32_t *residual;        /* [total_blocks * channels][residual_stride] */
  uint32_t residual_stride; /* int32 per recurrence, a multiple of kChunk */
  uint32_t reserved;
};

/* where block g of the batch lives (the same arithmetic as decode_blocks_kernel) */
struct BlockRef {
  const uint8_t *src; /* first byte of the block */
  uint64_t pcm_first; /* index of the block's first frame's channel-0 sample in the PCM buffer */
  uint32_t n;         /* samples per channel in this block (0: nothing to do) */
  uint32_t avail;     /* bytes of the stream present from the start of the block */
};

__device__ __forceinline__ BlockRef locate_block(const DecodeArgs &a, uint64_t g, uint32_t ch)
{
  uint32_t s;
  StreamDesc sd;
  uint64_t b;
  if (a.uni.enabled) {
    s = (uint32_t)g / a.uni.blocks_per_stream;
    b = (uint32_t)g - s * a.uni.blocks_per_stream;
    sd = uniform_stream(a.uni, s);
  } else {
    s = find_stream(a.block_prefix, a.num_streams, g);
    sd = a.streams[s];
    b = g - a.block_prefix[s];
  }
  const uint64_t first = b * a.samples_per_block;
  BlockRef r;
  r.n = 0;
  if (first < sd.num_samples) {
    const uint64_t left = sd.num_samples - first;
    r.n = left < a.samples_per_block ? (uint32_t)left : a.samples_per_block;
  }
  const uint64_t block_off = a.header_bytes + b * a.block_size;
  const uint64_t avail64 = sd.data_size > block_off ? sd.data_size - block_off : 0;
  r.avail = avail64 > 0x7FFFFFFFu ? 0x7FFFFFFFu : (uint32_t)avail64;
  r.src = a.data + sd.data_offset + block_off;
  r.pcm_first = sd.pcm_offset + first * ch;
  if (r.avail < (uint32_t)kBlockHeaderBytesPerCh * ch) r.n = 0; /* DecodeBlock: INSUFFICIENT_DATA (reported by the host) */
  return r;
}

/* x -> min(max(x + a, lo), hi) on biased step indices */
struct ClampAdd {
  int32_t a, lo, hi;
};
__device__ __forceinline__ int32_t apply(const ClampAdd &f, int32_t x) { return min(max(x + f.a, f.lo), f.hi); }
/* first `f`, then `g` */
__device__ __forceinline__ ClampAdd then(const ClampAdd &f, const ClampAdd &g)
{
  ClampAdd r;
  r.a = f.a + g.a;
  r.lo = min(max(f.lo + g.a, g.lo), g.hi);
  r.hi = min(max(f.hi + g.a, g.lo), g.hi);
  return r;
}

/*
 * Strand (1) for one recurrence, executed by one whole wave: lane i owns coded samples
 * [base + 16 i, base + 16 i + 16), base advancing by 1024 for long blocks.
 * Any channel count, any block size; bytes past the end of the stream read as zero, as in the
 * fused kernel.
 */
template <int BITS>
__device__ __forceinline__ void residuals_for_recurrence(const SplitDecodeArgs &a, uint64_t rec, uint32_t lane,
                                                         const uint32_t *s_step, const int32_t *s_delta, int32_t *out)
{
  constexpr int US = Pack<BITS>::kUnitSamples, UB = Pack<BITS>::kUnitBytes, UNITS = kChunk / US;
  constexpr uint32_t kMagMask = Pack<BITS>::kMagMax, kSign = Pack<BITS>::kSign;
  const uint32_t ch = a.d.channels;
  const uint64_t g = rec / ch;
  const uint32_t c = (uint32_t)(rec % ch);
  const BlockRef blk = locate_block(a.d, g, ch);
  const uint32_t coded = blk.n > (uint32_t)kTaps ? blk.n - kTaps : 0;
  if (coded == 0) return;

  /* step index the block header carries - reference src/aad_decoder.c:364-366 */
  int32_t carry = min((int32_t)(load_be16(blk.src + c * kBlockHeaderBytesPerCh) >> 4), (int32_t)kHeaderIdxMax) + kIdxBias;
  const uint32_t body = (uint32_t)kBlockHeaderBytesPerCh * ch + c * UB;
  const uint32_t unit_stride = UB * ch;

  for (uint32_t base = 0; base < coded; base += 64u * kChunk) {
    const uint32_t k0 = base + lane * kChunk;
    const uint32_t cnt = k0 < coded ? min(coded - k0, (uint32_t)kChunk) : 0u;
    /* the lane's codes */
    uint32_t code[kChunk];
#pragma unroll
    for (int u = 0; u < UNITS; u++) {
      const uint32_t o = body + (k0 / US + u) * unit_stride;
      uint32_t acc = 0;
#pragma unroll
      for (int q = 0; q < UB; q++) acc = (acc << 8) | ((u * US < (int)cnt && o + q < blk.avail) ? (uint32_t)blk.src[o + q] : 0u);
#pragma unroll
      for (int t = 0; t < US; t++) code[u * US + t] = (acc >> (BITS * (US - 1 - t))) & ((1u << BITS) - 1u);
    }
    /* this chunk's index walk as one clamp function */
    /* the identity: NO clamp of its own - a block header may carry an index above kIdxMax (up to kHeaderIdxMax, the
     * reference takes the field as it is), and x -> clamp(x, lo, hi) in front of the first delta would cut it down before
     * the delta is added (found by tests/test_gpu_bitstream_fuzz.py: header index 4087, first delta -14) */
    ClampAdd f = {0, -(1 << 20), 1 << 20};
#pragma unroll
    for (int j = 0; j < kChunk; j++) {
      if (j < (int)cnt) {
        const ClampAdd one = {s_delta[code[j] & kMagMask], kIdxMin, kIdxMax};
        f = then(f, one);
      }
    }
    /* inclusive scan over the wave's chunks */
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      ClampAdd prev;
      prev.a = __shfl_up(f.a, off, 64);
      prev.lo = __shfl_up(f.lo, off, 64);
      prev.hi = __shfl_up(f.hi, off, 64);
      if ((int)lane >= off) f = then(prev, f);
    }
    ClampAdd before;
    before.a = __shfl_up(f.a, 1, 64);
    before.lo = __shfl_up(f.lo, 1, 64);
    before.hi = __shfl_up(f.hi, 1, 64);
    int32_t idxb = lane == 0 ? carry : apply(before, carry);
    ClampAdd all;
    all.a = __shfl(f.a, 63, 64);
    all.lo = __shfl(f.lo, 63, 64);
    all.hi = __shfl(f.hi, 63, 64);
    carry = apply(all, carry);

    /* walk the chunk: step size, dequantised difference (reference :283-296), next index */
    int32_t qd[kChunk];
#pragma unroll
    for (int j = 0; j < kChunk; j++) {
      const uint32_t mag = code[j] & kMagMask;
      const uint32_t step = s_step[((uint32_t)idxb >> 4) & 0xFFu];
      const int32_t q = (int32_t)((step * ((mag << 1) | 1u)) >> (BITS - 1));
      qd[j] = (code[j] & kSign) ? -q : q;
      idxb = clamp_idx(idxb + s_delta[mag]);
    }
    if (cnt == (uint32_t)kChunk) {
#pragma unroll
      for (int v = 0; v < kChunk / 4; v++) {
        u32x4 w;
        w.x = (uint32_t)qd[4 * v];
        w.y = (uint32_t)qd[4 * v + 1];
        w.z = (uint32_t)qd[4 * v + 2];
        w.w = (uint32_t)qd[4 * v + 3];
        *reinterpret_cast<u32x4 *>(out + k0 + 4 * v) = w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < kChunk; j++)
        if (j < (int)cnt) out[k0 + j] = qd[j];
    }
  }
}

/* sixteen residuals of one recurrence, loaded a chunk ahead (cf. ChunkCodes) */
struct ChunkResiduals {
  u32x4 r[4];
  __device__ __forceinline__ void load(const int32_t *p)
  {
#pragma unroll
    for (int v = 0; v < 4; v++) r[v] = *reinterpret_cast<const u32x4 *>(p + 4 * v);
  }
  __device__ __forceinline__ void touch()
  {
#pragma unroll
    for (int v = 0; v < 4; v++) asm volatile("" : "+v"(r[v]) :: "memory");
  }
  __device__ __forceinline__ int32_t get(int j) const
  {
    const u32x4 v = r[j >> 2];
    return (int32_t)((j & 3) == 0 ? v.x : ((j & 3) == 1 ? v.y : ((j & 3) == 2 ? v.z : v.w)));
  }
};

/* state carried from sample to sample besides the quad lane itself */
struct PredictCarry {
  int32_t p;     /* prediction for the next sample */
  int32_t lmsd;  /* the next sample's weight increment, (qd * h + 2^14) >> 18, already computed */
  uint32_t up;   /* the next-older tap's history sample (DPP), already fetched */
};

__device__ __forceinline__ void predict_prime(const QuadLane &L, PredictCarry &C, int32_t qd0)
{
  C.p = predict(L);
  C.lmsd = mad_i24(qd0, L.h, 16384) >> 18;
  C.up = quad_dpp<0x90>((uint32_t)L.h);
}

/*
 * Sixteen steps of strand (2).  Per sample the dependent chain is
 *   add (qd + p) -> clip -> select into the history -> product -> 2 DPP butterfly adds -> shift
 * and everything else - the weight increment and the history shift of the NEXT sample, which
 * need only this sample's reconstructed value - is issued in the wait states the DPP adds need.
 * `ahead` is the first residual of the following chunk (anything when there is none).
 */
template <typename Finish>
__device__ __forceinline__ void predict_chunk16_quad(QuadLane &L, PredictCarry &C, const ChunkResiduals &cur, int32_t ahead,
                                                     int32_t *y, Finish finish)
{
  int32_t p = C.p, lmsd = C.lmsd;
  uint32_t up = C.up;
  static_for<0, kChunk>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const int32_t yy = clip16(cur.get(j) + p);
    L.w += lmsd;
    L.h = L.tap0 ? yy : (int32_t)up;
    uint32_t s = (uint32_t)L.h * (uint32_t)L.w + L.round;
    /* gap 1: the next sample's LMS product */
    const int32_t m = mad_i24(j + 1 < kChunk ? cur.get(j + 1 < kChunk ? j + 1 : j) : ahead, L.h, 16384);
    s += quad_dpp<0xB1>(s);
    /* gap 2: its shift, and the history sample the next shift moves up */
    lmsd = m >> 18;
    up = quad_dpp<0x90>((uint32_t)L.h);
    s += quad_dpp<0x4E>(s);
    p = (int32_t)s >> 15;
    y[j] = finish(yy);
    __builtin_amdgcn_sched_barrier(0);
  });
  C.p = p;
  C.lmsd = lmsd;
  C.up = up;
}

/*
 * Strand (2): the quad mapping of decode_blocks_kernel minus everything strand (1) already did.
 * CHF = 1 or 2 (the quad mapping exists for mono / stereo only).
 */
template <int CHF, bool MS>
__device__ __forceinline__ void predict_for_quad(const SplitDecodeArgs &a, uint64_t thread, const int32_t *res, uint32_t *stage)
{
  constexpr uint32_t ch = CHF;
  const uint64_t rec = thread >> 2;
  const uint32_t tap = threadIdx.x & 3u;
  const bool writer = tap == 0;
  const bool active = rec < a.d.total_blocks * ch;
  const uint64_t g = active ? rec / ch : 0;
  const uint32_t c = active ? (uint32_t)(rec % ch) : 0;
  BlockRef blk = locate_block(a.d, g, ch);
  if (!active) blk.n = 0;
  const uint32_t n = blk.n;
  int16_t *dst = a.d.pcm + blk.pcm_first + c;

  Lane H = {0, 0, 0, 0, 0, 0, 0, 0, kIdxBias};
  if (n) { /* block header - reference src/aad_decoder.c:364-380 (the step index went to strand (1)) */
    const uint8_t *hp = blk.src + c * kBlockHeaderBytesPerCh;
    const uint32_t shift = load_be16(hp) & 0xFu;
    H.w0 = (int32_t)((uint32_t)(int32_t)(int16_t)load_be16(hp + 2) << shift);
    H.h0 = (int16_t)load_be16(hp + 4);
    H.w1 = (int32_t)((uint32_t)(int32_t)(int16_t)load_be16(hp + 6) << shift);
    H.h1 = (int16_t)load_be16(hp + 8);
    H.w2 = (int32_t)((uint32_t)(int32_t)(int16_t)load_be16(hp + 10) << shift);
    H.h2 = (int16_t)load_be16(hp + 12);
    H.w3 = (int32_t)((uint32_t)(int32_t)(int16_t)load_be16(hp + 14) << shift);
    H.h3 = (int16_t)load_be16(hp + 16);
  }
  auto finish = [&](int32_t yv) -> int32_t {
    if (MS) {
      const int32_t other = (int32_t)pair_swap<true>((uint32_t)yv, c);
      return c == 0 ? clip16(yv + other) : clip16(other - yv);
    }
    return yv;
  };
  { /* the first four samples are stored verbatim in the header - reference :386-391 */
    const int32_t y0 = finish(H.h3), y1 = finish(H.h2), y2 = finish(H.h1), y3 = finish(H.h0);
    if (writer) {
      if (n > 0) dst[0] = (int16_t)y0;
      if (n > 1) dst[ch] = (int16_t)y1;
      if (n > 2) dst[2 * ch] = (int16_t)y2;
      if (n > 3) dst[3 * ch] = (int16_t)y3;
    }
  }
  QuadLane L = to_quad<false>(H, tap);
  AAD_PHASE_MARK(blockIdx.x == 0 && threadIdx.x == 0);
  const uint32_t coded = n > (uint32_t)kTaps ? n - kTaps : 0;
  const uint32_t full = coded / kChunk;
  int16_t *op = a.d.pcm + blk.pcm_first + (uint64_t)kTaps * ch; /* frame of the chunk's first sample, channel 0 */

  PredictCarry C = {0, 0, 0};
  {
    /* Three register sets rotate (the loop is unrolled by three): chunk k is consumed from one,
     * chunk k+1 - whose first residual the last sample of chunk k looks ahead to - sits in the
     * second, chunk k+2 is in flight into the third.  No copies, and the compiler's own vmcnt
     * waits land where a set is first read, two chunks after its loads were issued. */
    const int32_t *rp = res;
    ChunkResiduals b0, b1, b2;
#pragma unroll
    for (int v = 0; v < 4; v++) b0.r[v] = b1.r[v] = b2.r[v] = u32x4{0, 0, 0, 0};
    if (full) {
      b0.load(rp);
      if (full > 1) rp += kChunk;
      b1.load(rp);
      predict_prime(L, C, b0.get(0));
    }
    /* Stereo output.  All four lanes of a quad hold the same sixteen samples and the two quads of
     * a block sit side by side, so the L/R interleave is shared out over the block's eight lanes
     * instead of being done by the two tap-0 lanes with DPP swaps: every lane packs its chunk
     * into eight dwords and drops them into a 1 KB LDS staging area (two buffers, this wave's
     * sixteen rows), and one chunk later - no LDS round trip to wait for - lane (channel, tap)
     * picks dword 4 * channel + tap of the block's L row and of its R row and stores those two
     * frames: eight lanes x 8 bytes = the chunk's 64 contiguous bytes.  17 instruction slots per
     * chunk instead of 36. */
    constexpr bool COOP = CHF == 2;
    const uint32_t quad = (threadIdx.x & 63u) >> 2;
    const uint32_t part = threadIdx.x & 7u; /* = 4 * channel + tap */
    uint32_t *my_row = stage + quad * 8u;
    const uint32_t *l_row = stage + (quad & ~1u) * 8u + part, *r_row = l_row + 8;
    auto stage_chunk = [&](uint32_t k, const int32_t *y) {
      u32x4 lo, hi;
      lo.x = perm((uint32_t)y[1], (uint32_t)y[0], 0x05040100);
      lo.y = perm((uint32_t)y[3], (uint32_t)y[2], 0x05040100);
      lo.z = perm((uint32_t)y[5], (uint32_t)y[4], 0x05040100);
      lo.w = perm((uint32_t)y[7], (uint32_t)y[6], 0x05040100);
      hi.x = perm((uint32_t)y[9], (uint32_t)y[8], 0x05040100);
      hi.y = perm((uint32_t)y[11], (uint32_t)y[10], 0x05040100);
      hi.z = perm((uint32_t)y[13], (uint32_t)y[12], 0x05040100);
      hi.w = perm((uint32_t)y[15], (uint32_t)y[14], 0x05040100);
      uint32_t *dst = my_row + (k & 1u) * 128u;
      *reinterpret_cast<u32x4 *>(dst) = lo;
      *reinterpret_cast<u32x4 *>(dst + 4) = hi;
    };
    auto flush_chunk = [&](uint32_t k, uint32_t l, uint32_t r) { /* frames 2 * part, 2 * part + 1 of chunk k */
      u32x2 v;
      v.x = perm(r, l, 0x05040100);
      v.y = perm(r, l, 0x07060302);
      reinterpret_cast<U32x2 *>(op + (uint64_t)k * kChunk * ch + 4u * part)->v = v;
    };
    auto one = [&](uint32_t k, const ChunkResiduals &cur, const ChunkResiduals &ahead, ChunkResiduals &incoming) {
      if (k + 2 < full) rp += kChunk; /* prefetch chunk k+2 (clamped to the last full one) */
      incoming.load(rp);
      uint32_t l = 0, r = 0;
      if (COOP && k) { /* the previous chunk's two staged dwords: requested now, used after this chunk's arithmetic */
        l = l_row[((k - 1) & 1u) * 128u];
        r = r_row[((k - 1) & 1u) * 128u];
      }
      int32_t y[kChunk];
      predict_chunk16_quad(L, C, cur, ahead.get(0), y, finish);
      if constexpr (COOP) {
        stage_chunk(k, y);
        if (k) flush_chunk(k - 1, l, r);
      } else {
        if (writer) store_chunk_pcm<CHF, true>(op + (uint64_t)k * kChunk * ch, y, c, ch);
      }
    };
    AAD_PHASE_MARK(blockIdx.x == 0 && threadIdx.x == 0);
    for (uint32_t k = 0; k < full; k += 3) {
      one(k, b0, b1, b2);
      if (k + 1 < full) one(k + 1, b1, b2, b0);
      if (k + 2 < full) one(k + 2, b2, b0, b1);
    }
    AAD_PHASE_MARK(blockIdx.x == 0 && threadIdx.x == 0);
    if constexpr (COOP) {
      if (full) flush_chunk(full - 1, l_row[((full - 1) & 1u) * 128u], r_row[((full - 1) & 1u) * 128u]);
    }
  }
  /* What is left of the block (fewer than 16 samples), one at a time.  The residuals are fetched
   * with the same four wide loads first (the scratch rows are padded by a chunk): a load per
   * iteration would put a memory round trip on every one of these last samples. */
  int32_t p = full ? C.p : predict(L);
  const uint32_t rem = coded - full * kChunk;
  if (rem) {
    ChunkResiduals last;
    last.load(res + full * kChunk);
    static_for<0, kChunk - 1>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if ((uint32_t)j < rem) {
        const int32_t qd = last.get(j);
        const int32_t yy = clip16(qd + p);
        lms_and_shift<kShiftSelect>(L, qd, yy);
        p = predict(L);
        const int32_t yo = finish(yy);
        if (writer) dst[(uint64_t)(kTaps + full * kChunk + j) * ch] = (int16_t)yo;
      }
    });
  }
  AAD_PHASE_MARK(blockIdx.x == 0 && threadIdx.x == 0);
}

/*
 * Both strands in one launch.  A workgroup is 16 waves and owns 16 recurrences: first every wave
 * runs strand (1) for one of them (lane = 16-sample chunk), then - one barrier later - wave 0
 * alone runs strand (2) for all sixteen on the quad mapping while the other fifteen retire.  The
 * residuals cross over through LDS when a block's worth fits (LDSRES) and through the scratch
 * buffer in device memory otherwise.
 */
/* LDS-resident residuals: rows of kLdsResidualRow dwords, enough for blocks of up to
 * kLdsResidualMax coded samples per channel (every 4-bit geometry up to max_block_size 1024 and
 * most others); the row length is 20 mod 32 dwords so that the sixteen rows a wave reads at the
 * same sample offset spread over the banks.  132 KB of the CU's 160 KB: one workgroup per CU,
 * which is what this path is for (the host uses it up to one workgroup per CU). */
constexpr uint32_t kLdsResidualMax = 2048;
constexpr uint32_t kLdsResidualRow = kLdsResidualMax + kChunk + 4;

template <int BITS, int CHF, bool MS, bool LDSRES>
__global__ void __launch_bounds__(1024) decode_split_kernel(SplitDecodeArgs a)
{
  __shared__ uint32_t s_step[AAD_STEP_TABLE_LEN];
  __shared__ int32_t s_delta[8];
  __shared__ __attribute__((aligned(16))) int32_t s_res[LDSRES ? 16 * kLdsResidualRow : 4];
  __shared__ __attribute__((aligned(16))) uint32_t s_stage[2 * 16 * 8]; /* stereo output staging of the recurrence wave */
  AAD_PHASE_MARK(blockIdx.x == 0 && threadIdx.x == 0);
  for (uint32_t i = threadIdx.x; i < AAD_STEP_TABLE_LEN; i += blockDim.x) s_step[i] = c_step_table[i];
  if (threadIdx.x < 8) {
    const int16_t *dt = BITS == 4 ? c_delta4 : (BITS == 3 ? c_delta3 : c_delta2);
    s_delta[threadIdx.x] = dt[threadIdx.x & Pack<BITS>::kMagMax];
  }
  AAD_PHASE_MARK(blockIdx.x == 0 && threadIdx.x == 0);
  __syncthreads();
  AAD_PHASE_MARK(blockIdx.x == 0 && threadIdx.x == 0);
  {
    const uint32_t w = threadIdx.x >> 6;
    const uint64_t rec = (uint64_t)blockIdx.x * 16u + w;
    if (rec < a.d.total_blocks * CHF)
      residuals_for_recurrence<BITS>(a, rec, threadIdx.x & 63u, s_step, s_delta,
                                     LDSRES ? s_res + w * kLdsResidualRow : a.residual + rec * a.residual_stride);
  }
  AAD_PHASE_MARK(blockIdx.x == 0 && threadIdx.x == 0);
  /* the residuals are in LDS / in memory before wave 0 reads them (workgroup-scope release/acquire) */
  __syncthreads();
  AAD_PHASE_MARK(blockIdx.x == 0 && threadIdx.x == 0);
  if (threadIdx.x >= 64) return;
  const uint64_t thread = (uint64_t)blockIdx.x * 64u + threadIdx.x;
  const uint64_t rec = thread >> 2;
  const bool active = rec < a.d.total_blocks * CHF;
  predict_for_quad<CHF, MS>(a, thread,
                            LDSRES ? s_res + (threadIdx.x >> 2) * kLdsResidualRow
                                   : a.residual + (active ? rec : 0) * a.residual_stride,
                            s_stage);
}

} /* namespace aad */

#endif /* AAD_DECODE_SPLIT_HIP_H */
