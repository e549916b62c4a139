/*
 * aad_decode.hip.h - the decoder side of the device code: single step, the hand-pipelined
 * 16-sample chunk bodies (dense and quad mapping), code fetch, PCM stores and
 * decode_blocks_kernel, the one-lane-does-both-strands decoder (reference
 * src/aad_decoder.c:269-475).  The split decoder for small batches is aad_decode_split.hip.h;
 * shared pieces (tables, lane state, LMS, prediction, shuffles) are in aad_device.hip.h.
 */
#ifndef AAD_DECODE_HIP_H
#define AAD_DECODE_HIP_H

#include "aad_device.hip.h"

namespace aad {

/* one decoder step - reference src/aad_decoder.c:269-318; `code` in the low BITS bits.  Plain
 * form for tails; the bulk goes through decode_chunk16. */
template <int BITS, typename S>
__device__ __forceinline__ int32_t decode_step(S &L, uint32_t code, const char *lds)
{
  const uint32_t step = *reinterpret_cast<const uint32_t *>(lds + kLdsStepOff + slot_addr(L.idxb));
  const u32x2 t = code_record(lds, (code & ((1u << BITS) - 1u)) << kLdsCodeShift);
  const int32_t qd = record_dequantise<BITS>(step, t);
  const int32_t y = clip16(qd + predict(L));
  L.idxb = clamp_idx(L.idxb + record_delta(t));
  lms_and_shift(L, qd, y);
  return y;
}

/*
 * Sixteen decoder steps, software-pipelined by hand.  Codes are known a chunk ahead, so the
 * per-code record of sample j+2 is fetched during sample j; the only lookup on the recurrence
 * is the step size of the next sample, started as soon as the new index is known and hidden
 * behind this sample's reconstruction, LMS update and the next prediction (~22 instructions).
 *   A  step index of the next sample; start its step lookup and the record lookup of sample j+2
 *   B  dequantise (mad + shift), reconstruct, LMS, history shift, predict the next sample
 */
#ifndef AAD_DENSE_REC8
#define AAD_DENSE_REC8 0 /* experiment: 1 = the per-lane dense decoder takes its code records from the 8-byte table too */
#endif
/* REC8: the per-code records come from the 8-byte table (kLdsDenseCode8Off: conflict-free ds_read_b64, one v_mov_b32 for
 * the addend's upper word) instead of the 16-byte one */
template <int BITS, int N = kChunk, bool REC8 = false, typename S, typename Finish>
__device__ __forceinline__ void decode_chunk16(S &L, const uint32_t *w, const char *lds, int32_t *y, Finish finish)
{
  static_assert(N >= 2 && N <= kChunk, "the first N samples of a chunk's code words");
  constexpr int cpw = Pack<BITS>::kCodesPerWord;
  auto code_addr = [&](int j) -> uint32_t { /* (code << 4) - or << 3 - for sample j, j compile-time after unrolling */
    constexpr int sh = REC8 ? 3 : 4;
    const int pos = Pack<BITS>::pos(j % cpw);
    const uint32_t word = w[j / cpw];
    return (pos >= sh ? word >> (pos >= sh ? pos - sh : 0) : word << (sh - pos)) & (((1u << BITS) - 1u) << sh);
  };
  /* ONE lookup each for the code's record and the step, the step's address ONE instruction: measured
   * against per-code dword arrays and a dword step array (conflict-free, but three lookups and a
   * two-instruction address), same box, kernel time in us - 1000 x 16 blocks 66.4 vs 71.2, 1250 x 10
   * blocks 65.2 vs 69.8, 20 000 mono blocks 121 vs 137; at saturation, where the kernel waits for
   * memory, the same (0.60 ms) although the records cost 5 conflict cycles per lookup there */
  auto record = [&](uint32_t addr) -> u32x3 {
    if (REC8) {
      const u32x2 r = *reinterpret_cast<const u32x2 *>(lds + kLdsDenseCode8Off + addr);
      return u32x3{r.x, 0u, r.y};
    }
    return *reinterpret_cast<const u32x3 *>(lds + kLdsDenseCodeOff + addr);
  };
  const uint32_t copy = (threadIdx.x & 3u) << 2;
  auto step_at = [&](int32_t idxb) { return *reinterpret_cast<const uint32_t *>(lds + kLdsDenseStepOff + (((uint32_t)idxb & 0xFF0u) | copy)); };
  uint32_t step = step_at(L.idxb); /* step << 2 */
  /* per-code records two samples ahead: they depend on nothing but the code bits */
  u32x3 t0 = record(code_addr(0));
  u32x3 t1 = record(code_addr(1));
  int32_t p = predict(L);
  static_for<0, N>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    /* A */
    const uint32_t step_j = step;
    const u32x3 t_j = t0;
    t0 = t1;
    L.idxb = clamp_idx(L.idxb + (int32_t)(int16_t)t_j.x);
    if (j + 1 < N) step = step_at(L.idxb);
    if (j + 2 < N) t1 = record(code_addr(j + 2 < N ? j + 2 : j));
    __builtin_amdgcn_sched_barrier(0);
    /* B */
    const int32_t qd = dense_dequantise(step_j, t_j);
    const int32_t yy = clip16(qd + p);
    if (j + 1 < N) {
      p = lms_shift_predict(L, qd, yy);
      pin(p);
    } else {
      lms_and_shift(L, qd, yy);
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
  u32x2 t0, t1, t2;      /* per-code records of samples j, j+1, j+2 */
  int32_t p;             /* prediction for sample j */
  int32_t idx_next;      /* step index (biased) of the first sample after the chunk just finished */
};

template <int BITS>
__device__ __forceinline__ uint32_t chunk_code_addr(const uint32_t *w, const uint32_t *wn, int j)
{
  constexpr int cpw = Pack<BITS>::kCodesPerWord;
  constexpr int sh = kLdsCodeShift;
  const uint32_t word = j < kChunk ? w[j / cpw] : wn[(j - kChunk) / cpw];
  const int pos = Pack<BITS>::pos((j % kChunk) % cpw);
  return (pos >= sh ? word >> (pos >= sh ? pos - sh : 0) : word << (sh - pos)) & (((1u << BITS) - 1u) << sh);
}

template <int BITS>
__device__ __forceinline__ void decode_prime_quad(QuadLane &L, DecodeCarry &C, const uint32_t *w, const char *lds)
{
  auto record = [&](int j) { return code_record(lds, chunk_code_addr<BITS>(w, w, j)); };
  auto step_at = [&](int32_t idxb) { return *reinterpret_cast<const uint32_t *>(lds + kLdsWideOff + wide_addr(idxb)); };
  C.step0 = step_at(L.idxb);
  C.t0 = record(0);
  C.t1 = record(1);
  C.t2 = record(2);
  L.idxb = clamp_idx(L.idxb + record_delta(C.t0)); /* from here on L.idxb runs one sample ahead */
  C.step1 = step_at(L.idxb);
  C.p = predict(L);
  C.idx_next = L.idxb;
}

template <int BITS, typename Finish>
__device__ __forceinline__ void decode_chunk16_quad(QuadLane &L, DecodeCarry &C, const uint32_t *w, const uint32_t *wn,
                                                    const char *lds, int32_t *y, Finish finish)
{
  auto record = [&](int j) { return code_record(lds, chunk_code_addr<BITS>(w, wn, j)); };
  auto step_at = [&](int32_t idxb) { return *reinterpret_cast<const uint32_t *>(lds + kLdsWideOff + wide_addr(idxb)); };
  uint32_t step0 = C.step0, step1 = C.step1;
  u32x2 t0 = C.t0, t1 = C.t1, t2 = C.t2;
  int32_t p = C.p;
  static_for<0, kChunk>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const int32_t qd = record_dequantise<BITS>(step0, t0);
    const int32_t yy = clip16(qd + p);
    lms_and_shift<kShiftSelect>(L, qd, yy);
    y[j] = finish(yy);
    uint32_t s = (uint32_t)L.h * (uint32_t)L.w + L.round;
    pin(s);
    /* gap 1: index of sample j+2 and the record lookup of sample j+3.  The record is started
     * BEFORE the step lookup below: LDS results return in order, so the wait for a step size at
     * the top of a sample also covers the record whose delta is needed in the middle of the one
     * before - one s_waitcnt per sample, not two. */
    int32_t idx2 = clamp_idx(L.idxb + record_delta(t1));
    const u32x2 t3 = record(j + 3);
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
  uint32_t stream_stores; /* dense stereo kernel: every chunk store of every block is a whole 64-byte granule (host-checked): launch the NT instantiation */
  uint32_t pcm_aligned16; /* every stream's PCM starts on a 16-byte boundary relative to `pcm` (host-checked): what the sector-tiled kernel needs */
  uint32_t code_phase_uniform; /* every block of every stream starts at the same offset inside a granule (64 bytes mono, 128 stereo) relative to `data` (host-checked; 2: if no stream has a second block): 3-bit rows on the sector-tiled kernel */
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

/* The same for any channel count: the chunk's units of channel c sit UB * channels bytes apart, so they
 * are fetched byte by byte (8 / 6 / 4 loads per chunk) - still one chunk ahead of the arithmetic, like
 * the wide loads above.  `p` points at the first byte of the chunk's first unit of channel c. */
template <int BITS>
struct ChunkCodesAny {
  static constexpr int kUB = Pack<BITS>::kUnitBytes, kUnits = kChunk / Pack<BITS>::kUnitSamples, kBytes = kUnits * kUB;
  uint32_t b[kBytes];
  __device__ __forceinline__ void load(const uint8_t *p, uint32_t unit_stride)
  {
#pragma unroll
    for (int u = 0; u < kUnits; u++)
#pragma unroll
      for (int q = 0; q < kUB; q++) b[u * kUB + q] = p[(uint32_t)u * unit_stride + q];
  }
  __device__ __forceinline__ void touch() /* see ChunkCodes::touch */
  {
    asm volatile("" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) :: "memory");
    if (kBytes == 6) asm volatile("" : "+v"(b[kBytes - 2]), "+v"(b[kBytes - 1]) :: "memory");
    if (kBytes == 8) asm volatile("" : "+v"(b[kBytes - 4]), "+v"(b[kBytes - 3]), "+v"(b[kBytes - 2]), "+v"(b[kBytes - 1]) :: "memory");
  }
  /* big-endian code words, as ChunkCodes::unpack leaves them */
  __device__ __forceinline__ void unpack(uint32_t *w) const
  {
    if (BITS == 3) {
      w[0] = (b[0] << 16) | (b[1] << 8) | b[2];
      w[1] = (b[3] << 16) | (b[4] << 8) | b[5];
    } else {
      w[0] = (b[0] << 24) | (b[1] << 16) | (b[2] << 8) | b[3];
      if (BITS == 4) w[1] = (b[4] << 24) | (b[5] << 16) | (b[6] << 8) | b[kBytes - 1];
    }
  }
};

/* Mono, dense mapping: the code bytes of EIGHT chunks (64 / 48 / 32 bytes for 4- / 3- / 2-bit codes) with one
 * group of wide loads.  A lane that fetches its 8 (6, 4) bytes chunk by chunk comes back to the same 64-byte
 * sector eight times, microseconds apart on a full chip - with one sector per lane in flight the L2 cannot hold
 * them, and the saturated mono decoder fetched 5.6x its code bytes (tools/saturated_traffic.sh).  Eight chunks
 * at a time a sector is visited twice (the code bytes start 49 bytes into an image: a group straddles). */
template <int BITS>
struct GroupCodes {
  static constexpr int kChunks = 8, kBytes = kChunks * Pack<BITS>::kChunkBytes, kDwords = kBytes / 4, kVec = kDwords / 4;
  uint32_t d[kDwords];
  __device__ __forceinline__ void load(const uint8_t *p)
  {
#pragma unroll
    for (int v = 0; v < kVec; v++) {
      const u32x4 q = reinterpret_cast<const U32x4 *>(p + 16 * v)->v;
      d[4 * v] = q.x; d[4 * v + 1] = q.y; d[4 * v + 2] = q.z; d[4 * v + 3] = q.w;
    }
  }
  __device__ __forceinline__ void touch() /* see ChunkCodes::touch */
  {
#pragma unroll
    for (int v = 0; v < kVec; v++) asm volatile("" : "+v"(d[4 * v]), "+v"(d[4 * v + 1]), "+v"(d[4 * v + 2]), "+v"(d[4 * v + 3]) :: "memory");
  }
  /* big-endian code words of chunk i (compile-time after unrolling), as ChunkCodes<BITS, 1>::unpack leaves them */
  __device__ __forceinline__ void unpack(int i, uint32_t *w) const
  {
    if (BITS == 4) {
      w[0] = perm(0, d[2 * i], 0x00010203);
      w[1] = perm(0, d[2 * i + 1], 0x00010203);
    } else if (BITS == 2) {
      w[0] = perm(0, d[i], 0x00010203);
    } else { /* six bytes from byte 6 i: dword-aligned for even i, two bytes in for odd i */
      const int q = (6 * i) / 4;
      const uint32_t lo = d[q], hi = d[q + 1 < kDwords ? q + 1 : q];
      w[0] = perm(hi, lo, (i & 1) ? 0x0c020304u : 0x0c000102u);
      w[1] = perm(hi, lo, (i & 1) ? 0x0c050607u : 0x0c030405u);
    }
  }
};

/* Stereo, dense mapping: the same for a pair of lanes - 64 (48, 64) bytes of L/R-interleaved codes = four (four,
 * eight) chunks per group of loads, both lanes of the pair reading the same bytes and keeping their own */
template <int BITS>
struct StereoGroupCodes {
  static constexpr int kRaw = 2 * Pack<BITS>::kChunkBytes / 4;  /* dwords per chunk of the pair: 4 / 3 / 2 */
  static constexpr int kChunks = BITS == 2 ? 8 : 4;
  static constexpr int kDwords = kChunks * kRaw, kBytes = 4 * kDwords, kVec = kDwords / 4;
  uint32_t d[kDwords];
  __device__ __forceinline__ void load(const uint8_t *p)
  {
#pragma unroll
    for (int v = 0; v < kVec; v++) {
      const u32x4 q = reinterpret_cast<const U32x4 *>(p + 16 * v)->v;
      d[4 * v] = q.x; d[4 * v + 1] = q.y; d[4 * v + 2] = q.z; d[4 * v + 3] = q.w;
    }
  }
  __device__ __forceinline__ void touch()
  {
#pragma unroll
    for (int v = 0; v < kVec; v++) asm volatile("" : "+v"(d[4 * v]), "+v"(d[4 * v + 1]), "+v"(d[4 * v + 2]), "+v"(d[4 * v + 3]) :: "memory");
  }
  /* big-endian code words of channel c in chunk i (compile-time), as ChunkCodes<BITS, 2>::unpack leaves them */
  __device__ __forceinline__ void unpack(int i, uint32_t c, uint32_t *w) const
  {
    const uint32_t *r = d + i * kRaw;
    if (BITS == 4) {
      const uint32_t sel = 0x00020406u + c * 0x01010101u;
      w[0] = perm(r[1], r[0], sel);
      w[1] = perm(r[3], r[2], sel);
    } else if (BITS == 2) {
      w[0] = perm(r[1], r[0], 0x00020406u + c * 0x01010101u);
    } else {
      w[0] = perm(r[1], r[0], c ? 0x0c030405u : 0x0c000102u);
      w[1] = perm(r[2], r[1], c ? 0x0c050607u : 0x0c020304u);
    }
  }
};

/* Write 16 decoded samples of channel c (y[], int16 range) as interleaved PCM.  Mono: two 16-byte
 * stores.  Stereo: the two lanes of a pair trade half of their packed samples through DPP and
 * each writes 2 x 16 contiguous bytes of L/R frames.  A vector-memory instruction costs a lone
 * wave ~17 cycles to issue whatever its width, so few wide stores beat one short per sample. */
/* a 16-byte store at any (2-byte) alignment; nt = non-temporal (streamed out: the dense decoder's PCM
 * is written once and never read back, and at saturation the L2 is better spent on the code bytes
 * every block comes back to eight times per 128-byte line) */
typedef uint32_t u32x4_u2 __attribute__((ext_vector_type(4), aligned(2)));
template <bool NT>
__device__ __forceinline__ void store_u32x4(int16_t *p, u32x4 v)
{
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4_u2 *>(p));
  else reinterpret_cast<U32x4 *>(p)->v = v;
}

/* what one lane stores of a 16-sample chunk (mono / stereo): two 16-byte vectors */
struct ChunkPcm {
  u32x4 v[2];
};

template <int CHF, bool QUAD>
__device__ __forceinline__ ChunkPcm pack_chunk_pcm(const int32_t *y, uint32_t c)
{
  static_assert(CHF == 1 || CHF == 2, "the fast paths");
  ChunkPcm o;
  if (CHF == 1) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
      o.v[h].x = perm((uint32_t)y[8 * h + 1], (uint32_t)y[8 * h + 0], 0x05040100);
      o.v[h].y = perm((uint32_t)y[8 * h + 3], (uint32_t)y[8 * h + 2], 0x05040100);
      o.v[h].z = perm((uint32_t)y[8 * h + 5], (uint32_t)y[8 * h + 4], 0x05040100);
      o.v[h].w = perm((uint32_t)y[8 * h + 7], (uint32_t)y[8 * h + 6], 0x05040100);
    }
  } else {
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
      o.v[h].x = perm(ka, ra, sel_lo);
      o.v[h].y = perm(ka, ra, sel_hi);
      o.v[h].z = perm(kb, rb, sel_lo);
      o.v[h].w = perm(kb, rb, sel_hi);
    }
  }
  return o;
}

template <int CHF, bool NT>
__device__ __forceinline__ void put_chunk_pcm(int16_t *frame0, const ChunkPcm &o, uint32_t c)
{
#pragma unroll
  for (int h = 0; h < 2; h++) store_u32x4<NT>(frame0 + (CHF == 1 ? 8 * h : 16 * h + 8 * (int)c), o.v[h]);
}

/* NT: a compile-time choice (a run-time flag in the chunk loop cost the dense kernel its software
 * pipelining: 0.66 -> 1.04 ms at saturation); see DecodeArgs::stream_stores */
template <int CHF, bool QUAD, bool NT = false>
__device__ __forceinline__ void store_chunk_pcm(int16_t *frame0, const int32_t *y, uint32_t c, uint32_t ch)
{
  if constexpr (CHF == 1 || CHF == 2) {
    put_chunk_pcm<CHF, NT>(frame0, pack_chunk_pcm<CHF, QUAD>(y, c), c);
  } else {
#pragma unroll
    for (int j = 0; j < kChunk; j++) frame0[(uint32_t)j * ch + c] = (int16_t)y[j];
  }
}

/*
 * Block-parallel decode (reference src/aad_decoder.c:321-475, looped by :514-534).
 * CHF: 1 / 2 = specialised channel counts with wide chunk loads, 0 = any channel count (byte loads).
 */
/* NT: the dense stereo kernel's PCM stores are non-temporal - launched only where every store is a
 * whole 64-byte granule (DecodeArgs::stream_stores): a partial granule written around the L2 is a
 * read-modify-write at the memory (mono and unaligned geometries ran up to 2.6x slower with it) */
template <int BITS, int CHF, bool MS, bool QUAD, bool NT = false>
__global__ void __launch_bounds__(256) decode_blocks_kernel(DecodeArgs a)
{
  static_assert(!QUAD || CHF != 0, "the quad mapping exists for the mono / stereo fast paths");
  static_assert(!NT || (CHF == 2 && !QUAD && BITS != 3), "streamed stores: dense stereo kernel with the 16-frame lead chunk");
  __shared__ __attribute__((aligned(16))) char lds[QUAD ? kLdsBytesQuad : kLdsBytesDenseDec];
  AAD_PHASE_MARK(blockIdx.x == 0 && threadIdx.x == 0);
  stage_tables<BITS, QUAD>(lds);
  if constexpr (!QUAD) stage_dense_decode_tables<BITS>(lds);
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
    H.idxb = min((int32_t)(v >> 4), (int32_t)kHeaderIdxMax) + kIdxBias;
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

  constexpr int US = Pack<BITS>::kUnitSamples, UB = Pack<BITS>::kUnitBytes;
  const uint32_t coded = n > (uint32_t)kTaps ? n - kTaps : 0;
  uint32_t done = 0; /* coded samples finished */

  /* Dense mapping, 4- and 2-bit: the block opens with ONE chunk of the four verbatim frames plus
   * TWELVE decoded samples.  Chunks of sixteen decoded samples behind four verbatim frames put every
   * PCM store 16 bytes (stereo) off a 64-byte boundary: at saturation the half-written 64-byte
   * granules fell out of the L2 before the next chunk completed them and HBM saw 1.8x the output
   * bytes (profiles/r01_saturated_pmc_summary.txt).  This way every store of a block whose first
   * frame is 64-byte aligned is a whole granule.  (Twelve 3-bit samples are one and a half pack
   * units: 3-bit streams keep chunks of sixteen.) */
  constexpr bool kLeadChunk = CHF != 0 && !QUAD && BITS != 3;
  constexpr uint32_t kLead = 12, kLeadBytes = kLead * BITS / 8 * (CHF ? CHF : 1);
  constexpr uint32_t kLeadLoadBytes = ChunkCodes<BITS, (CHF ? CHF : 1)>::kLoadBytes;
  const bool lead = kLeadChunk && coded >= kLead && avail >= (uint32_t)kBlockHeaderBytesPerCh * ch + kLeadLoadBytes;

  /* the first four samples are stored verbatim in the header - reference :386-391 */
  const int32_t y0 = finish(H.h3), y1 = finish(H.h2), y2 = finish(H.h1), y3 = finish(H.h0);
  /* Mono 3-bit, dense mapping (no lead chunk: twelve 3-bit samples are one and a half units): the four verbatim
   * samples - 8 bytes - are not stored by themselves but CARRIED in front of the first pair of chunks, whose last
   * 8 bytes are carried in turn: every pair then goes out as the 64 bytes from 8 bytes in front of its first
   * sample, a whole sector on a block whose first frame is 64-byte aligned, instead of 56 + 8 bytes of two. */
  constexpr bool kCarryPcm = CHF == 1 && !QUAD && BITS == 3;
  u32x2 pcm_carry = {0, 0};
  bool carrying = false;
  if constexpr (kCarryPcm) {
    if (n > 3) {
      pcm_carry = u32x2{perm((uint32_t)y1, (uint32_t)y0, 0x05040100), perm((uint32_t)y3, (uint32_t)y2, 0x05040100)};
      carrying = true;
    }
  }
  if (writer && !lead && !carrying) {
    if (n > 0) dst[0] = (int16_t)y0;
    if (n > 1) dst[ch] = (int16_t)y1;
    if (n > 2) dst[2 * ch] = (int16_t)y2;
    if (n > 3) dst[3 * ch] = (int16_t)y3;
  }
  using S = std::conditional_t<QUAD, QuadLane, Lane>;
  S L;
  if constexpr (QUAD) L = to_quad<false>(H, tap); else L = H;
  AAD_PHASE_MARK(blockIdx.x == 0 && threadIdx.x == 0);

  if constexpr (CHF != 0) {
    /* full 16-sample chunks whose wide load stays inside the stream's bytes */
    using CC = ChunkCodes<BITS, (CHF ? CHF : 1)>;
    constexpr uint32_t kStride = Pack<BITS>::kChunkBytes * (CHF ? CHF : 1);
    const uint32_t body = (uint32_t)kBlockHeaderBytesPerCh * ch;
    const uint32_t lead_bytes = lead ? kLeadBytes : 0u;
    uint32_t full = (coded - (lead ? kLead : 0u)) / kChunk;
    if (avail < body + lead_bytes + CC::kLoadBytes) {
      full = 0;
    } else {
      const uint32_t fit = (avail - body - lead_bytes - CC::kLoadBytes) / kStride + 1;
      full = full < fit ? full : fit;
    }
    const uint8_t *cp = src + body;
    int16_t *op = a.pcm + sd.pcm_offset + (first + kTaps) * ch; /* frame of this chunk's first sample, channel 0 */
    CC next;
    next.r[0] = next.r[1] = next.r[2] = next.r[3] = 0;
    if (full || lead) next.load(cp);
    next.touch();
    ChunkPcm lead_pcm;
    lead_pcm.v[0] = lead_pcm.v[1] = u32x4{0, 0, 0, 0};
    bool lead_pending = false;
    if constexpr (kLeadChunk) {
      if (lead) {
        uint32_t w[2] = {0, 0};
        next.unpack(c, w);
        cp += kLeadBytes;
        if (full) next.load(cp);
        int32_t y[kChunk];
        y[0] = y0;
        y[1] = y1;
        y[2] = y2;
        y[3] = y3;
        decode_chunk16<BITS, (int)kLead, AAD_DENSE_REC8 != 0>(L, w, lds, y + kTaps, finish);
        next.touch();
        if constexpr (CHF == 1) { /* mono: 32 bytes - they wait for the next chunk's 32 (below): a whole 64-byte sector */
          lead_pcm = pack_chunk_pcm<1, false>(y, c);
          lead_pending = true;
        } else {
          if (writer) store_chunk_pcm<CHF, QUAD, NT>(op - (uint64_t)kTaps * ch, y, c, ch); /* frames 0-15 of the block */
        }
        op += (uint64_t)kLead * ch;
        done = kLead;
      }
    }
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
      /* Mono: two chunks at a time - the first one's packed PCM waits in registers (eight of them) and
       * both go out back to back, a whole 64-byte granule instead of two halves a chunk's worth of time
       * apart. */
      auto body = [&](uint32_t k) -> ChunkPcm {
        uint32_t w[2] = {0, 0};
        next.unpack(c, w);
        /* prefetch the next chunk (the last iteration re-reads its own: an unconditional load lands
         * straight in `next`'s registers, a conditional one would be copied - and waited for - at once);
         * it is consumed (touch) only after this chunk's arithmetic */
        if (k + 1 < full) cp += kStride;
        next.load(cp);
        int32_t y[kChunk];
        decode_chunk16<BITS, kChunk, AAD_DENSE_REC8 != 0>(L, w, lds, y, finish);
        next.touch();
        return pack_chunk_pcm<(CHF ? CHF : 1), false>(y, c);
      };
      /* a pair of mono chunks: two 32-byte halves back to back, or - carrying - the 64 bytes from 8 bytes in front */
      auto emit_pair = [&](const ChunkPcm &a, const ChunkPcm &b) {
        if (kCarryPcm && carrying) {
          int16_t *at = op - (uint64_t)kTaps;
          store_u32x4<false>(at, u32x4{pcm_carry.x, pcm_carry.y, a.v[0].x, a.v[0].y});
          store_u32x4<false>(at + 8, u32x4{a.v[0].z, a.v[0].w, a.v[1].x, a.v[1].y});
          store_u32x4<false>(at + 16, u32x4{a.v[1].z, a.v[1].w, b.v[0].x, b.v[0].y});
          store_u32x4<false>(at + 24, u32x4{b.v[0].z, b.v[0].w, b.v[1].x, b.v[1].y});
          pcm_carry = u32x2{b.v[1].z, b.v[1].w};
        } else {
          put_chunk_pcm<1, false>(op, a, c);
          put_chunk_pcm<1, false>(op + (uint64_t)kChunk, b, c);
        }
        op += (uint64_t)2 * kChunk;
      };
      uint32_t k = 0;
      if constexpr (CHF == 1) {
        /* the lead chunk's 32 bytes go out with chunk 0's: with the block's first frame on a 64-byte boundary
         * that is sector 0, and every pair of chunks behind it (1-2, 3-4, ...) is a whole sector as well */
        if (lead_pending) {
          int16_t *lp = op - (uint64_t)(kLead + kTaps) * ch;
          if (full) {
            const ChunkPcm a = body(0);
            put_chunk_pcm<1, NT>(lp, lead_pcm, c);
            put_chunk_pcm<1, NT>(op, a, c);
            op += (uint64_t)kChunk * ch;
            k = 1;
          } else {
            put_chunk_pcm<1, NT>(lp, lead_pcm, c);
          }
        }
      }
      if constexpr (CHF == 1) {
        /* groups of eight chunks while they last (GroupCodes): `next` is refilled for what follows */
        using GC = GroupCodes<BITS>;
        const uint32_t groups = (full - k) / GC::kChunks;
        if (groups) {
          GC cur, nxt;
          cur.load(cp); /* cp: chunk k's codes (body() leaves it there, with the same bytes in `next`) */
          for (uint32_t g = 0; g < groups; g++) {
            if (g + 1 < groups) cp += GC::kBytes; /* unconditional prefetch: the last group re-reads itself */
            nxt.load(cp);
            static_for<0, GC::kChunks / 2>([&](auto ic) {
              constexpr int i = decltype(ic)::value;
              uint32_t wa[2] = {0, 0}, wb[2] = {0, 0};
              cur.unpack(2 * i, wa);
              cur.unpack(2 * i + 1, wb);
              int32_t y[kChunk];
              decode_chunk16<BITS, kChunk, AAD_DENSE_REC8 != 0>(L, wa, lds, y, finish);
              const ChunkPcm a = pack_chunk_pcm<1, false>(y, c);
              decode_chunk16<BITS, kChunk, AAD_DENSE_REC8 != 0>(L, wb, lds, y, finish);
              const ChunkPcm b = pack_chunk_pcm<1, false>(y, c);
              emit_pair(a, b);
            });
            nxt.touch();
#pragma unroll
            for (int j = 0; j < GC::kDwords; j++) cur.d[j] = nxt.d[j];
          }
          k += groups * GC::kChunks;
          cp += GC::kBytes; /* the last group was its own prefetch */
          if (k < full) next.load(cp);
          next.touch();
        }
      }
      /* stereo chunks are whole granules already and the pairing costs a latency-bound launch 2 %
       * (1000 x 16 blocks: 71.2 -> 72.9 us) for 2.6 % at saturation: mono only */
      if constexpr (CHF == 2) {
        /* groups of chunks while they last (StereoGroupCodes); every chunk's PCM is a whole 64-byte store already */
        using SG = StereoGroupCodes<BITS>;
        const uint32_t groups = (full - k) / SG::kChunks;
        if (groups) {
          SG cur, nxt;
          cur.load(cp);
          for (uint32_t g = 0; g < groups; g++) {
            if (g + 1 < groups) cp += SG::kBytes; /* unconditional prefetch: the last group re-reads itself */
            nxt.load(cp);
            static_for<0, SG::kChunks>([&](auto ic) {
              constexpr int i = decltype(ic)::value;
              uint32_t w[2] = {0, 0};
              cur.unpack(i, c, w);
              int32_t y[kChunk];
              decode_chunk16<BITS, kChunk, AAD_DENSE_REC8 != 0>(L, w, lds, y, finish);
              put_chunk_pcm<2, NT>(op, pack_chunk_pcm<2, false>(y, c), c);
              op += (uint64_t)kChunk * ch;
            });
            nxt.touch();
#pragma unroll
            for (int j = 0; j < SG::kDwords; j++) cur.d[j] = nxt.d[j];
          }
          k += groups * SG::kChunks;
          cp += SG::kBytes; /* the last group was its own prefetch */
          if (k < full) next.load(cp);
          next.touch();
        }
      }
      if constexpr (CHF == 1) {
        for (; k + 2 <= full; k += 2) {
          const ChunkPcm a = body(k), b = body(k + 1);
          emit_pair(a, b);
        }
      }
      if constexpr (kCarryPcm) { /* what is still carried: the 8 bytes in front of the next sample */
        if (carrying) reinterpret_cast<U32x2 *>(op - (uint64_t)kTaps)->v = pcm_carry;
        carrying = false;
      }
      for (; k < full; k++) {
        const ChunkPcm a = body(k);
        put_chunk_pcm<(CHF ? CHF : 1), NT>(op, a, c);
        op += (uint64_t)kChunk * ch;
      }
    }
    done += full * kChunk;
  } else {
    /* any channel count (BASELINE config 4: eight): whole 16-sample chunks through the pipelined body,
     * their code bytes fetched one chunk ahead (byte loads, the units of a channel are UB * channels
     * apart), as many as lie completely inside the bytes that are there */
    using CA = ChunkCodesAny<BITS>;
    const uint32_t body = (uint32_t)kBlockHeaderBytesPerCh * ch;
    const uint32_t unit_stride = UB * ch, row = CA::kUnits * unit_stride; /* one chunk of every channel */
    uint32_t full = coded / kChunk;
    const uint32_t fit = avail > body ? (avail - body) / row : 0u;
    full = full < fit ? full : fit;
    const uint8_t *cp = src + body + c * UB;
    int16_t *op = a.pcm + sd.pcm_offset + (first + kTaps) * ch;
    CA next;
    for (auto &v : next.b) v = 0;
    if (full) next.load(cp, unit_stride);
    next.touch();
    for (uint32_t k = 0; k < full; k++) {
      uint32_t w[2] = {0, 0};
      next.unpack(w);
      if (k + 1 < full) cp += row; /* unconditional prefetch: the last iteration re-reads its own chunk */
      next.load(cp, unit_stride);
      int32_t y[kChunk];
      decode_chunk16<BITS, kChunk, AAD_DENSE_REC8 != 0>(L, w, lds, y, finish);
      next.touch();
      store_chunk_pcm<0, false>(op, y, c, ch);
      op += (uint64_t)kChunk * ch;
    }
    done += full * kChunk;
  }
  AAD_PHASE_MARK(blockIdx.x == 0 && threadIdx.x == 0);

  /* remaining units: byte loads, bytes past the stream read as zero */
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

} /* namespace aad */

#endif /* AAD_DECODE_HIP_H */
