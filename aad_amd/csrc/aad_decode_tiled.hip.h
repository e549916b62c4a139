/*
 * aad_decode_tiled.hip.h - the dense decoder with SECTOR-TILED I/O (reference src/aad_decoder.c:321-475,
 * the same arithmetic as decode_blocks_kernel: lane = (block, channel), decode_chunk16 per 16 samples).
 *
 * Why: in the dense mapping the lanes of a wave are a whole block apart in memory, so when every lane moves
 * its own bytes a wave-level load or store touches 64 different 128-byte lines for 16 bytes each.  Measured
 * on a chip-filling batch (profiles/r03a_saturated_geometries_pmc.txt, tools/microbench/ubench_fetch.hip):
 *   - a read that misses the L2 always fills a whole 128-byte line; a lane that walks its line in 64-byte
 *     bursts a chunk-group's worth of time apart fetches it twice - three times when the bursts straddle
 *     (code bytes start 49 / 67 bytes into an image): the mono decoder moved 3.5x its code bytes;
 *   - the same pattern moves 3.7-4.5 TB/s where rows of adjacent lanes move 4.9-5.3 TB/s;
 *   - the kernel's VALU sat idle 26-41 % of the time waiting for those lines.
 * Here the memory side of a wave works on ROWS instead: a row is one block (mono: one lane, stereo: the
 * lane pair), and a wave-level access covers whole aligned granules of a row with ADJACENT lanes -
 * mono: 4 lanes x 16 B = one 64-byte sector of each of 16 rows per instruction, stereo: 8 lanes x 16 B =
 * one 128-byte line of each of 8 rows.  The granules pass through a two-granule ring per row in LDS:
 *   in   code bytes: cooperative global_load -> ds_write_b128 into the row's ring; the row's lane(s) read
 *        their chunk's bytes back as aligned dwords (a granule is sector-aligned in MEMORY, the chunk's bytes
 *        sit at the byte phase the .aad layout gives them - one v_perm_b32 selector per lane absorbs it);
 *        the block header comes through the ring as well (18 byte loads per lane less);
 *   out  PCM: the lanes write their packed chunks (ds_write_b128) at the row's phase; after every second
 *        chunk the oldest granule of every row is complete and leaves with one cooperative store
 *        (ds_read_b128 -> global_store_dwordx4), 16-byte pieces outside a row's range masked.
 * Every granule is fetched once and written once, as whole sectors / lines, two periods ahead of its use
 * (memory latency never meets the recurrence), with a quarter of the L2 requests.  LDS: 272 (mono) or 528
 * (stereo) bytes per row = 17 KB per wave, so a CU holds 8 waves (two per SIMD) - enough, because what
 * is left to wait for is LDS, and the chunk body was software-pipelined for a lone wave.
 *
 * Scope: mono / stereo, every block's PCM 16-byte aligned (host-checked: uniform batches are); 3-bit codes (a chunk is 6 / 12
 * bytes, which divide no granule) only in batches whose code bytes share their phase - see "3-bit rows" below.
 * Blocks shorter than a chunk, truncated images and the last samples of a block take the same per-lane
 * tail as decode_blocks_kernel; both kernels produce identical bytes (tests run them side by side).
 */
#ifndef AAD_DECODE_TILED_HIP_H
#define AAD_DECODE_TILED_HIP_H

#include "aad_decode.hip.h"

namespace aad {

template <int BITS, int CHF>
struct DecodeTile {
  static_assert((BITS == 4 || BITS == 3 || BITS == 2) && (CHF == 1 || CHF == 2), "mono / stereo");
  static constexpr bool k3 = BITS == 3;                        /* 3-bit codes: see "3-bit rows" below */
  static constexpr int kWaves = 4;                             /* waves per workgroup */
  static constexpr int kRows = 64 / CHF;                       /* rows (blocks) per wave */
  static constexpr int kG = CHF == 1 ? 64 : 128;               /* granule bytes, both directions */
  static constexpr int kGLog2 = CHF == 1 ? 6 : 7;
  static constexpr int kRing = 2 * kG;                         /* two granules per row and direction */
  static constexpr int kMirror = 16;                           /* the input ring's first 16 bytes again behind its end: a chunk's aligned dwords never wrap */
  /* the output ring carries 16 bytes of padding: rows a multiple of 128 bytes apart put the same piece of every row on
   * the same banks (an eight-way conflict on every ds_write_b128 of packed PCM) */
  static constexpr int kInPitch = kRing + kMirror, kOutPitch = kRing + 16;
  static constexpr int kCb = Pack<BITS>::kChunkBytes * CHF;    /* code bytes of a row per chunk: 8 / 16 (4-bit), 6 / 12 (3-bit), 4 / 8 (2-bit) */
  static constexpr int kInPeriod = k3 ? 0 : kG / kCb;          /* chunks per input granule: 8 (4-bit), 16 (2-bit); 3-bit: 10 2/3 */
  static constexpr int kPcmBytes = 2 * kChunk * CHF;           /* PCM bytes of a row per chunk: 32 / 64 */
  static_assert(kG / kPcmBytes == 2, "an output granule is two chunks");
  static constexpr int kLead = k3 ? 0 : 12;                    /* decoded samples of the lead chunk (+ 4 verbatim); 3-bit: none */
  static constexpr int kLeadBytes = kLead * BITS / 8 * CHF;
  static constexpr int kShift = k3 ? 2 * kTaps * CHF : 0;      /* 3-bit: chunk j's PCM starts kShift + j * kPcmBytes into the block */
  static constexpr int kRaw = (kCb + 3) / 4 + 1;               /* aligned dwords that hold a chunk's code bytes at any byte phase */
  /* mono 2-bit blocks are 8 mod 16 bytes of PCM long (4028 samples at 1024 bytes): every second block of a stream starts 8 bytes
   * off the piece grid.  Such a row opens with a SHORT lead chunk - 4 verbatim + 8 decoded samples = 24 bytes, two code bytes -
   * which puts its chunks on the grid (and on the same positions as its neighbours': theta + 24 and theta + 32 range over the same
   * values), and the first piece of its first granule is half a piece (its upper 8 bytes). */
  static constexpr bool kOdd8 = BITS == 2 && CHF == 1;
  static constexpr int kShortLead = 8, kShortLeadBytes = kShortLead * BITS / 8;
  static constexpr int kLanesPerRow = kG / 16;                 /* lanes that cover one granule: 4 / 8 */
  static constexpr int kRowsPerInst = 64 / kLanesPerRow;       /* 16 / 8 */
  static constexpr int kInst = kRows / kRowsPerInst;           /* wave-level accesses per granule of every row: 4 */
  static constexpr int kMetaBytes = 32;                        /* per row, see RowMeta */
  /* the row records are read once, before the first PCM byte is written: they lie in the output rings' space */
  static constexpr int kInOff = 0, kOutOff = kRows * kInPitch, kMetaOff = kOutOff;
  static constexpr int kWaveBytes = kOutOff + kRows * kOutPitch;
  static_assert(kMetaBytes <= kOutPitch, "a row's record fits its own output ring");
};

/* what the lanes that move a row's granules need to know about it (written by the row's channel-0 lane) */
struct RowMeta {
  uint64_t in_base;  /* address of input granule 0: the aligned granule that holds the first code byte behind the lead chunk */
  int32_t g_min, g_max; /* granules of this row that hold at least one byte of the block (loads are clamped to them) */
  uint64_t out_base; /* address of output granule 0: the aligned granule that holds the block's first PCM byte */
  uint32_t lo, hi;   /* bytes of the row's PCM that come out of the ring, relative to out_base (multiples of 16) */
};
static_assert(sizeof(RowMeta) == 32, "RowMeta is one 32-byte LDS record");

constexpr int kLdsTileOff = (kLdsBytesDenseDec + 15) & ~15;

/*
 * 3-bit rows.  A 3-bit unit is eight samples in three bytes, so (a) no lead chunk of twelve samples exists: the four verbatim
 * samples go into the output ring by themselves and chunk j's PCM sits 8 (mono) / 16 (stereo) bytes further on - mono chunks
 * are written as 8 + 16 + 8 bytes (three aligned LDS writes), a row's last piece may be half a piece (stored as 8 bytes), and an
 * output granule is complete one chunk later than on 4- / 2-bit rows; (b) a chunk is 6 / 12 code bytes, which divide no granule:
 * rows of different byte phase would be up to three granules apart at the same chunk, and a third ring slot does not fit eight
 * waves per CU (measured with three slots and six waves: VALU-active 63 % instead of 80 %, no faster than the per-lane kernel).
 * So 3-bit rows take this kernel only when EVERY row's code bytes sit at the same offset U inside their granule (host-checked,
 * DecodeArgs::code_phase_uniform: image pitch and block size multiples of 128 bytes - uniform batches are built that way); the
 * next granule then goes in when chunk j's aligned dwords first reach it: need(j) = (((U + kCb * j) & ~3) + 4 * kRaw - 1) >> kGLog2;
 * (c) a mono chunk's byte phase alternates (6 = 2 mod 4).
 */
/*
 * A chunk in two strands.  What a sample needs from the tables - its code's record and the step size at its step index -
 * depends on the CODES only (idx' = clamp(idx + delta[code]) never sees a sample), so all sixteen lookups of a chunk are
 * made before its arithmetic starts: walk_chunk (strand 1) fills 48 registers, run_chunk (strand 2) is then pure VALU work
 * with no LDS wait inside.  The kernel runs strand 1 of chunk j + 1 in front of strand 2 of chunk j, so the lookups'
 * latency - long with eight waves sharing the CU's LDS - is covered by a whole chunk of arithmetic.  decode_chunk16
 * interleaves the two sample by sample (the lookup of sample j + 1 behind ~22 instructions of sample j): right for a
 * lone wave with 64 registers, not for two waves per SIMD with 256.
 */
struct ChunkWalk {
  uint32_t step[kChunk]; /* step << 2 of every sample */
  u32x2 rec[kChunk];     /* {bias << 29 | delta & 0xFFFF, sm21 << 27} of every sample's code */
};

template <int BITS>
__device__ __forceinline__ void walk_chunk(int32_t &idxb, const uint32_t *w, const char *lds, ChunkWalk &W)
{
  constexpr int cpw = Pack<BITS>::kCodesPerWord;
  const uint32_t copy = (threadIdx.x & 3u) << 2;
  static_for<0, kChunk>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    constexpr int sh = 3, pos = Pack<BITS>::pos(j % cpw);
    const uint32_t word = w[j / cpw];
    const uint32_t addr = (pos >= sh ? word >> (pos >= sh ? pos - sh : 0) : word << (pos >= sh ? 0 : sh - pos)) & (((1u << BITS) - 1u) << sh);
    W.rec[j] = *reinterpret_cast<const u32x2 *>(lds + kLdsDenseCode8Off + addr);
  });
  static_for<0, kChunk>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    W.step[j] = *reinterpret_cast<const uint32_t *>(lds + kLdsDenseStepOff + (((uint32_t)idxb & 0xFF0u) | copy));
    idxb = clamp_idx(idxb + (int32_t)(int16_t)W.rec[j].x);
  });
}

/* p: the prediction for the chunk's first sample on entry, for the next chunk's on exit */
template <int BITS, typename Finish>
__device__ __forceinline__ void run_chunk(Lane &L, const ChunkWalk &W, int32_t &p, int32_t *y, Finish finish)
{
  static_for<0, kChunk>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const int32_t qd = dense_dequantise(W.step[j], u32x3{W.rec[j].x, 0u, W.rec[j].y});
    const int32_t yy = clip16(qd + p);
    p = lms_shift_predict(L, qd, yy);
    y[j] = finish(yy);
  });
}

template <int BITS, int CHF, bool MS>
__global__ void __launch_bounds__((64 * DecodeTile<BITS, CHF>::kWaves), 2) decode_tiled_kernel(DecodeArgs a)
{
  using T = DecodeTile<BITS, CHF>;
  constexpr bool k3 = T::k3;
  constexpr uint32_t ch = CHF;
  constexpr uint32_t kRingMask = T::kRing - 1;
  __shared__ __attribute__((aligned(16))) char lds[kLdsTileOff + T::kWaves * T::kWaveBytes];
  stage_tables<BITS, false>(lds);
  stage_dense_decode_tables<BITS>(lds);

  const uint32_t wl = threadIdx.x & 63u; /* lane within the wave */
  char *const tile = lds + kLdsTileOff + (threadIdx.x >> 6) * T::kWaveBytes;
  const uint64_t lane = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; /* index of the (block, channel) recurrence */
  const uint64_t wave_first = lane - wl;
  if (wave_first >= a.total_blocks * ch) return; /* the whole wave: nothing to decode, nothing to move */
  const bool active = lane < a.total_blocks * ch;
  const uint64_t g = active ? lane / ch : 0;
  const uint32_t c = active ? (uint32_t)(lane % ch) : 0;
  const uint32_t my_row = wl / CHF;

  /* ---- this lane's block, exactly as decode_blocks_kernel finds it */
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
  uint32_t n = 0;
  if (active && first < sd.num_samples) {
    const uint64_t left = sd.num_samples - first;
    n = left < a.samples_per_block ? (uint32_t)left : a.samples_per_block;
  }
  const uint64_t block_off = a.header_bytes + b * a.block_size;
  const uint64_t avail64 = sd.data_size > block_off ? sd.data_size - block_off : 0;
  const uint32_t avail = avail64 > 0x7FFFFFFFu ? 0x7FFFFFFFu : (uint32_t)avail64;
  const uint8_t *src = a.data + sd.data_offset + block_off;
  int16_t *dst = a.pcm + sd.pcm_offset + first * ch + c;
  constexpr uint32_t body = (uint32_t)kBlockHeaderBytesPerCh * ch;
  if (avail < body) n = 0;

  constexpr int US = Pack<BITS>::kUnitSamples, UB = Pack<BITS>::kUnitBytes;
  const uint32_t coded = n > (uint32_t)kTaps ? n - kTaps : 0;
  /* lead: the row's first samples go through the output ring - 4- / 2-bit: the lead chunk (4 verbatim + 12 decoded); 3-bit: the 4 verbatim */
  const uintptr_t pcm0 = reinterpret_cast<uintptr_t>(a.pcm + sd.pcm_offset + first * ch);
  const uintptr_t out_base = pcm0 & ~(uintptr_t)(T::kG - 1);
  /* a multiple of 16 (mono 2-bit: of 8, see kOdd8): the host only launches this kernel on such layouts */
  const uint32_t theta = (uint32_t)(pcm0 - out_base);
  const bool odd8 = T::kOdd8 && (theta & 8u) != 0;
  const uint32_t lead_dec = odd8 ? (uint32_t)T::kShortLead : (uint32_t)T::kLead;          /* decoded samples of the lead chunk */
  const uint32_t lead_bytes = odd8 ? (uint32_t)T::kShortLeadBytes : (uint32_t)T::kLeadBytes; /* ... and its code bytes */
  const uint32_t tbase = odd8 ? theta - 8u : theta; /* chunk j's PCM sits at tbase + (j + 1) * kPcmBytes (3-bit: theta + kShift + j * ...) */
  const bool lead = k3 ? (n >= (uint32_t)kTaps) : (coded >= lead_dec && avail >= body + lead_bytes);
  uint32_t full = 0; /* whole 16-sample chunks behind the lead chunk whose code bytes are all there */
  if (lead) {
    full = (coded - lead_dec) / kChunk;
    const uint32_t fit = (avail - body - lead_bytes) / T::kCb;
    full = full < fit ? full : fit;
  }

  /* ---- the row's place in memory, published for the lanes that move its granules */
  const uintptr_t code0 = reinterpret_cast<uintptr_t>(src) + body + lead_bytes; /* first code byte behind the lead chunk */
  const uintptr_t in_base = code0 & ~(uintptr_t)(T::kG - 1);
  const int32_t hdr_pos = (int32_t)(reinterpret_cast<uintptr_t>(src) - in_base); /* block start relative to granule 0 (<= 0 ... < kG) */
  if (c == 0) {
    RowMeta m;
    m.in_base = in_base;
    const uint32_t present = avail < a.block_size ? avail : a.block_size; /* bytes of this block that exist */
    m.g_min = hdr_pos >> T::kGLog2;                                        /* floor: -1 or 0 */
    m.g_max = n ? (int32_t)((uint32_t)(hdr_pos + (int32_t)present - 1 + T::kG) >> T::kGLog2) - 1 : m.g_min;
    if (m.g_max < m.g_min) m.g_max = m.g_min;
    m.out_base = out_base;
    m.lo = theta;
    m.hi = lead ? (k3 ? theta + T::kShift + full * T::kPcmBytes : tbase + (1u + full) * T::kPcmBytes) : theta;
    *reinterpret_cast<RowMeta *>(tile + T::kMetaOff + my_row * T::kMetaBytes) = m;
  }
  /* trip count of the wave: the longest row's */
  uint32_t full_max = full;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const uint32_t other = (uint32_t)__shfl_xor((int)full_max, o, 64);
    full_max = other > full_max ? other : full_max;
  }
  full_max = (uint32_t)__builtin_amdgcn_readfirstlane((int)full_max);
  wave_lds_fence();

  /* ---- the rows this lane moves: instruction i covers rows i * kRowsPerInst + wl / kLanesPerRow, piece wl % kLanesPerRow */
  const uint32_t piece = wl % T::kLanesPerRow, sub_row = wl / T::kLanesPerRow;
  uint64_t mv_in[T::kInst], mv_out[T::kInst];
  int32_t mv_gmin[T::kInst], mv_gmax[T::kInst];
  uint32_t mv_lo[T::kInst], mv_hi[T::kInst];
#pragma unroll
  for (int i = 0; i < T::kInst; i++) {
    const RowMeta m = *reinterpret_cast<const RowMeta *>(tile + T::kMetaOff + (i * T::kRowsPerInst + sub_row) * T::kMetaBytes);
    mv_in[i] = m.in_base + 16u * piece;
    mv_out[i] = m.out_base + 16u * piece;
    mv_gmin[i] = m.g_min;
    mv_gmax[i] = m.g_max;
    mv_lo[i] = m.lo;
    mv_hi[i] = m.hi;
  }
  /* cooperative moves.  Rows of inactive lanes (the tail of the last wave) were never published: their metadata is
   * whatever the LDS held - so those lanes publish too (c == 0 above covers them: g = 0, n = 0, hi = lo). */
  auto load_granule = [&](int32_t v, u32x4 (&r)[T::kInst]) {
#pragma unroll
    for (int i = 0; i < T::kInst; i++) {
      const int32_t gv = min(max(v, mv_gmin[i]), mv_gmax[i]);
      r[i] = *reinterpret_cast<const u32x4 *>(mv_in[i] + ((int64_t)gv << T::kGLog2));
    }
  };
  auto put_granule = [&](int32_t v, const u32x4 (&r)[T::kInst]) {
    const uint32_t slot = ((uint32_t)v & 1u) * T::kG + 16u * piece;
#pragma unroll
    for (int i = 0; i < T::kInst; i++) {
      char *ring = tile + T::kInOff + (i * T::kRowsPerInst + sub_row) * T::kInPitch;
      *reinterpret_cast<u32x4 *>(ring + slot) = r[i];
      if (slot == 0) *reinterpret_cast<u32x4 *>(ring + T::kRing) = r[i]; /* the mirror of the ring's first 16 bytes */
    }
  };
  auto read_granule = [&](uint32_t t, u32x4 (&r)[T::kInst]) { /* this lane's pieces of output granule t, out of the rows' rings */
    const uint32_t slot = (t & 1u) * T::kG + 16u * piece;
#pragma unroll
    for (int i = 0; i < T::kInst; i++)
      r[i] = *reinterpret_cast<const u32x4 *>(tile + T::kOutOff + (i * T::kRowsPerInst + sub_row) * T::kOutPitch + slot);
  };
  auto write_granule = [&](uint32_t t, const u32x4 (&r)[T::kInst]) { /* ... to memory, the pieces that belong to their row */
    const uint32_t at = t * T::kG + 16u * piece;
#pragma unroll
    for (int i = 0; i < T::kInst; i++) {
      if (k3 && CHF == 1) { /* a mono 3-bit row ends 8 bytes into a piece */
        if (at >= mv_lo[i] && at + 16u <= mv_hi[i]) store_through(mv_out[i] + (uint64_t)t * T::kG, r[i]);
        else if (at >= mv_lo[i] && at < mv_hi[i]) *reinterpret_cast<u32x2 *>(mv_out[i] + (uint64_t)t * T::kG) = u32x2{r[i].x, r[i].y};
      } else if (T::kOdd8) { /* a mono 2-bit row may START 8 bytes into a piece (its end is on the grid) */
        if (at >= mv_lo[i] && at < mv_hi[i]) store_through(mv_out[i] + (uint64_t)t * T::kG, r[i]);
        else if (at + 8u == mv_lo[i] && mv_lo[i] < mv_hi[i]) *reinterpret_cast<u32x2 *>(mv_out[i] + (uint64_t)t * T::kG + 8u) = u32x2{r[i].z, r[i].w}; /* (a row with nothing in the ring has hi == lo) */
      } else if (at >= mv_lo[i] && at < mv_hi[i]) {
        store_through(mv_out[i] + (uint64_t)t * T::kG, r[i]);
      }
    }
  };

  /* ---- prologue: granules -1 and 0 (block header, lead chunk, first codes) and granule 1 */
  u32x4 ga[T::kInst], gb[T::kInst], gn[T::kInst];
  load_granule(-1, ga);
  load_granule(0, gb);
  load_granule(1, gn);
  put_granule(-1, ga);
  put_granule(0, gb);
  wave_lds_fence();

  const char *const in_ring = tile + T::kInOff + my_row * T::kInPitch;
  char *const out_ring = tile + T::kOutOff + my_row * T::kOutPitch;
  auto ring_byte = [&](int32_t pos) -> uint32_t { return (uint32_t)(uint8_t)in_ring[(uint32_t)pos & kRingMask]; };
  auto ring_be16 = [&](int32_t pos) -> uint32_t { return (ring_byte(pos) << 8) | ring_byte(pos + 1); };

  Lane H = {0, 0, 0, 0, 0, 0, 0, 0, kIdxBias};
  if (n) { /* block header - reference src/aad_decoder.c:364-380 */
    const int32_t hp = hdr_pos + (int32_t)(c * kBlockHeaderBytesPerCh);
    const uint32_t v = ring_be16(hp);
    H.idxb = min((int32_t)(v >> 4), (int32_t)kHeaderIdxMax) + kIdxBias;
    const uint32_t shift = v & 0xFu;
    H.w0 = (int32_t)((uint32_t)(int32_t)(int16_t)ring_be16(hp + 2) << shift);
    H.h0 = (int16_t)ring_be16(hp + 4);
    H.w1 = (int32_t)((uint32_t)(int32_t)(int16_t)ring_be16(hp + 6) << shift);
    H.h1 = (int16_t)ring_be16(hp + 8);
    H.w2 = (int32_t)((uint32_t)(int32_t)(int16_t)ring_be16(hp + 10) << shift);
    H.h2 = (int16_t)ring_be16(hp + 12);
    H.w3 = (int32_t)((uint32_t)(int32_t)(int16_t)ring_be16(hp + 14) << shift);
    H.h3 = (int16_t)ring_be16(hp + 16);
  }
  auto finish = [&](int32_t y) -> int32_t {
    if (MS) {
      const int32_t other = (int32_t)pair_swap<false>((uint32_t)y, c);
      return c == 0 ? clip16(y + other) : clip16(other - y);
    }
    return y;
  };
  const int32_t y0 = finish(H.h3), y1 = finish(H.h2), y2 = finish(H.h1), y3 = finish(H.h0);
  Lane L = H;

  /* The code bytes of the chunk whose first byte is `pos` bytes from granule 0 (negative: the lead chunk), read as
   * aligned dwords (`raw`: the ring's mirror lets them run past its end) one chunk AHEAD of their use, and the
   * chunk's big-endian code words from them.  The byte phase pos & 3 is absorbed by the selector of a v_perm_b32
   * (mono) or by v_alignbyte_b32 (stereo: the pair's L/R-interleaved bytes are realigned, then this channel's
   * picked); it is the same for every chunk of a block (a chunk is 4 .. 16 bytes). */
  constexpr int kRaw = T::kRaw;
  auto fetch_raw = [&](int32_t pos, uint32_t (&raw)[kRaw]) {
    const char *at = in_ring + ((uint32_t)pos & (kRingMask & ~3u));
#pragma unroll
    for (int k = 0; k < kRaw; k++) raw[k] = *reinterpret_cast<const uint32_t *>(at + 4 * k);
  };
  auto unpack = [&](const uint32_t (&raw)[kRaw], uint32_t ph, uint32_t *w) {
    if constexpr (k3) {
      /* the chunk's bytes from phase 0: E0 .. E7 (mono: six of them) or E0 .. E11 (the pair's L R L R units) */
      uint32_t e[kRaw - 1];
#pragma unroll
      for (int k = 0; k < kRaw - 1; k++) e[k] = __builtin_amdgcn_alignbyte(raw[k + 1], raw[k], ph);
      if constexpr (CHF == 1) {
        w[0] = perm(0u, e[0], 0x0c000102);    /* E0 E1 E2 */
        w[1] = perm(e[1], e[0], 0x0c030405);  /* E3 E4 E5 */
      } else {
        w[0] = perm(e[1], e[0], c ? 0x0c030405u : 0x0c000102u); /* E3 E4 E5 : E0 E1 E2 */
        w[1] = perm(e[2], e[1], c ? 0x0c050607u : 0x0c020304u); /* E9 E10 E11 : E6 E7 E8 */
      }
    } else if constexpr (CHF == 1) {
      const uint32_t sel = 0x00010203u + 0x01010101u * ph; /* bytes ph .. ph + 3 of a dword pair, most significant first */
      w[0] = perm(raw[1], raw[0], sel);
      if constexpr (BITS == 4) w[1] = perm(raw[2], raw[1], sel);
    } else {
      uint32_t e[kRaw - 1];
#pragma unroll
      for (int k = 0; k < kRaw - 1; k++) e[k] = __builtin_amdgcn_alignbyte(raw[k + 1], raw[k], ph);
      const uint32_t pick = 0x00020406u + c * 0x01010101u;
      w[0] = perm(e[1], e[0], pick);
      if constexpr (BITS == 4) w[1] = perm(e[3], e[2], pick);
    }
  };
  auto put_pcm = [&](uint32_t x, const ChunkPcm &o) { /* x: the chunk's first PCM byte, relative to the block's */
    if constexpr (k3 && CHF == 1) { /* x = 8 mod 16: 8 + 16 + 8 bytes, each piece aligned to its size (a piece never wraps) */
      *reinterpret_cast<u32x2 *>(out_ring + ((theta + x) & kRingMask)) = u32x2{o.v[0].x, o.v[0].y};
      *reinterpret_cast<u32x4 *>(out_ring + ((theta + x + 8u) & kRingMask)) = u32x4{o.v[0].z, o.v[0].w, o.v[1].x, o.v[1].y};
      *reinterpret_cast<u32x2 *>(out_ring + ((theta + x + 24u) & kRingMask)) = u32x2{o.v[1].z, o.v[1].w};
    } else {
#pragma unroll
      for (int h = 0; h < 2; h++)
        *reinterpret_cast<u32x4 *>(out_ring + ((tbase + x + (CHF == 1 ? 16u * h : 32u * h + 16u * c)) & kRingMask)) = o.v[h];
    }
  };

  const int32_t s_pos = (int32_t)(code0 - in_base); /* first code byte behind the lead chunk, 0 <= s_pos < kG */
  uint32_t raw[kRaw];
  ChunkPcm pending; /* the packed PCM of the chunk just decoded: written to the ring after the next chunk's code bytes are asked for */
  pending.v[0] = pending.v[1] = u32x4{0, 0, 0, 0};
  if constexpr (k3) {
    if (lead) { /* the verbatim frames: the rows' first kShift bytes of the output ring (theta + kShift <= kG: no wrap) */
      int16_t *v = reinterpret_cast<int16_t *>(out_ring + theta) + c;
      v[0] = (int16_t)y0;
      v[ch] = (int16_t)y1;
      v[2 * ch] = (int16_t)y2;
      v[3 * ch] = (int16_t)y3;
    }
  } else if (lead) {
    uint32_t w[2] = {0, 0};
    fetch_raw(s_pos - (int32_t)lead_bytes, raw);
    unpack(raw, (uint32_t)(s_pos - (int32_t)lead_bytes) & 3u, w);
    int32_t y[kChunk];
    y[0] = y0;
    y[1] = y1;
    y[2] = y2;
    y[3] = y3;
    if (T::kOdd8 && odd8) { /* the short lead chunk: 4 + 8 samples = 24 bytes, at theta (8 mod 16): 8 + 16 bytes */
#pragma unroll
      for (int j = kTaps + T::kShortLead; j < kChunk; j++) y[j] = 0;
      decode_chunk16<BITS, (T::kOdd8 ? T::kShortLead : kChunk), true>(L, w, lds, y + kTaps, finish);
      const ChunkPcm o = pack_chunk_pcm<CHF, false>(y, c);
      *reinterpret_cast<u32x2 *>(out_ring + (theta & kRingMask)) = u32x2{o.v[0].x, o.v[0].y};
      *reinterpret_cast<u32x4 *>(out_ring + ((theta + 8u) & kRingMask)) = u32x4{o.v[0].z, o.v[0].w, o.v[1].x, o.v[1].y};
    } else {
      decode_chunk16<BITS, (k3 ? kChunk : T::kLead), true>(L, w, lds, y + kTaps, finish);
      pending = pack_chunk_pcm<CHF, false>(y, c);
    }
  }
  if (!lead) {
    if (n > 0) dst[0] = (int16_t)y0;
    if (n > 1) dst[ch] = (int16_t)y1;
    if (n > 2) dst[2 * ch] = (int16_t)y2;
    if (n > 3) dst[3 * ch] = (int16_t)y3;
  }
  wave_lds_fence();
  put_granule(1, gn); /* over granule -1: header and lead chunk have been read */
  load_granule(2, gn);
  wave_lds_fence();
  /* byte phase of a chunk's code bytes: constant within a block, except mono 3-bit (6-byte chunks): ph, ph ^ 2, ph, ... */
  const uint32_t ph = (uint32_t)s_pos & 3u;
  const uint32_t ph_odd = (k3 && CHF == 1) ? ph ^ 2u : ph;
  int32_t pos = s_pos;
  auto next_pos = [&]() { pos += T::kCb; };
  /* 3-bit: the offset of the code bytes inside their granule, the same for every row of the batch (see "3-bit rows") */
  const uint32_t uphase = k3 ? (uint32_t)__builtin_amdgcn_readfirstlane(s_pos) : 0u;
  fetch_raw(pos, raw);
  if (!k3 && lead && !odd8) put_pcm(0, pending);
  ChunkWalk wa, wb; /* the tables of the chunk in arithmetic and of the one behind it */
  int32_t idx_run = L.idxb; /* the step index runs a chunk ahead of the samples */
  if (0 < full) {
    uint32_t w[2] = {0, 0};
    unpack(raw, ph, w);
    walk_chunk<BITS>(idx_run, w, lds, wa);
  }
  next_pos();
  fetch_raw(pos, raw); /* chunk 1's code bytes */
  int32_t p = predict(L);
  wave_lds_fence();

  /* ---- steady state, one iteration per chunk j of every row (cur: chunk j's tables, `raw`: chunk j + 1's code bytes):
   *   when chunk j + 2 opens an input period: granule period + 2 replaces granule period (every row has read its bytes
   *     of chunks <= j + 1 - they are in registers);
   *   odd j: output granule (j - 1) / 2 is complete in every row since chunk j - 1 and chunk j is about to write over it in
   *     some: its pieces are read now and stored behind the chunk's arithmetic;
   *   strand 1 of chunk j + 1, strand 2 of chunk j, ask for chunk j + 2's code bytes, write chunk j's PCM. */
  /* 3-bit: x = kShift + j * kPcmBytes, granule t is complete in every row once chunk 2t + 1 is written and chunk 2t + 2 writes
   *   over it in some: it leaves at even j >= 2; the input granule `period + 2` goes in as soon as the chunk whose bytes this
   *   iteration fetches (j + 2) needs it (see "3-bit rows"). */
  uint32_t x = k3 ? T::kShift : T::kPcmBytes;
  int32_t period = 0;
  auto one = [&](uint32_t j, const ChunkWalk &cur, ChunkWalk &next, uint32_t ph_next) {
    const bool refill = k3 ? (int32_t)((((uphase + T::kCb * (j + 2)) & ~3u) + 4 * kRaw - 1) >> T::kGLog2) > period + 1
                           : (j + 2) % (k3 ? 1 : T::kInPeriod) == 0;
    if (refill) {
      put_granule(period + 2, gn);
      period++;
      load_granule(period + 2, gn);
    }
    u32x4 leaving[T::kInst];
    const bool store_now = k3 ? ((j & 1u) == 0 && j >= 2) : (j & 1u) != 0;
    const uint32_t t_now = k3 ? (j - 2) >> 1 : (j - 1) >> 1;
    if (store_now) read_granule(t_now, leaving);
    if (j + 1 < full) {
      uint32_t w[2] = {0, 0};
      unpack(raw, ph_next, w);
      walk_chunk<BITS>(idx_run, w, lds, next);
    }
    if (j < full) {
      int32_t y[kChunk];
      run_chunk<BITS>(L, cur, p, y, finish);
      pending = pack_chunk_pcm<CHF, false>(y, c);
    }
    if (store_now) write_granule(t_now, leaving);
    wave_lds_fence();
    next_pos();
    fetch_raw(pos, raw);
    if (j < full) put_pcm(x, pending);
    x += T::kPcmBytes;
    wave_lds_fence();
  };
  for (uint32_t j = 0; j < full_max; j += 2) {
    one(j, wa, wb, ph_odd); /* unpacks chunk j + 1 */
    if (j + 1 < full_max) one(j + 1, wb, wa, ph);
  }
  L.idxb = idx_run; /* the tail continues where the last walked chunk ended */
  /* what the rows still hold: the (two) granules the loop has not stored */
  const uint32_t t_first = k3 ? (full_max ? (full_max - 1) >> 1 : 0u) : full_max >> 1;
  for (uint32_t t = t_first; t <= t_first + 1; t++) {
    u32x4 leaving[T::kInst];
    read_granule(t, leaving);
    write_granule(t, leaving);
  }

  /* ---- remaining units of the block, per lane: byte loads, bytes past the stream read as zero (decode_blocks_kernel) */
  {
    const uint32_t done = lead ? lead_dec + full * kChunk : 0u;
    const uint32_t unit_stride = UB * ch;
    const uint32_t base = body + c * UB;
    for (uint32_t i = done; i < coded; i += US) {
      const uint32_t o = base + (i / US) * unit_stride;
      uint32_t acc = 0;
#pragma unroll
      for (int k = 0; k < UB; k++) acc = (acc << 8) | (o + k < avail ? (uint32_t)src[o + k] : 0u);
      acc <<= 32 - 8 * UB;
#pragma unroll
      for (int k = 0; k < US; k++) {
        const int32_t y = finish(decode_step<BITS>(L, acc >> (32 - BITS), lds));
        acc <<= BITS;
        if (i + k < coded) dst[(uint64_t)(kTaps + i + k) * ch] = (int16_t)y;
      }
    }
  }
}

} /* namespace aad */

#endif /* AAD_DECODE_TILED_HIP_H */
