/*
 * aad_hip_engine.hip - host side of the batched C-ABI declared in include/aad_hip.h:
 * contexts, plans (uploaded stream tables), kernel launches and the host-memory convenience
 * calls.  Device code lives in aad_encode.hip.h / aad_decode.hip.h (shared parts: aad_device.hip.h)
 * and, for the split decoder, in its own unit aad_decode_split.hip.  gfx950 only; no CPU code path -
 * every entry point that needs the GPU fails with AAD_APIRESULT_NG when HIP does.
 */
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "../../include/aad_hip.h"
#include "aad_compare.hip.h"
#include "aad_decode_split_launch.h"
#include "aad_launch.h"
#include "aad_decode_tiled_launch.h"
#include "aad_decode.hip.h"
#include "aad_encode.hip.h"
#include "aad_format.h"
#include "aad_hip_internal.h"

namespace aad {
thread_local LaunchSignal tl_launch_signal = {nullptr, nullptr}; /* aad_launch.h */
}

static_assert(sizeof(AADHipStreamDesc) == sizeof(aad::StreamDesc), "stream table layout");
static_assert(sizeof(AADHipLaneState) == sizeof(aad::LaneStateRecord), "lane state layout");
static_assert(sizeof(AADHipErrorStats) == sizeof(aad::ErrorStatsRecord), "error stats layout");

/* grow-only pinned-host + device buffer pair used by the host-memory calls.  Everything a run
 * needs (stream table, block prefix, carried state, payload) is laid out in ONE pinned block and
 * crosses PCIe in ONE copy each way: a copy per stream from pageable memory cost ~10 us apiece
 * (20 ms for a 1000-stream batch), and per-call hipMalloc / table upload / extra synchronisations
 * cost more than the kernels themselves (round 1: 2.9 ms per call for a 0.075 ms kernel). */
struct Staging {
  void *host = nullptr, *dev = nullptr;
  size_t cap = 0;
};

/* A few helper threads that only ever memcpy: the staging copies between the caller's pageable buffers
 * and the pinned blocks are the slowest stage of the host-memory path (one core moves ~10 GB/s, PCIe
 * ~50 GB/s), so chunks of a megabyte and more are filled and drained by up to four threads, each on a
 * contiguous range of streams.  Started at a context's first such chunk, joined in ContextDestroy; no
 * HIP call is ever made from them. */
struct StagingPool {
  std::vector<std::thread> threads;
  std::mutex lock;
  std::condition_variable work, done;
  std::function<void(unsigned)> job;
  unsigned generation = 0, pending = 0;
  bool stop = false;
};

struct AADHipContext {
  int device;
  hipStream_t stream;
  bool owns_stream;
  char last_error[256];
  /* double-buffered so that a big batch can be cut into chunks: while chunk k is on the device the
   * host fills chunk k+1's input block and drains chunk k-1's output block */
  Staging in[2], out[2];
  hipEvent_t chunk_done[2];
  hipEvent_t piece_done[3]; /* the earlier pieces of a one-tile decode's output copy */
  /* a cut batch runs its copies on streams of their own, so that tile k+1 goes up and tile k-1
   * comes down while tile k computes */
  hipStream_t up_stream, down_stream;
  hipEvent_t uploaded[2], computed[2];
  bool have_events, have_pipeline;
  /* device blocks of the host-memory reconstruction (grow-only): the wave's PCM, its output + statistics */
  void *d_rc_in, *d_rc_out;
  size_t rc_in_capacity, rc_out_capacity;
  /* device-only scratch of the reconstruction modes (the .aad images never leave HBM) */
  void *d_scratch;
  size_t scratch_capacity;
  /* scratch of the split decoder (dequantised differences), grow-only and shared by every decode
   * plan of the context: their runs are ordered by the context's one stream */
  int32_t *d_residual;
  uint64_t residual_capacity;
  /* scratch of the dual trial search (three block-sized slots per stream, aad_encode.hip.h
   * encode_block_dual), grow-only, shared by the context's encode launches like d_residual */
  uint8_t *d_trial;
  uint64_t trial_capacity;
  /* AADHip_ContextSetOption; the defaults come from the environment ONCE, at creation */
  aad::LaunchSignal signal_next; /* AADHip_ContextSignalNextRun: the events the next plan run records around its work (one-shot) */
  bool signal_refused;           /* ROC_SYSTEM_SCOPE_SIGNAL=0 at creation: an event on a dispatch packet is never seen by another queue */
  int32_t lane_mapping; /* enum AADHipLaneMapping */
  int32_t trial_lanes;  /* enum AADHipTrialLanes */
  int32_t compare_sequential; /* AAD_HIP_OPTION_COMPARE_ORDER: the -c sums always in the reference's order */
  int64_t tile_bytes;      /* 0 = the built-in tile budget of the host-memory path, else that many bytes */
  void *d_state;           /* predictor states of a group of streams between its tiles (host-memory encode) */
  size_t state_capacity;   /* in records */
  int32_t staging_threads; /* 0 = by core count, else the number of threads that copy (1 = the caller alone) */
  StagingPool *pool;    /* staging helper threads, created on demand */
};

struct AADHipEncodePlan {
  AADHipContext *ctx;
  aad::EncodeArgs args;
  aad::StreamDesc *d_streams;
};

struct AADHipDecodePlan {
  AADHipContext *ctx;
  aad::DecodeArgs args;
  aad::StreamDesc *d_streams;
  uint64_t *d_prefix;
};

struct AADHipReconstructPlan {
  AADHipContext *ctx;
  AADHipEncodePlan *encode;
  AADHipDecodePlan *decode;
  aad::CompareArgs args; /* streams -> the decode plan's table */
  uint64_t *d_segment_prefix;
  aad::ErrorPartial *d_partials;
};

namespace {

bool hip_ok(AADHipContext *ctx, hipError_t e, const char *what)
{
  if (e == hipSuccess) return true;
  if (ctx) snprintf(ctx->last_error, sizeof(ctx->last_error), "%s: %s", what, hipGetErrorString(e));
  return false;
}

struct DeviceGuard { /* select the context's device for the calling thread, restore on exit */
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(AADHipContext *ctx)
  {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = hip_ok(ctx, hipSetDevice(ctx->device), "hipSetDevice");
  }
  ~DeviceGuard()
  {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

/* arithmetic-progression stream tables get the table-free kernel path (aad::UniformLayout) */
aad::UniformLayout detect_uniform(uint32_t n, const AADHipStreamDesc *t)
{
  aad::UniformLayout u;
  memset(&u, 0, sizeof(u));
  if (n == 0) return u;
  const uint64_t ps = n > 1 ? t[1].pcm_offset - t[0].pcm_offset : 0, ds = n > 1 ? t[1].data_offset - t[0].data_offset : 0;
  for (uint32_t i = 0; i < n; i++) {
    if (t[i].num_samples != t[0].num_samples || t[i].data_size != t[0].data_size) return u;
    if (t[i].pcm_offset != t[0].pcm_offset + (uint64_t)i * ps || t[i].data_offset != t[0].data_offset + (uint64_t)i * ds) return u;
  }
  u.pcm_base = t[0].pcm_offset;
  u.pcm_stride = ps;
  u.data_base = t[0].data_offset;
  u.data_stride = ds;
  u.data_size = t[0].data_size;
  u.num_samples = t[0].num_samples;
  u.enabled = 1;
  return u;
}

bool hip_ok(AADHipContext *ctx, hipError_t e, const char *what);

bool staging_reserve(AADHipContext *ctx, Staging &st, size_t bytes)
{
  if (bytes <= st.cap) return true;
  if (st.host) (void)hipHostFree(st.host);
  if (st.dev) (void)hipFree(st.dev);
  st.host = st.dev = nullptr;
  st.cap = 0;
  const size_t want = bytes + bytes / 4 + 4096;
  if (!hip_ok(ctx, hipHostMalloc(&st.host, want, hipHostMallocDefault), "hipHostMalloc")) return false;
  if (!hip_ok(ctx, hipMalloc(&st.dev, want), "hipMalloc staging")) return false;
  st.cap = want;
  return true;
}

void staging_release(Staging &st)
{
  if (st.host) (void)hipHostFree(st.host);
  if (st.dev) (void)hipFree(st.dev);
  st = Staging();
}

template <typename T>
bool upload(AADHipContext *ctx, T **dst, const T *src, size_t count)
{
  if (!hip_ok(ctx, hipMalloc((void **)dst, sizeof(T) * (count ? count : 1)), "hipMalloc")) return false;
  if (count == 0) return true;
  /* pageable source: the copy is complete (staged) when this returns */
  return hip_ok(ctx, hipMemcpyAsync(*dst, src, sizeof(T) * count, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync") &&
         hip_ok(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
}


/* Lanes are scarce in every BASELINE config: while the batch has fewer
 * waves than the chip has SIMDs (256 CUs x 4) each wave gets a workgroup of its own so the
 * dispatcher spreads them over as many SIMDs as possible; big batches use 256-thread
 * workgroups so four waves share one LDS copy of the tables. */
unsigned pick_workgroup(uint64_t threads) { return threads <= 64ull * 1024ull ? 64u : 256u; }

/* Lane mapping by batch size.  "quad" (four lanes per recurrence, fewer instructions on the
 * recurrence's critical path) while the batch cannot fill the chip anyway, "dense" (one lane per
 * recurrence, fewest total instructions) beyond; the decoder has the split quad kernel below the
 * fused one.  The crossovers were measured per (bits, channels) geometry on one-block streams
 * (tools/mapping_crossover.py, profiles/r02_mapping_crossover.jsonl): they sit where the quad
 * mappings start to put a second wave on a SIMD (4 x 16384 lanes = one wave on each of the 1024
 * SIMDs) and depend on the geometry only for the split decoder, whose 3-bit stereo unpacking is the
 * most expensive strand-1 work.  A context option (AADHip_ContextSetOption, default from
 * AAD_HIP_MAPPING at context creation) forces one mapping; the parity tests run all of them. */
struct MappingLimits {
  uint32_t encode_quad;  /* recurrences up to which encode uses the quad mapping */
  uint32_t decode_split; /* ... decode uses the split quad decoder */
  uint32_t decode_fused; /* ... the fused quad decoder; dense beyond */
};

MappingLimits mapping_limits(uint32_t bits, uint32_t channels)
{
  /* tools/mapping_crossover.py, profiles/r02_mapping_crossover.jsonl.  Since the dense decoder's sample
   * went from 32.5 to 24 instructions it is as fast as the fused quad decoder at every batch size
   * (59-60 us on one-block stereo 4-bit streams, 250 to 48 000 recurrences; fused 61-67 us up to 16 384):
   * "auto" no longer picks the fused kernel (its range is empty), the option still forces it. */
  /* Round 4 (tools/size_sweep.py --mapping quad | dense, profiles/r04_decode_split_crossover.txt): the split decoder runs
   * 1024-thread workgroups of 16 recurrences, ONE to a CU (84-94 VGPRs x 16 waves), i.e. rounds of 4096 recurrences: its time is
   * about 0.025 + 0.015 ms x rounds on stereo 4-bit.  Up to two rounds (8192 recurrences) it beats the dense kernel in every
   * geometry (mono 4-bit 0.090 vs 0.116 ms at 8192), from the third round on (9000) it loses in every
   * geometry (0.120 vs 0.116) and its residual scratch (recurrences x block x 4 bytes, ~100 MB at 12 288 mono rows) pushes the
   * NEXT launch's input out of the caches: a mono 4-bit encode behind it took 0.15-0.20 ms instead of 0.127.  Round 2's
   * per-geometry limits (12 288; 9 216 / 8 192 for 4- / 3-bit stereo) predate the dense decoder's round-3 speed-ups. */
  MappingLimits m = {16384u, 8192u, 0u};
  m.decode_fused = m.decode_split;
  return m;
}

bool pick_quad(const AADHipContext *ctx, uint64_t recurrences, uint32_t channels, uint32_t bits)
{
  if (channels > 2) return false;
  if (ctx->lane_mapping == AAD_HIP_LANE_MAPPING_DENSE || ctx->lane_mapping == AAD_HIP_LANE_MAPPING_DENSE_TILED) return false;
  if (ctx->lane_mapping == AAD_HIP_LANE_MAPPING_QUAD || ctx->lane_mapping == AAD_HIP_LANE_MAPPING_QUAD_FUSED) return true;
  return recurrences <= mapping_limits(bits, channels).encode_quad;
}

/* lds_pad: dynamic LDS the kernel never touches - it only lowers the number of workgroups a CU holds (see dense_encode_lds_pad) */
/* the dense encoders whose output goes through the rows' byte rings (aad_encode.hip.h ByteRing): mono / stereo */
template <int BITS, bool TRIALS>
bool launch_encode_ring(const aad::EncodeArgs &a, dim3 grid, dim3 block, hipStream_t stream, unsigned lds_pad)
{
  if (a.channels == 1)
    AAD_LAUNCH((aad::encode_streams_kernel<BITS, 1, false, false, TRIALS, false, true>), grid, block, lds_pad, stream, a);
  else if (a.channels == 2 && a.mid_side)
    AAD_LAUNCH((aad::encode_streams_kernel<BITS, 2, true, false, TRIALS, false, true>), grid, block, lds_pad, stream, a);
  else if (a.channels == 2)
    AAD_LAUNCH((aad::encode_streams_kernel<BITS, 2, false, false, TRIALS, false, true>), grid, block, lds_pad, stream, a);
  else
    return false;
  return true;
}

/* Which dense encoders append to the byte ring.  Policy (same-box A/B on the saturated batches, profiles/r03_encoder_byte_ring.txt):
 * every mono encoder and the stereo 4-bit one - their writes fall from 1.3-2.1x to 1.00-1.06x of the code bytes and the kernels
 * get 3-8 % faster; stereo 3- and 2-bit - no: they are VALU-saturated (95 % active), wrote only 1.18x in total before, and the
 * ring's extra ~1 VALU instruction per sample costs them 3-7 % of their time.  AAD_HIP_ENCODE_RING (read at every launch: the
 * tests flip it) = 0: never (A/B measurements), = 2: every geometry that can.
 * Round 4: only in four-wave workgroups.  Batches of 16 385 .. 65 536 lanes run one-wave workgroups (a wave per SIMD, as many
 * CUs as possible); there the ring is 1-4 % SLOWER than the plain stores (same-box A/B, profiles/r04_encoder_ring_midsize.txt:
 * mono 40 000 streams 0.1835 vs 0.1811 ms, stereo 4-bit 28 000 streams 0.1044 vs 0.1003 ms) - a launch that leaves SIMDs idle
 * gains nothing from fewer write sectors and pays the ring's instructions on its critical path. */
bool encode_ring_wanted(uint32_t bits, uint32_t channels, unsigned workgroup)
{
  const char *e = getenv("AAD_HIP_ENCODE_RING");
  if (e != nullptr && e[0] == '0') return false;
  if (e != nullptr && e[0] == '2') return true;
  return workgroup == 256u && (channels == 1 || bits == 4);
}

/* One-wave workgroups (pick_workgroup: up to 65 536 lanes) only while ALL of them can be resident at once: a dense mono encoder
 * holds 45-52 KB of LDS per workgroup (wide table + code staging or ring rows), so a CU takes three of them and its fourth SIMD
 * stays empty - from 49 153 lanes (769 waves) on the launch ran in two rounds (mono 4-bit, 64 000 one-block streams: 0.35-0.41 ms
 * against 0.19 ms for 48 000, profiles/r04_encoder_ring_midsize.txt).  Four-wave workgroups share one table: two per CU. */
unsigned dense_encode_workgroup(uint64_t lanes, unsigned lds_one_wave)
{
  const unsigned wg = pick_workgroup(lanes);
  if (wg != 64u) return wg;
  const uint64_t waves = (lanes + 63u) / 64u;
  const uint64_t resident = 256ull * ((160u << 10) / ((lds_one_wave + 1023u) & ~1023u)); /* CUs x workgroups whose LDS fits */
  return waves > resident ? 256u : 64u;
}

template <int BITS, bool QUAD, bool TRIALS, bool DUAL>
void launch_encode_mapped(const aad::EncodeArgs &a, dim3 grid, dim3 block, hipStream_t stream, unsigned lds_pad = 0)
{
  if (a.channels == 1)
    AAD_LAUNCH((aad::encode_streams_kernel<BITS, 1, false, QUAD, TRIALS, DUAL>), grid, block, lds_pad, stream, a);
  else if (a.channels == 2 && a.mid_side)
    AAD_LAUNCH((aad::encode_streams_kernel<BITS, 2, true, QUAD, TRIALS, DUAL>), grid, block, lds_pad, stream, a);
  else if (a.channels == 2)
    AAD_LAUNCH((aad::encode_streams_kernel<BITS, 2, false, QUAD, TRIALS, DUAL>), grid, block, lds_pad, stream, a);
  else if constexpr (!QUAD)
    AAD_LAUNCH((aad::encode_streams_kernel<BITS, 0, false, false, TRIALS, false>), grid, block, lds_pad, stream, a);
}

/* Occupancy cap of the dense 4-bit encoders on chip-filling batches: unused dynamic LDS up to 80 KB per workgroup, so that a
 * CU holds two workgroups = two waves per SIMD instead of three or four.  These kernels are bound by VALU issue (85 % VALU-
 * active with two waves as with four), but every resident lane keeps one 128-byte line of PCM and one sector of codes alive
 * in the L2 between its visits: 8192 lines per CU at four waves per SIMD - 8 MiB per XCD against 4 MiB of L2 - and a line
 * was fetched 2.2 (mono) / 1.27 (stereo) times; with two waves per SIMD 1.57 / 1.01 times, at the same kernel time
 * (profiles/r03_encoder_occupancy_cap.txt).  The 3- and 2-bit encoders are VALU-saturated (96-103 % active) and lose 4-6 % of
 * their time under the same cap for a similar cut in traffic: they keep their occupancy.  AAD_HIP_ENCODE_LDS_PAD (bytes, read
 * once) overrides the policy for experiments. */
unsigned dense_encode_lds_pad(uint32_t bits, uint64_t lanes, unsigned static_lds)
{
  static const int forced = [] {
    const char *e = getenv("AAD_HIP_ENCODE_LDS_PAD");
    return e ? atoi(e) : -1;
  }();
  if (forced >= 0) return (unsigned)forced;
  constexpr unsigned kTarget = 80u << 10; /* two workgroups per CU (160 KB of LDS) */
  if (bits != 4 || lanes < 65536 || static_lds >= kTarget) return 0;
  return kTarget - static_lds;
}

/* On the quad mapping the trial search's probe strand gets lanes of its own ("dual"): one pass of
 * latency less per block with a predecessor, nothing lost otherwise (tools/trial_probe.py).
 * AAD_HIP_OPTION_TRIAL_LANES = single keeps both strands on the same lanes (the parity tests run both). */
/* Round 4: ... up to kDualMaxRecurrences.  The dual layout spends eight lanes per recurrence; from ~640 waves on (5120 recurrences)
 * its launch slows down faster than the work grows and the one-after-the-other layout - flat up to 16 384 recurrences - overtakes it
 * in every geometry (stereo 4-bit, t = 2: dual 0.146 / 0.161 / 0.231 ms at 4096 / 5120 / 6144 recurrences, single 0.188-0.190;
 * profiles/r04_trial_search_size_sweep.txt). */
constexpr uint64_t kDualMaxRecurrences = 5120;
bool pick_dual(const AADHipContext *ctx, const aad::EncodeArgs &a, bool quad)
{
  if (!quad || a.trials == 0) return false;
  if (a.trial_scratch == nullptr) return false; /* run_encode could not provide the slots */
  if ((uint64_t)a.num_streams * a.channels > kDualMaxRecurrences) return false;
  return ctx->trial_lanes != AAD_HIP_TRIAL_LANES_SINGLE;
}

/* the dual trial search keeps up to two alternative encodes of a block (and what measuring lanes write)
 * beside the image: three slots of one block per stream */
constexpr uint64_t kMaxTrialScratchBytes = 1ull << 30;
uint32_t trial_slot_bytes(const aad::EncodeArgs &a) { return (a.block_size + 16u + 63u) & ~63u; }

template <int BITS>
void launch_encode(const AADHipContext *ctx, const aad::EncodeArgs &a)
{
  const hipStream_t stream = ctx->stream;
  const uint64_t lanes = (uint64_t)a.num_streams * a.channels;
  const bool quad = pick_quad(ctx, lanes, a.channels, BITS);
  const bool dual = pick_dual(ctx, a, quad);
  const uint64_t threads = quad ? lanes * (dual ? 8 : 4) : lanes;
  /* dual: eight lanes per recurrence put a wave on twice as many CUs as the trial-free launch; two waves
   * per workgroup (two SIMDs of one CU) keep a small batch on half the chip, so that a decode launched
   * beside it finds free CUs (bench.py's pipelined step with trials 2: 158 -> see DESIGN.md) */
  /* (round 4: two-wave workgroups only while there is at most one of them per CU - 256 workgroups, 4096 recurrences.  Beyond that a
   * CU receives a second workgroup whose two waves land on the SIMDs the first one's already use, the other two SIMDs stay empty
   * and the launch takes 1.6x as long: stereo 4-bit, t = 2, 6000 recurrences 0.229 ms against 0.143 at 4096;
   * profiles/r04_trial_search_size_sweep.txt.  One-wave workgroups spread over the SIMDs.) */
  unsigned wg = dual && threads <= 32ull * 1024ull ? 128u : pick_workgroup(threads);
  if (!quad) /* the dense encoders, with and without the trial search: one-wave workgroups only while their LDS lets all of them be resident */
    wg = dense_encode_workgroup(lanes, a.channels == 1 ? (unsigned)aad::kLdsBytesEncoder<BITS, 1, false>
                                       : (a.channels == 2 ? (unsigned)aad::kLdsBytesEncoder<BITS, 2, false> : (unsigned)aad::kLdsBytesEncoder<BITS, 0, false>));
  const dim3 grid((unsigned)((threads + wg - 1) / wg)), block(wg);
  if (a.trials) {
    if (dual) launch_encode_mapped<BITS, true, true, true>(a, grid, block, stream);
    else if (quad) launch_encode_mapped<BITS, true, true, false>(a, grid, block, stream);
    else launch_encode_mapped<BITS, false, true, false>(a, grid, block, stream);
  } else {
    if (quad) launch_encode_mapped<BITS, true, false, false>(a, grid, block, stream);
    else {
      const bool ring = a.ring_ok && a.channels <= 2 && encode_ring_wanted(BITS, a.channels, wg);
      /* the rows' byte rings: dynamic LDS, one wave's worth per wave of the workgroup */
      const unsigned ring_lds = ring ? (wg / 64u) * (unsigned)(a.channels == 1 ? aad::kLdsRingBytesPerWave<1> : aad::kLdsRingBytesPerWave<2>) : 0u;
      const unsigned static_lds = ring ? (unsigned)aad::kLdsCodeStageOff + ring_lds
                                  : (a.channels == 1 ? (unsigned)aad::kLdsBytesEncoder<BITS, 1, false>
                                                     : (a.channels == 2 ? (unsigned)aad::kLdsBytesEncoder<BITS, 2, false> : (unsigned)aad::kLdsBytesEncoder<BITS, 0, false>));
      const unsigned pad = wg == 256u ? dense_encode_lds_pad(BITS, lanes, static_lds) : 0u;
      if (!(ring && launch_encode_ring<BITS, false>(a, grid, block, stream, ring_lds + pad)))
        launch_encode_mapped<BITS, false, false, false>(a, grid, block, stream, pad);
    }
  }
}

/* Occupancy cap of the per-lane dense decoders on chip-filling batches (what is left to them since the sector-tiled kernel:
 * 3-bit streams, layouts whose PCM is not 16-byte aligned, more than two channels): unused dynamic LDS up to 80 KB per
 * workgroup = two workgroups per CU = two waves per SIMD.  These kernels wait for memory (56-59 % VALU-active on mono 3-bit
 * streams), and what they wait for is lines that were evicted between two visits of the same lane: with fewer lanes resident
 * the L2 keeps more of them.  Mono 3-bit, 524 288 blocks: 2.00 -> 1.72 ms (688 -> 801 Gsamples/s), stereo 3-bit 0.762 -> 0.733 ms;
 * one wave per SIMD is slower again (1.75 / 0.86 ms); profiles/r03_decoder_occupancy_cap.txt.  AAD_HIP_DECODE_LDS_PAD (bytes, read once) overrides the policy. */
unsigned dense_decode_lds_pad(uint64_t lanes, uint32_t channels, uint32_t bits)
{
  static const int forced = [] {
    const char *e = getenv("AAD_HIP_DECODE_LDS_PAD");
    return e ? atoi(e) : -1;
  }();
  if (forced >= 0) return (unsigned)forced;
  constexpr unsigned kTarget = 80u << 10;
  /* same-box A/B of every geometry on this kernel: mono 4- / 3- / 2-bit +0 / +16 / +7 %, stereo 3- / 2-bit +4 / +2 %, stereo
   * 4-bit (streamed stores) -4 %: that one keeps its occupancy, and so do the any-channel launches (no change) */
  const bool gains = channels == 1 || (channels == 2 && bits != 4);
  return gains && lanes >= 65536 ? kTarget - (unsigned)aad::kLdsBytesDenseDec : 0u;
}

/* The dense stereo decoder's streamed (non-temporal) PCM stores: AAD_HIP_DECODE_NT_MIN (lanes, read once) is the batch size
 * from which they are taken where the layout allows (DecodeArgs::stream_stores); measurement aid, never changes a byte. */
uint64_t decode_nt_min_lanes()
{
  static const uint64_t v = [] {
    const char *e = getenv("AAD_HIP_DECODE_NT_MIN");
    return e ? (uint64_t)atoll(e) : 0ull;
  }();
  return v;
}

template <int BITS, bool QUAD>
void launch_decode_mapped(const aad::DecodeArgs &args, dim3 grid, dim3 block, hipStream_t stream)
{
  aad::DecodeArgs a = args;
  if (a.total_blocks * a.channels < decode_nt_min_lanes()) a.stream_stores = 0;
  const unsigned lds_pad = !QUAD && block.x == 256u ? dense_decode_lds_pad(a.total_blocks * a.channels, a.channels, a.bits) : 0u;
  if (a.channels == 1)
    AAD_LAUNCH((aad::decode_blocks_kernel<BITS, 1, false, QUAD>), grid, block, lds_pad, stream, a);
  else if (a.channels == 2 && a.mid_side) {
    if constexpr (!QUAD && BITS != 3) {
      if (a.stream_stores) {
        AAD_LAUNCH((aad::decode_blocks_kernel<BITS, 2, true, false, true>), grid, block, lds_pad, stream, a);
        return;
      }
    }
    AAD_LAUNCH((aad::decode_blocks_kernel<BITS, 2, true, QUAD>), grid, block, lds_pad, stream, a);
  } else if (a.channels == 2) {
    if constexpr (!QUAD && BITS != 3) {
      if (a.stream_stores) {
        AAD_LAUNCH((aad::decode_blocks_kernel<BITS, 2, false, false, true>), grid, block, lds_pad, stream, a);
        return;
      }
    }
    AAD_LAUNCH((aad::decode_blocks_kernel<BITS, 2, false, QUAD>), grid, block, lds_pad, stream, a);
  }
  else if constexpr (!QUAD)
    AAD_LAUNCH((aad::decode_blocks_kernel<BITS, 0, false, false>), grid, block, lds_pad, stream, a);
}

/* Quad decode runs its two strands on different lanes (aad_decode_split.hip.h) unless
 * AAD_HIP_MAPPING=quad-fused asks for the one-lane-does-both kernel or the residual scratch would be
 * unreasonably large. */
constexpr uint64_t kMaxResidualBytes = 1ull << 30;

/* Decode mapping by batch size: see mapping_limits.  The context option forces one. */
enum class DecodeMapping { Dense, QuadFused, QuadSplit };
/* Dense batches at and beyond this many recurrences take the sector-tiled kernel where it applies (aad_decode_tiled.hip.h).
 * Measured against the per-lane kernel (with its own occupancy cap) on one-block streams, same box, tools/saturated_probe.py
 * (profiles/r03_tiled_decode_crossover.txt): mono 4-bit wins from 65 536 blocks on (one wave per SIMD: 0.156 vs 0.198 ms;
 * 0.43 vs 0.48 ms at 196 608; 1.10 vs 1.29 ms at 524 288), stereo 4-bit from ~393 216 recurrences (0.42 vs 0.45 ms; equal at
 * 262 144, the per-lane kernel ahead below); mono 2-bit like mono 4-bit (2.15 vs 2.52 ms at 524 288 blocks); mono 3-bit (where
 * the batch's layout admits it: aad_decode_tiled.hip.h "3-bit rows") 1.41-1.43 vs 1.70 ms at 524 288 blocks.  On STEREO 2-bit
 * streams the tiled kernel moves 1.02x the algorithmic bytes instead of 1.35x but takes 3-7 % longer (1.04-1.08 vs 1.01-1.02 ms),
 * on stereo 3-bit streams both take 0.73 ms: "auto" keeps the per-lane kernel there, AAD_HIP_LANE_MAPPING_DENSE_TILED selects the
 * tiled one at any size. */
uint64_t tiled_decode_min(uint32_t bits, uint32_t channels)
{
  /* round 4 (tools/size_sweep.py --mapping dense | dense-tiled): at exactly 65 536 mono lanes the per-lane kernel still runs one
   * wave per SIMD in one-wave workgroups and is 7 % ahead (0.154 vs 0.165 ms); from the next lane on it needs a second wave per
   * SIMD and the tiled kernel is level (80 000) to 18 % ahead (524 288) */
  if (channels == 1) return 65537u;
  return bits == 4 ? 393216u : ~0ull;
}

DecodeMapping pick_decode_mapping(const AADHipContext *ctx, uint64_t recurrences, uint32_t channels, uint32_t bits)
{
  if (channels > 2) return DecodeMapping::Dense;
  switch (ctx->lane_mapping) {
    case AAD_HIP_LANE_MAPPING_DENSE:
    case AAD_HIP_LANE_MAPPING_DENSE_TILED: return DecodeMapping::Dense;
    case AAD_HIP_LANE_MAPPING_QUAD_FUSED: return DecodeMapping::QuadFused;
    case AAD_HIP_LANE_MAPPING_QUAD: return DecodeMapping::QuadSplit;
    default: break;
  }
  const MappingLimits m = mapping_limits(bits, channels);
  if (recurrences <= m.decode_split) return DecodeMapping::QuadSplit;
  if (recurrences <= m.decode_fused) return DecodeMapping::QuadFused;
  return DecodeMapping::Dense;
}

bool want_split_decode(const AADHipContext *ctx, const aad::DecodeArgs &a, uint64_t *bytes, uint32_t *stride)
{
  const uint64_t recurrences = a.total_blocks * a.channels;
  if (pick_decode_mapping(ctx, recurrences, a.channels, a.bits) != DecodeMapping::QuadSplit) return false;
  /* 64-bit: samples_per_block comes straight from a file header and may be anything */
  const uint64_t coded = a.samples_per_block > 4 ? (uint64_t)a.samples_per_block - 4 : 0;
  const uint64_t row = (coded + 15u) / 16u * 16u + 16u;
  if (row > kMaxResidualBytes / sizeof(int32_t)) return false;
  *stride = (uint32_t)row;
  *bytes = recurrences * row * sizeof(int32_t);
  return *bytes <= kMaxResidualBytes; /* else the fused quad kernel */
}

template <int BITS>
void launch_decode(const AADHipContext *ctx, const aad::DecodeArgs &a, int32_t *residual, uint32_t residual_stride)
{
  const hipStream_t stream = ctx->stream;
  if (residual != nullptr && aad::launch_decode_split(a, residual, residual_stride, stream)) return;
  const uint64_t lanes = a.total_blocks * a.channels;
  /* a split decode that could not have its scratch buffer: the fused kernel when the quad mapping is
   * forced, the dense one otherwise */
  const DecodeMapping pick = pick_decode_mapping(ctx, lanes, a.channels, BITS);
  /* dense: the sector-tiled kernel for chip-filling batches (or when the option asks for it), where it applies */
  if (pick == DecodeMapping::Dense && ctx->lane_mapping != AAD_HIP_LANE_MAPPING_DENSE &&
      (ctx->lane_mapping == AAD_HIP_LANE_MAPPING_DENSE_TILED || lanes >= tiled_decode_min(BITS, a.channels)) && aad::launch_decode_tiled(a, stream))
    return;
  const bool quad = pick == DecodeMapping::QuadFused || (pick == DecodeMapping::QuadSplit && ctx->lane_mapping == AAD_HIP_LANE_MAPPING_QUAD);
  const uint64_t threads = quad ? lanes * 4 : lanes;
  const unsigned wg = pick_workgroup(threads);
  const dim3 grid((unsigned)((threads + wg - 1) / wg)), block(wg);
  if (quad) launch_decode_mapped<BITS, true>(a, grid, block, stream);
  else launch_decode_mapped<BITS, false>(a, grid, block, stream);
}

} /* namespace */

/* ---- plan construction without any device work ------------------------------------------------
 * Validation + launch arguments.  AADHip_*PlanCreate adds the table upload; the host-memory
 * calls put the tables into the block that carries the payload instead. */
namespace {

/* lead_frames: context frames at the head of every stream (EncodeArgs::lead_frames) - 0, or one block */
AADApiResult encode_plan_init(const struct AADEncodeParameter *parameter, uint32_t num_streams,
                              const struct AADHipStreamDesc *streams, aad::EncodeArgs *args, uint32_t lead_frames = 0,
                              bool streams_checked = false)
{
  AADHeaderInfo h;
  if (AADFormat_ParameterToHeader(parameter, 1, AAD_HIP_MAX_NUM_CHANNELS, &h) != AAD_APIRESULT_OK)
    return AAD_APIRESULT_INVALID_FORMAT;
  /* what AADEncoder_EncodeHeader would reject (bits == 1, zero rate, M/S on mono, ...) */
  if (!AADFormat_HeaderFieldsValid(&h, AAD_HIP_MAX_NUM_CHANNELS)) return AAD_APIRESULT_INVALID_FORMAT;
  if (h.ch_process_method == AAD_CH_PROCESS_METHOD_MS && h.num_channels != 2) return AAD_APIRESULT_INVALID_FORMAT;
  for (uint32_t i = 0; !streams_checked && i < num_streams; i++) {
    if (streams[i].num_samples <= lead_frames) return AAD_APIRESULT_INVALID_FORMAT; /* src/aad_encoder.c:157-159 */
    h.num_samples = streams[i].num_samples - lead_frames;
    if (streams[i].data_size < AADFormat_EncodedSize(&h)) return AAD_APIRESULT_INSUFFICIENT_BUFFER;
  }
  memset(args, 0, sizeof(*args));
  args->num_streams = num_streams;
  args->channels = h.num_channels;
  args->block_size = h.block_size;
  args->samples_per_block = h.num_samples_per_block;
  args->mid_side = h.ch_process_method == AAD_CH_PROCESS_METHOD_MS;
  args->trials = parameter->num_encode_trials;
  args->lead_frames = lead_frames;
  args->uni = detect_uniform(num_streams, streams);
  /* the rows' byte rings store whole 64-byte sectors of an image at their own addresses: images on 64-byte boundaries */
  args->ring_ok = 1;
  for (uint32_t i = 0; i < num_streams; i++)
    if (streams[i].data_offset % 64u != 0) {
      args->ring_ok = 0;
      break;
    }
  h.num_samples = 0;
  AADFormat_PutHeader(&h, args->header_template);
  args->bits = h.bits_per_sample;
  return AAD_APIRESULT_OK;
}

/* prefix: num_streams + 1 entries, the exclusive prefix sum of blocks per stream */
/* known_blocks: per-stream block counts of streams that an earlier call of this function has already
 * validated (the tiles of the host-memory path) - the checks and their divisions are skipped */
AADApiResult decode_plan_init(const struct AADHeaderInfo *format, int32_t has_file_header, uint32_t num_streams,
                              const struct AADHipStreamDesc *streams, uint64_t *prefix, aad::DecodeArgs *args,
                              const uint64_t *known_blocks = nullptr)
{
  AADHeaderInfo h = *format;
  h.num_samples = 1; /* per-stream counts come from the table */
  if (!AADFormat_HeaderAcceptedByDecoder(&h, AAD_HIP_MAX_NUM_CHANNELS)) return AAD_APIRESULT_INVALID_FORMAT;
  if (h.ch_process_method == AAD_CH_PROCESS_METHOD_MS && h.num_channels != 2) return AAD_APIRESULT_INVALID_FORMAT;
  uint64_t blocks = 0;
  const uint32_t head = has_file_header ? AAD_HEADER_SIZE : 0;
  for (uint32_t i = 0; known_blocks != nullptr && i < num_streams; i++) {
    prefix[i] = blocks;
    blocks += known_blocks[i];
  }
  for (uint32_t i = 0; known_blocks == nullptr && i < num_streams; i++) {
    prefix[i] = blocks;
    /* per-block loop counters on the device are 32-bit: a header that claims a block of 2^31
     * samples and more (nothing in the reference's checks forbids it) is refused here */
    if (!AADFormat_DecodeWorkBounded(&h, streams[i].num_samples)) return AAD_APIRESULT_INVALID_FORMAT;
    /* the reference walks blocks while samples remain AND bytes remain (src/aad_decoder.c:514) */
    const uint64_t by_samples = ((uint64_t)streams[i].num_samples + h.num_samples_per_block - 1) / h.num_samples_per_block;
    const uint64_t payload = streams[i].data_size > head ? streams[i].data_size - head : 0;
    const uint64_t by_bytes = (payload + h.block_size - 1) / h.block_size;
    const uint64_t nblk = by_samples < by_bytes ? by_samples : by_bytes;
    /* a present block shorter than its header is the reference's INSUFFICIENT_DATA (src/aad_decoder.c:347-349) */
    if (nblk > 0) {
      const uint64_t last_bytes = payload - (nblk - 1) * h.block_size;
      if (last_bytes < (uint64_t)AAD_BLOCK_HEADER_BYTES_PER_CH * h.num_channels) return AAD_APIRESULT_INSUFFICIENT_DATA;
    }
    blocks += nblk;
  }
  prefix[num_streams] = blocks;
  memset(args, 0, sizeof(*args));
  args->total_blocks = blocks;
  args->num_streams = num_streams;
  args->channels = h.num_channels;
  args->block_size = h.block_size;
  args->samples_per_block = h.num_samples_per_block;
  args->header_bytes = head;
  args->uni = detect_uniform(num_streams, streams);
  if (args->uni.enabled) {
    args->uni.blocks_per_stream = (uint32_t)(num_streams ? prefix[1] - prefix[0] : 0);
    if (args->uni.blocks_per_stream == 0 || blocks >= 0xFFFFFFFFull) args->uni.enabled = 0;
  }
  args->mid_side = h.ch_process_method == AAD_CH_PROCESS_METHOD_MS;
  args->bits = h.bits_per_sample;
  /* the sector-tiled dense decoder moves PCM in 16-byte pieces: every stream's PCM must start on a piece boundary */
  args->pcm_aligned16 = 1;
  for (uint32_t i = 0; i < num_streams; i++)
    if ((streams[i].pcm_offset * 2u) % 16u != 0) {
      args->pcm_aligned16 = 0;
      break;
    }
  /* ... and its 3-bit rows want every block's code bytes at the same offset inside a granule (64 bytes mono, 128 stereo): all
   * images start at the same offset inside one, and so does every block of an image */
  {
    const uint32_t granule = h.num_channels == 1 ? 64u : 128u;
    args->code_phase_uniform = (h.block_size % granule == 0) ? 1 : 2; /* 2: one-block streams only (checked at launch) */
    for (uint32_t i = 1; i < num_streams; i++)
      if ((streams[i].data_offset - streams[0].data_offset) % granule != 0) {
        args->code_phase_uniform = 0;
        break;
      }
  }
  /* Dense stereo 4-/2-bit decode opens every block with a 16-frame chunk, so its stores are whole
   * 64-byte granules exactly when every block's first frame is 64-byte aligned: a uniform layout
   * whose stream pitch and block length (in PCM bytes) are multiples of 64.  Only then are they
   * issued non-temporal (the base pointer is checked at run time). */
  args->stream_stores = args->uni.enabled && h.num_channels == 2 && h.bits_per_sample != 3 &&
                        (args->uni.pcm_base * 2) % 64 == 0 && (args->uni.pcm_stride * 2) % 64 == 0 &&
                        ((uint64_t)h.num_samples_per_block * 4) % 64 == 0;
  return AAD_APIRESULT_OK;
}

/* launch with fully populated arguments (device pointers set) on the context's stream */
/* AADHip_ContextSignalNextRun: the pending events, taken by the plan run that starts now ... */
aad::LaunchSignal take_signal(AADHipContext *ctx)
{
  const aad::LaunchSignal s = ctx->signal_next;
  ctx->signal_next = aad::LaunchSignal{nullptr, nullptr};
  return s;
}
/* ... and settled when it ends: a run that launched its kernel has handed the events to it (aad_launch.h); one that launched
 * nothing (an empty plan) records them behind whatever the stream holds; a failed run leaves them unrecorded */
AADApiResult finish_signal(AADHipContext *ctx, const aad::LaunchSignal &signal, AADApiResult rc)
{
  const bool taken = aad::tl_launch_signal.start == nullptr && aad::tl_launch_signal.stop == nullptr;
  aad::tl_launch_signal = aad::LaunchSignal{nullptr, nullptr};
  if (taken || rc != AAD_APIRESULT_OK) return rc;
  if (signal.start != nullptr && !hip_ok(ctx, hipEventRecord(signal.start, ctx->stream), "hipEventRecord")) return AAD_APIRESULT_NG;
  if (signal.stop != nullptr && !hip_ok(ctx, hipEventRecord(signal.stop, ctx->stream), "hipEventRecord")) return AAD_APIRESULT_NG;
  return rc;
}

AADApiResult run_encode(AADHipContext *ctx, const aad::EncodeArgs &args)
{
  if (args.num_streams == 0) return AAD_APIRESULT_OK;
  aad::EncodeArgs a = args;
  a.trial_scratch = nullptr;
  a.trial_slot_bytes = 0;
  if ((reinterpret_cast<uintptr_t>(a.data) & 63u) != 0) a.ring_ok = 0;
  if (a.trials != 0 && a.channels <= 2 && ctx->trial_lanes != AAD_HIP_TRIAL_LANES_SINGLE &&
      (uint64_t)a.num_streams * a.channels <= kDualMaxRecurrences &&
      pick_quad(ctx, (uint64_t)a.num_streams * a.channels, a.channels, a.bits)) {
    const uint64_t want = (uint64_t)a.num_streams * 3u * trial_slot_bytes(a);
    if (want <= kMaxTrialScratchBytes) { /* else: the search and the encode one after the other on the same lanes */
      if (ctx->trial_capacity < want) {
        if (ctx->d_trial) {
          if (!hip_ok(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize")) return AAD_APIRESULT_NG;
          (void)hipFree(ctx->d_trial);
          ctx->d_trial = nullptr;
          ctx->trial_capacity = 0;
        }
        if (!hip_ok(ctx, hipMalloc((void **)&ctx->d_trial, want), "hipMalloc trial scratch")) return AAD_APIRESULT_NG;
        ctx->trial_capacity = want;
      }
      a.trial_scratch = ctx->d_trial;
      a.trial_slot_bytes = trial_slot_bytes(a);
    }
  }
  switch (a.bits) {
    case 4: launch_encode<4>(ctx, a); break;
    case 3: launch_encode<3>(ctx, a); break;
    case 2: launch_encode<2>(ctx, a); break;
    default: return AAD_APIRESULT_INVALID_FORMAT;
  }
  return hip_ok(ctx, hipGetLastError(), "encode launch") ? AAD_APIRESULT_OK : AAD_APIRESULT_NG;
}

AADApiResult run_decode(AADHipContext *ctx, const aad::DecodeArgs &a)
{
  if (a.total_blocks == 0) return AAD_APIRESULT_OK;
  uint64_t residual_bytes = 0;
  uint32_t residual_stride = 0;
  int32_t *residual = nullptr;
  const bool split = want_split_decode(ctx, a, &residual_bytes, &residual_stride);
  const bool split_in_lds = split && aad::decode_split_fits_lds(a);
  if (split && !split_in_lds) {
    if (ctx->residual_capacity < residual_bytes) { /* first such decode of this size on the context */
      if (ctx->d_residual) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(ctx->d_residual);
        ctx->d_residual = nullptr;
        ctx->residual_capacity = 0;
      }
      if (!hip_ok(ctx, hipMalloc((void **)&ctx->d_residual, residual_bytes), "hipMalloc residual scratch")) return AAD_APIRESULT_NG;
      ctx->residual_capacity = residual_bytes;
    }
    residual = ctx->d_residual;
  }
  if (split_in_lds) {
    if (!aad::launch_decode_split(a, nullptr, 0, ctx->stream)) return AAD_APIRESULT_INVALID_FORMAT;
    return hip_ok(ctx, hipGetLastError(), "decode launch") ? AAD_APIRESULT_OK : AAD_APIRESULT_NG;
  }
  switch (a.bits) {
    case 4: launch_decode<4>(ctx, a, residual, residual_stride); break;
    case 3: launch_decode<3>(ctx, a, residual, residual_stride); break;
    case 2: launch_decode<2>(ctx, a, residual, residual_stride); break;
    default: return AAD_APIRESULT_INVALID_FORMAT;
  }
  return hip_ok(ctx, hipGetLastError(), "decode launch") ? AAD_APIRESULT_OK : AAD_APIRESULT_NG;
}

void staging_pool_stop(AADHipContext *ctx)
{
  StagingPool *p = ctx->pool;
  if (p == nullptr) return;
  {
    std::lock_guard<std::mutex> g(p->lock);
    p->stop = true;
  }
  p->work.notify_all();
  for (auto &t : p->threads) t.join();
  delete p;
  ctx->pool = nullptr;
}

int32_t option_from_env(const char *name, const char *const *words, int32_t count)
{
  const char *e = getenv(name);
  if (e == nullptr) return 0;
  for (int32_t i = 0; i < count; i++)
    if (strcmp(e, words[i]) == 0) return i;
  return 0;
}

} /* namespace */

extern "C" {

int32_t AADHip_GetDeviceCount(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

AADApiResult AADHip_ContextCreate(int32_t device_index, void *hip_stream, struct AADHipContext **context)
{
  if (context == nullptr) return AAD_APIRESULT_INVALID_ARGUMENT;
  *context = nullptr;
  if (device_index < 0 || device_index >= AADHip_GetDeviceCount()) return AAD_APIRESULT_NG;
  AADHipContext *ctx = new (std::nothrow) AADHipContext();
  if (ctx == nullptr) return AAD_APIRESULT_NG;
  ctx->device = device_index;
  ctx->stream = static_cast<hipStream_t>(hip_stream);
  ctx->owns_stream = false;
  ctx->last_error[0] = 0;
  ctx->have_events = false;
  ctx->have_pipeline = false;
  ctx->d_scratch = nullptr;
  ctx->scratch_capacity = 0;
  ctx->d_rc_in = ctx->d_rc_out = nullptr;
  ctx->rc_in_capacity = ctx->rc_out_capacity = 0;
  ctx->d_residual = nullptr;
  ctx->residual_capacity = 0;
  ctx->d_trial = nullptr;
  ctx->trial_capacity = 0;
  ctx->signal_next = aad::LaunchSignal{nullptr, nullptr};
  ctx->pool = nullptr;
  ctx->staging_threads = 0;
  ctx->tile_bytes = 0;
  ctx->d_state = nullptr;
  ctx->state_capacity = 0;
  /* the environment is consulted here (and when the legacy API takes a parked context back into
   * use), never on a launch path */
  AADHipInternal_ContextOptionsFromEnvironment(ctx);
  DeviceGuard guard(ctx);
  if (!guard.ok) {
    delete ctx;
    return AAD_APIRESULT_NG;
  }
  if (hip_stream == nullptr) {
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
      delete ctx;
      return AAD_APIRESULT_NG;
    }
    ctx->owns_stream = true;
  }
  *context = ctx;
  return AAD_APIRESULT_OK;
}

void AADHip_ContextDestroy(struct AADHipContext *ctx)
{
  if (ctx == nullptr) return;
  staging_pool_stop(ctx);
  {
    DeviceGuard guard(ctx);
    if (guard.ok) {
      (void)hipStreamSynchronize(ctx->stream);
      for (int b = 0; b < 2; b++) {
        staging_release(ctx->in[b]);
        staging_release(ctx->out[b]);
        if (ctx->have_events) (void)hipEventDestroy(ctx->chunk_done[b]);
        if (ctx->have_events && b == 0)
          for (int i = 0; i < 3; i++) (void)hipEventDestroy(ctx->piece_done[i]);
        if (ctx->have_pipeline) {
          (void)hipEventDestroy(ctx->uploaded[b]);
          (void)hipEventDestroy(ctx->computed[b]);
        }
      }
      if (ctx->have_pipeline) {
        (void)hipStreamDestroy(ctx->up_stream);
        (void)hipStreamDestroy(ctx->down_stream);
      }
      if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
      if (ctx->d_rc_in) (void)hipFree(ctx->d_rc_in);
      if (ctx->d_rc_out) (void)hipFree(ctx->d_rc_out);
      if (ctx->d_residual) (void)hipFree(ctx->d_residual);
      if (ctx->d_trial) (void)hipFree(ctx->d_trial);
      if (ctx->d_state) (void)hipFree(ctx->d_state);
      if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    }
  }
  delete ctx;
}

AADApiResult AADHip_ContextSynchronize(struct AADHipContext *ctx)
{
  if (ctx == nullptr) return AAD_APIRESULT_INVALID_ARGUMENT;
  DeviceGuard guard(ctx);
  if (!guard.ok) return AAD_APIRESULT_NG;
  return hip_ok(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize") ? AAD_APIRESULT_OK : AAD_APIRESULT_NG;
}

const char *AADHip_ContextLastError(const struct AADHipContext *ctx) { return ctx ? ctx->last_error : ""; }

void AADHipInternal_ContextOptionsFromEnvironment(struct AADHipContext *ctx)
{
  static const char *const kMappings[] = {"auto", "dense", "quad", "quad-fused", "dense-tiled"};
  static const char *const kTrialLanes[] = {"dual", "single"};
  ctx->signal_refused = AADHip_SignalNextRunSupported() == 0;
  ctx->lane_mapping = option_from_env("AAD_HIP_MAPPING", kMappings, 5);
  ctx->trial_lanes = option_from_env("AAD_HIP_TRIAL_LANES", kTrialLanes, 2);
  static const char *const kCompareOrders[] = {"auto", "sequential"};
  ctx->compare_sequential = option_from_env("AAD_HIP_COMPARE_ORDER", kCompareOrders, 2);
  const char *threads = getenv("AAD_HIP_STAGING_THREADS");
  if (threads != nullptr && threads[0] >= '1' && threads[0] <= '8' && threads[1] == '\0') ctx->staging_threads = threads[0] - '0';
  const char *tile = getenv("AAD_HIP_TILE_KBYTES");
  if (tile != nullptr) {
    const long long v = atoll(tile);
    if (v > 0 && v <= 0x7FFFFFFF) ctx->tile_bytes = (int64_t)v << 10;
  }
}

int32_t AADHipInternal_ContextDevice(const struct AADHipContext *ctx) { return ctx->device; }

/* ROC_SYSTEM_SCOPE_SIGNAL=0 makes the ROCm runtime give kernel dispatches DEVICE-scope completion signals.  The events of
 * AADHip_ContextSignalNextRun ARE the kernel's completion signal (hipExtLaunchKernelGGL's stop event), and a wait on one
 * from another queue (hipStreamWaitEvent on a second stream: the two-stream pipeline of bench.py / EncodeDecodePipeline)
 * then never returns - recorded in tools/experiments/README.md ("runtime knobs") and gpurun_out/runtime_knobs.txt of round 3.
 * The library does not try to work under that setting: it says no. */
int32_t AADHip_SignalNextRunSupported(void)
{
  const char *v = getenv("ROC_SYSTEM_SCOPE_SIGNAL");
  return (v != nullptr && v[0] == '0' && v[1] == '\0') ? 0 : 1;
}

AADApiResult AADHip_ContextSignalNextRun(struct AADHipContext *ctx, void *hip_start_event, void *hip_stop_event)
{
  if (ctx == nullptr) return AAD_APIRESULT_INVALID_ARGUMENT;
  if (ctx->signal_refused && (hip_start_event != nullptr || hip_stop_event != nullptr)) {
    snprintf(ctx->last_error, sizeof(ctx->last_error),
             "AADHip_ContextSignalNextRun: refused, ROC_SYSTEM_SCOPE_SIGNAL=0 (device-scope completion signals: a wait on the "
             "run's event from another stream would never return); record an event behind the run instead");
    return AAD_APIRESULT_NG;
  }
  ctx->signal_next = aad::LaunchSignal{static_cast<hipEvent_t>(hip_start_event), static_cast<hipEvent_t>(hip_stop_event)};
  return AAD_APIRESULT_OK;
}

AADApiResult AADHip_ContextSetOption(struct AADHipContext *ctx, int32_t option, int32_t value)
{
  if (ctx == nullptr) return AAD_APIRESULT_INVALID_ARGUMENT;
  switch (option) {
    case AAD_HIP_OPTION_LANE_MAPPING:
      if (value < AAD_HIP_LANE_MAPPING_AUTO || value > AAD_HIP_LANE_MAPPING_DENSE_TILED) return AAD_APIRESULT_INVALID_ARGUMENT;
      ctx->lane_mapping = value;
      return AAD_APIRESULT_OK;
    case AAD_HIP_OPTION_TRIAL_LANES:
      if (value != AAD_HIP_TRIAL_LANES_DUAL && value != AAD_HIP_TRIAL_LANES_SINGLE) return AAD_APIRESULT_INVALID_ARGUMENT;
      ctx->trial_lanes = value;
      return AAD_APIRESULT_OK;
    case AAD_HIP_OPTION_COMPARE_ORDER:
      if (value != 0 && value != 1) return AAD_APIRESULT_INVALID_ARGUMENT;
      ctx->compare_sequential = value;
      return AAD_APIRESULT_OK;
    case AAD_HIP_OPTION_STAGING_THREADS:
      if (value < 0 || value > 8) return AAD_APIRESULT_INVALID_ARGUMENT;
      ctx->staging_threads = value;
      return AAD_APIRESULT_OK;
    case AAD_HIP_OPTION_TILE_KBYTES:
      if (value < 0) return AAD_APIRESULT_INVALID_ARGUMENT;
      ctx->tile_bytes = (int64_t)value << 10;
      return AAD_APIRESULT_OK;
    default:
      return AAD_APIRESULT_INVALID_ARGUMENT;
  }
}

uint64_t AADHip_CalculateEncodedSize(const struct AADEncodeParameter *parameter, uint32_t num_samples)
{
  AADHeaderInfo h;
  if (num_samples == 0) return 0;
  if (AADFormat_ParameterToHeader(parameter, num_samples, AAD_HIP_MAX_NUM_CHANNELS, &h) != AAD_APIRESULT_OK) return 0;
  if (!AADFormat_HeaderFieldsValid(&h, AAD_HIP_MAX_NUM_CHANNELS)) return 0;
  return AADFormat_EncodedSize(&h);
}

/* ------------------------------------------------------------------------------- encode -- */

AADApiResult AADHip_EncodePlanCreate(struct AADHipContext *ctx, const struct AADEncodeParameter *parameter,
                                     uint32_t num_streams, const struct AADHipStreamDesc *streams,
                                     struct AADHipEncodePlan **plan)
{
  if (ctx == nullptr || parameter == nullptr || plan == nullptr || (num_streams != 0 && streams == nullptr))
    return AAD_APIRESULT_INVALID_ARGUMENT;
  *plan = nullptr;
  aad::EncodeArgs args;
  const AADApiResult rc = encode_plan_init(parameter, num_streams, streams, &args);
  if (rc != AAD_APIRESULT_OK) return rc;
  AADHipEncodePlan *p = new (std::nothrow) AADHipEncodePlan();
  if (p == nullptr) return AAD_APIRESULT_NG;
  p->ctx = ctx;
  p->d_streams = nullptr;
  DeviceGuard guard(ctx);
  if (!guard.ok || !upload(ctx, &p->d_streams, reinterpret_cast<const aad::StreamDesc *>(streams), num_streams)) {
    if (p->d_streams) (void)hipFree(p->d_streams);
    delete p;
    return AAD_APIRESULT_NG;
  }
  p->args = args;
  p->args.streams = p->d_streams;
  *plan = p;
  return AAD_APIRESULT_OK;
}

void AADHip_EncodePlanDestroy(struct AADHipEncodePlan *plan)
{
  if (plan == nullptr) return;
  DeviceGuard guard(plan->ctx);
  if (guard.ok) {
    (void)hipStreamSynchronize(plan->ctx->stream);
    (void)hipFree(plan->d_streams);
  }
  delete plan;
}

AADApiResult AADHip_EncodePlanRun(struct AADHipEncodePlan *plan, const int16_t *device_pcm, uint8_t *device_data,
                                  struct AADHipLaneState *device_state)
{
  if (plan == nullptr || device_pcm == nullptr || device_data == nullptr) return AAD_APIRESULT_INVALID_ARGUMENT;
  AADHipContext *ctx = plan->ctx;
  const aad::LaunchSignal signal = take_signal(ctx);
  DeviceGuard guard(ctx);
  if (!guard.ok) return AAD_APIRESULT_NG;
  if (plan->args.num_streams == 0) return finish_signal(ctx, signal, AAD_APIRESULT_OK);
  aad::EncodeArgs a = plan->args;
  a.pcm = device_pcm;
  a.data = device_data;
  a.state = reinterpret_cast<const aad::LaneStateRecord *>(device_state);
  a.state_out = reinterpret_cast<aad::LaneStateRecord *>(device_state);
  aad::tl_launch_signal = signal; /* the run's one kernel takes it (aad_launch.h) */
  return finish_signal(ctx, signal, run_encode(ctx, a));
}

/* ------------------------------------------------------------------------------- decode -- */

AADApiResult AADHip_DecodePlanCreate(struct AADHipContext *ctx, const struct AADHeaderInfo *format,
                                     int32_t has_file_header, uint32_t num_streams,
                                     const struct AADHipStreamDesc *streams, struct AADHipDecodePlan **plan)
{
  if (ctx == nullptr || format == nullptr || plan == nullptr || (num_streams != 0 && streams == nullptr))
    return AAD_APIRESULT_INVALID_ARGUMENT;
  *plan = nullptr;
  std::vector<uint64_t> prefix((size_t)num_streams + 1);
  aad::DecodeArgs args;
  const AADApiResult rc = decode_plan_init(format, has_file_header, num_streams, streams, prefix.data(), &args);
  if (rc != AAD_APIRESULT_OK) return rc;

  AADHipDecodePlan *p = new (std::nothrow) AADHipDecodePlan();
  if (p == nullptr) return AAD_APIRESULT_NG;
  p->ctx = ctx;
  p->d_streams = nullptr;
  p->d_prefix = nullptr;
  DeviceGuard guard(ctx);
  if (!guard.ok || !upload(ctx, &p->d_streams, reinterpret_cast<const aad::StreamDesc *>(streams), num_streams) ||
      !upload(ctx, &p->d_prefix, prefix.data(), prefix.size())) {
    if (p->d_streams) (void)hipFree(p->d_streams);
    if (p->d_prefix) (void)hipFree(p->d_prefix);
    delete p;
    return AAD_APIRESULT_NG;
  }
  p->args = args;
  p->args.streams = p->d_streams;
  p->args.block_prefix = p->d_prefix;
  *plan = p;
  return AAD_APIRESULT_OK;
}

void AADHip_DecodePlanDestroy(struct AADHipDecodePlan *plan)
{
  if (plan == nullptr) return;
  DeviceGuard guard(plan->ctx);
  if (guard.ok) {
    (void)hipStreamSynchronize(plan->ctx->stream);
    (void)hipFree(plan->d_streams);
    (void)hipFree(plan->d_prefix);
  }
  delete plan;
}

AADApiResult AADHip_DecodePlanRun(struct AADHipDecodePlan *plan, const uint8_t *device_data, int16_t *device_pcm)
{
  if (plan == nullptr || device_data == nullptr || device_pcm == nullptr) return AAD_APIRESULT_INVALID_ARGUMENT;
  AADHipContext *ctx = plan->ctx;
  const aad::LaunchSignal signal = take_signal(ctx);
  DeviceGuard guard(ctx);
  if (!guard.ok) return AAD_APIRESULT_NG;
  if (plan->args.total_blocks == 0) return finish_signal(ctx, signal, AAD_APIRESULT_OK);
  aad::DecodeArgs a = plan->args;
  a.data = device_data;
  a.pcm = device_pcm;
  if ((reinterpret_cast<uintptr_t>(device_pcm) & 63u) != 0) a.stream_stores = 0;
  aad::tl_launch_signal = signal;
  return finish_signal(ctx, signal, run_decode(ctx, a));
}

} /* extern "C" */

/* --------------------------------------------------------------- host-memory calls ----------
 *
 * Call pattern served: the reference CLI's (src/main.c:182-198 encode, :91-106 decode) - host
 * buffers in, host buffers out, synchronous - for one stream (legacy API) or many (AADHip_*Batch).
 *
 * One chunk of streams = one pinned input block {stream table, [block prefix], [state], payload}
 * -> ONE H2D copy -> kernel -> ONE D2H copy of the output block {payload, [state]}.  No hipMalloc,
 * no plan object, no synchronisation besides the wait for the chunk's last copy.  Batches above
 * kChunkBudget bytes are cut into chunks that alternate between two block pairs: the host fills
 * chunk k+1 and drains chunk k-1 while chunk k is on the device (the staging memcpys are the
 * slowest stage of this path - one core moves ~10 GB/s, PCIe ~50 GB/s, the kernels more).
 */
namespace {

uint64_t round_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

constexpr uint64_t kChunkBudget = 16ull << 20; /* payload bytes (in + out) per tile of a cut batch */
constexpr uint64_t kCutAbove = kChunkBudget;

constexpr uint64_t kParallelStagingAbove = 1ull << 20; /* bytes in a chunk from which the helper threads pay */

unsigned staging_threads(const AADHipContext *ctx)
{
  if (ctx->staging_threads > 0) return (unsigned)ctx->staging_threads;
  static const unsigned hw = std::thread::hardware_concurrency(); /* asked once: it reads /sys */
  /* round 4 (tools/host_size_sweep.py, 100 000 one-block stereo streams = 0.5 GB per direction): decode 7.9 / 10.9 / 12.0 / 13.6
   * Gsamples/s with 2 / 4 / 6 / 8 copying threads - the drain of 2 bytes per sample into pageable memory is the slowest stage -
   * so a host with cores to spare takes eight */
  return hw >= 32 ? 8u : (hw >= 8 ? 4u : (hw >= 4 ? 2u : 1u));
}

void staging_pool_main(StagingPool *p, unsigned index)
{
  unsigned seen = 0;
  for (;;) {
    std::function<void(unsigned)> job;
    {
      std::unique_lock<std::mutex> g(p->lock);
      p->work.wait(g, [&] { return p->stop || p->generation != seen; });
      if (p->stop) return;
      seen = p->generation;
      job = p->job;
    }
    job(index);
    {
      std::lock_guard<std::mutex> g(p->lock);
      if (--p->pending == 0) p->done.notify_one();
    }
  }
}

/* Run body(first, last) over [0, count) cut into contiguous ranges of about equal cost, one per thread
 * (the caller's included).  `prefix` has count + 1 entries: the running cost. */
template <class Body>
void staged_span(AADHipContext *ctx, uint32_t first, uint32_t count, const std::vector<uint64_t> &prefix, Body body)
{
  const uint64_t base = prefix[first];
  const unsigned want = count - first < 2 || prefix[count] - base < kParallelStagingAbove ? 1u : staging_threads(ctx);
  if (want <= 1) {
    body(first, count);
    return;
  }
  if (ctx->pool != nullptr && ctx->pool->threads.size() + 1 != want) staging_pool_stop(ctx);
  if (ctx->pool == nullptr) {
    ctx->pool = new (std::nothrow) StagingPool();
    if (ctx->pool == nullptr) {
      body(first, count);
      return;
    }
    /* These entry points are extern "C": an exception must not leave them.  A thread that cannot be created
     * (RLIMIT_NPROC, a cgroup's pids limit: std::system_error) ends the pool - the threads that did start are joined - and
     * the caller copies alone. */
    try {
      for (unsigned t = 1; t < want; t++) ctx->pool->threads.emplace_back(staging_pool_main, ctx->pool, t);
    } catch (...) {
      staging_pool_stop(ctx);
      body(first, count);
      return;
    }
  }
  StagingPool *p = ctx->pool;
  const unsigned parts = (unsigned)p->threads.size() + 1;
  auto bound = [&](unsigned t) -> uint32_t { /* first item of part t */
    if (t >= parts) return count;
    if (t == 0) return first;
    const uint64_t target = base + (prefix[count] - base) / parts * t;
    uint32_t lo = first, hi = count;
    while (lo < hi) {
      const uint32_t mid = lo + (hi - lo) / 2;
      if (prefix[mid] < target) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  auto part = [&](unsigned t) {
    const uint32_t a = bound(t), b = bound(t + 1);
    if (b > a) body(a, b);
  };
  {
    std::lock_guard<std::mutex> g(p->lock);
    p->job = part;
    p->pending = parts - 1;
    p->generation++;
  }
  p->work.notify_all();
  part(0);
  std::unique_lock<std::mutex> g(p->lock);
  p->done.wait(g, [&] { return p->pending == 0; });
}

template <class Body>
void staged_ranges(AADHipContext *ctx, uint32_t count, const std::vector<uint64_t> &prefix, Body body)
{
  staged_span(ctx, 0u, count, prefix, body);
}

/* A batch that travels as ONE tile has nothing to overlap with, so its big copy is cut into pieces
 * instead: encode sends piece p up while the host fills piece p + 1, decode drains piece p while
 * piece p + 1 comes down.  Item index where each piece ends, by running bytes. */
constexpr uint32_t kMaxPieces = 4;
constexpr uint64_t kPieceBytes = 1ull << 20;

uint32_t cut_pieces(const std::vector<uint64_t> &prefix, uint32_t count, bool wanted, uint32_t *end)
{
  const uint64_t total = prefix[count];
  uint32_t pieces = wanted ? (uint32_t)(total / kPieceBytes) : 1u;
  pieces = pieces < 1 ? 1 : (pieces > kMaxPieces ? kMaxPieces : pieces);
  uint32_t at = 0, made = 0;
  for (uint32_t p = 1; p < pieces; p++) {
    const uint64_t target = total / pieces * p;
    while (at < count && prefix[at] < target) at++;
    if (at > (made ? end[made - 1] : 0u) && at < count) end[made++] = at;
  }
  end[made++] = count;
  return made;
}

bool ensure_events(AADHipContext *ctx)
{
  if (ctx->have_events) return true;
  if (!hip_ok(ctx, hipEventCreateWithFlags(&ctx->chunk_done[0], hipEventDisableTiming), "hipEventCreate")) return false;
  if (!hip_ok(ctx, hipEventCreateWithFlags(&ctx->chunk_done[1], hipEventDisableTiming), "hipEventCreate")) {
    (void)hipEventDestroy(ctx->chunk_done[0]);
    return false;
  }
  for (int i = 0; i < 3; i++)
    if (!hip_ok(ctx, hipEventCreateWithFlags(&ctx->piece_done[i], hipEventDisableTiming), "hipEventCreate")) {
      for (int k = 0; k < i; k++) (void)hipEventDestroy(ctx->piece_done[k]);
      (void)hipEventDestroy(ctx->chunk_done[0]);
      (void)hipEventDestroy(ctx->chunk_done[1]);
      return false;
    }
  ctx->have_events = true;
  return true;
}

/* the copy streams and their events: made at a context's first cut batch (a process that only ever
 * sends small calls keeps one stream - every extra one costs the runtime a little on each call) */
bool ensure_pipeline(AADHipContext *ctx)
{
  if (ctx->have_pipeline) return true;
  hipEvent_t ev[4];
  int made = 0;
  for (; made < 4; made++)
    if (!hip_ok(ctx, hipEventCreateWithFlags(&ev[made], hipEventDisableTiming), "hipEventCreate")) break;
  bool ok = made == 4;
  if (ok && !hip_ok(ctx, hipStreamCreateWithFlags(&ctx->up_stream, hipStreamNonBlocking), "hipStreamCreate")) ok = false;
  if (ok && !hip_ok(ctx, hipStreamCreateWithFlags(&ctx->down_stream, hipStreamNonBlocking), "hipStreamCreate")) {
    (void)hipStreamDestroy(ctx->up_stream);
    ok = false;
  }
  if (!ok) {
    for (int i = 0; i < made; i++) (void)hipEventDestroy(ev[i]);
    return false;
  }
  for (int b = 0; b < 2; b++) {
    ctx->uploaded[b] = ev[b];
    ctx->computed[b] = ev[2 + b];
  }
  ctx->have_pipeline = true;
  return true;
}

/* the streams a tile's three stages run on: all the context's own for a batch that goes as one
 * tile (no cross-stream hop on the latency path of small calls), three different ones for a cut batch */
struct Route {
  hipStream_t up, run, down;
};

Route route_for(const AADHipContext *ctx, bool piped)
{
  if (piped) return {ctx->up_stream, ctx->stream, ctx->down_stream};
  return {ctx->stream, ctx->stream, ctx->stream};
}

/* make `to` wait for what `from` holds so far */
bool hop(AADHipContext *ctx, hipEvent_t event, hipStream_t from, hipStream_t to)
{
  if (from == to) return true;
  return hip_ok(ctx, hipEventRecord(event, from), "hipEventRecord") &&
         hip_ok(ctx, hipStreamWaitEvent(to, event, 0), "hipStreamWaitEvent");
}

void settle(const Route &r)
{
  (void)hipStreamSynchronize(r.up);
  (void)hipStreamSynchronize(r.run);
  (void)hipStreamSynchronize(r.down);
}

/*
 * How a batch is cut.  A stream's blocks are chained in the encoder (block k starts from the
 * predictor block k-1 left behind), so one launch costs about `blocks per stream` x 64 us however
 * few streams it holds: cutting a batch of long streams BY STREAM would pay that chain once per cut.
 * The batch is therefore cut both ways: consecutive streams form a GROUP until one block of each
 * fills the tile budget, and a group is walked in TILES of `budget / (streams alive x block bytes)`
 * blocks of every stream at once, the predictor state staying on the device between tiles (encode;
 * decode blocks are independent and only share the tiling).  Inside a group streams are ordered
 * longest first, so that the streams still alive at any block are a prefix of that order and a
 * stream's state record keeps its index for the whole group.
 */
struct TileStep {
  uint32_t alive;          /* streams in the tile: order[0 .. alive) */
  uint64_t block0, block1; /* blocks [block0, block1) of each */
  bool group_first, group_last;
};

struct TilePlanner {
  const uint64_t *blocks; /* per stream: blocks to walk (0 = nothing to do) */
  uint32_t n;
  uint64_t block_cost, budget;
  uint32_t next_stream = 0;
  bool in_group = false;
  uint64_t block0 = 0;
  std::vector<uint32_t> order; /* the current group, longest stream first */

  TilePlanner(const uint64_t *blocks_per_stream, uint32_t num_streams, uint64_t bytes_per_block, uint64_t tile_budget)
      : blocks(blocks_per_stream), n(num_streams), block_cost(bytes_per_block ? bytes_per_block : 1), budget(tile_budget) {}

  bool next(TileStep *t)
  {
    for (;;) {
      bool first = false;
      if (!in_group) {
        if (next_stream >= n) return false;
        order.clear();
        uint64_t cost = 0;
        do {
          order.push_back(next_stream);
          if (blocks[next_stream]) cost += block_cost;
          next_stream++;
        } while (next_stream < n && cost < budget);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return blocks[x] > blocks[y]; });
        block0 = 0;
        in_group = true;
        first = true;
      }
      /* streams with more than block0 blocks: a prefix of the order */
      uint32_t lo = 0, hi = (uint32_t)order.size();
      while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (blocks[order[mid]] > block0) lo = mid + 1; else hi = mid;
      }
      if (lo == 0) { /* a group of empty streams */
        in_group = false;
        continue;
      }
      const uint64_t longest = blocks[order[0]];
      uint64_t per_tile = budget / ((uint64_t)lo * block_cost);
      if (per_tile == 0) per_tile = 1;
      t->alive = lo;
      t->block0 = block0;
      t->block1 = longest - block0 <= per_tile ? longest : block0 + per_tile;
      t->group_first = first;
      t->group_last = t->block1 >= longest;
      block0 = t->block1;
      if (t->group_last) in_group = false;
      return true;
    }
  }
};

/* what the host still has to do with a tile once its copies are back */
struct Delivery {
  uint32_t stream;
  uint64_t src;   /* offset in the pair's output block (encode: bytes, decode: int16 elements) */
  uint64_t dst;   /* encode: byte offset in the caller's image; decode: first frame */
  uint64_t count; /* encode: bytes; decode: frames */
};

struct Flight {
  bool active = false;
  uint64_t sequence = 0;
  std::vector<Delivery> items;
  std::vector<uint64_t> cost;        /* running bytes over items: [items + 1] */
  std::vector<uint32_t> state_order; /* encode, last tile of a group: the group's order, else empty */
  uint64_t state_src = 0;
  uint32_t pieces = 1, piece_end[4] = {0, 0, 0, 0}; /* decode: item index where each piece of the output copy ends */
};

bool batch_is_cut(const AADHipContext *ctx, uint64_t total_bytes) { return ctx->tile_bytes > 0 || total_bytes > kCutAbove; }

uint64_t tile_budget(const AADHipContext *ctx, uint64_t total_bytes)
{
  if (ctx->tile_bytes > 0) return (uint64_t)ctx->tile_bytes; /* forced: tests, tuning */
  return total_bytes > kCutAbove ? kChunkBudget : ~0ull >> 8;
}

bool state_reserve(AADHipContext *ctx, size_t records)
{
  if (records <= ctx->state_capacity) return true;
  if (ctx->d_state) (void)hipFree(ctx->d_state); /* waits for the device: nothing is still reading it */
  ctx->d_state = nullptr;
  ctx->state_capacity = 0;
  const size_t want = records + records / 4 + 64;
  if (!hip_ok(ctx, hipMalloc(&ctx->d_state, sizeof(AADHipLaneState) * want), "hipMalloc lane states")) return false;
  ctx->state_capacity = want;
  return true;
}

/*
 * Encode num_streams host streams.  fill(i, frame0, frames, dst) writes that range of stream i's
 * interleaved int16 frames, drain(i, offset, src, size) receives bytes [offset, offset + size) of
 * its .aad image.  `state` as in AADHip_EncodeBatch.
 */
template <class Fill, class Drain>
AADApiResult encode_host(AADHipContext *ctx, const struct AADEncodeParameter *parameter, uint32_t num_streams,
                         const uint32_t *num_samples, const uint64_t *data_capacity, uint64_t *output_size,
                         struct AADHipLaneState *state, Fill fill, Drain drain)
{
  AADHeaderInfo h;
  if (AADFormat_ParameterToHeader(parameter, 1, AAD_HIP_MAX_NUM_CHANNELS, &h) != AAD_APIRESULT_OK ||
      !AADFormat_HeaderFieldsValid(&h, AAD_HIP_MAX_NUM_CHANNELS))
    return AAD_APIRESULT_INVALID_FORMAT;
  const uint32_t ch = h.num_channels, spb = h.num_samples_per_block;
  std::vector<uint64_t> sizes(num_streams), blocks(num_streams);
  uint64_t total = 0;
  for (uint32_t i = 0; i < num_streams; i++) {
    if (num_samples[i] == 0) return AAD_APIRESULT_INVALID_FORMAT; /* src/aad_encoder.c:157-159 */
    h.num_samples = num_samples[i];
    sizes[i] = AADFormat_EncodedSize(&h);
    if (data_capacity[i] < sizes[i]) return AAD_APIRESULT_INSUFFICIENT_BUFFER;
    blocks[i] = ((uint64_t)num_samples[i] + spb - 1) / spb;
    total += (uint64_t)num_samples[i] * ch * sizeof(int16_t) + sizes[i];
  }
  DeviceGuard guard(ctx);
  if (!guard.ok || !ensure_events(ctx)) return AAD_APIRESULT_NG;

  TilePlanner planner(blocks.data(), num_streams, (uint64_t)spb * ch * sizeof(int16_t) + h.block_size, tile_budget(ctx, total));
  const bool piped = batch_is_cut(ctx, total);
  if (piped && !ensure_pipeline(ctx)) return AAD_APIRESULT_NG;
  const Route route = route_for(ctx, piped);
  Flight flight[2];
  std::vector<AADHipStreamDesc> table;
  std::vector<uint64_t> fill_cost;
  uint64_t sequence = 0;

  /* wait for the tile on pair b and hand its image bytes (and states) to the caller */
  auto finish = [&](int b) -> bool {
    Flight &f = flight[b];
    if (!f.active) return true;
    f.active = false;
    if (!hip_ok(ctx, hipEventSynchronize(ctx->chunk_done[b]), "hipEventSynchronize")) return false;
    const uint8_t *out = static_cast<const uint8_t *>(ctx->out[b].host);
    staged_ranges(ctx, (uint32_t)f.items.size(), f.cost, [&](uint32_t lo, uint32_t hi) {
      for (uint32_t k = lo; k < hi; k++) {
        const Delivery &d = f.items[k];
        drain(d.stream, d.dst, out + d.src, d.count);
        if (d.dst == 0 && d.count < sizes[d.stream]) { /* first tile of a longer stream: the header carries the whole count */
          const uint32_t all = num_samples[d.stream];
          const uint8_t be[4] = {(uint8_t)(all >> 24), (uint8_t)(all >> 16), (uint8_t)(all >> 8), (uint8_t)all};
          drain(d.stream, 14, be, 4);
        }
      }
    });
    if (state) {
      const AADHipLaneState *records = reinterpret_cast<const AADHipLaneState *>(out + f.state_src);
      for (size_t slot = 0; slot < f.state_order.size(); slot++)
        memcpy(state + (size_t)f.state_order[slot] * ch, records + slot * ch, sizeof(AADHipLaneState) * ch);
    }
    return true;
  };

  AADApiResult rc = AAD_APIRESULT_OK;
  TileStep step;
  while (rc == AAD_APIRESULT_OK && planner.next(&step)) {
    const int b = (int)(sequence & 1);
    if (!finish(b)) { rc = AAD_APIRESULT_NG; break; }
    const uint32_t n = step.alive;
    const std::vector<uint32_t> &order = planner.order;
    const uint32_t group_size = (uint32_t)order.size();
    Flight &f = flight[b];
    /* input block: table | state | pcm ; output block: image slices | state */
    table.resize(n);
    f.items.resize(n);
    f.cost.resize((size_t)n + 1);
    fill_cost.resize((size_t)n + 1);
    uint64_t pcm_elems = 0, data_bytes = 0;
    /* the trial search looks one block back in the input: later tiles bring that block along */
    const uint32_t lead = parameter->num_encode_trials > 0 && step.block0 > 0 ? spb : 0;
    for (uint32_t k = 0; k < n; k++) {
      const uint32_t i = order[k];
      const uint64_t frame0 = step.block0 * spb;
      const uint64_t frame1 = step.block1 * spb < num_samples[i] ? step.block1 * spb : num_samples[i];
      /* file header + this tile's blocks: up to the stream's end, or whole blocks */
      const uint64_t slice = AAD_HEADER_SIZE + (step.block1 >= blocks[i] ? sizes[i] - AAD_HEADER_SIZE - step.block0 * h.block_size
                                                                         : (step.block1 - step.block0) * h.block_size);
      h.num_samples = (uint32_t)(frame1 - frame0) + lead;
      table[k].pcm_offset = pcm_elems;
      table[k].data_offset = data_bytes;
      table[k].data_size = slice;
      table[k].num_samples = h.num_samples;
      table[k].reserved = 0;
      /* the first tile delivers the file header too; later ones only their blocks */
      const uint64_t skip = step.block0 ? AAD_HEADER_SIZE : 0;
      f.items[k] = {i, data_bytes + skip, step.block0 ? AAD_HEADER_SIZE + step.block0 * h.block_size : 0, slice - skip};
      f.cost[k] = data_bytes;
      fill_cost[k] = pcm_elems * sizeof(int16_t);
      pcm_elems += round_up((uint64_t)h.num_samples * ch, 8);
      data_bytes += round_up(slice, 16);
    }
    f.cost[n] = data_bytes;
    fill_cost[n] = pcm_elems * sizeof(int16_t);
    aad::EncodeArgs a;
    rc = encode_plan_init(parameter, n, table.data(), &a, lead, true);
    if (rc != AAD_APIRESULT_OK) break;
    /* states live on the device while a group has tiles to go; the caller's come in with the first
     * tile and leave with the last */
    const bool lone = step.group_first && step.group_last; /* the group's only tile: states travel inside its blocks */
    const bool carry = !lone;
    const bool state_in = state != nullptr && step.group_first, state_back = state != nullptr && step.group_last;
    const size_t table_bytes = round_up(sizeof(AADHipStreamDesc) * (size_t)n, 64);
    const size_t state_in_bytes = state_in ? sizeof(AADHipLaneState) * (size_t)n * ch : 0;
    const size_t state_back_bytes = state_back ? sizeof(AADHipLaneState) * (size_t)group_size * ch : 0;
    const size_t pcm_off = table_bytes + round_up(state_in_bytes, 64);
    const size_t in_bytes = pcm_off + pcm_elems * sizeof(int16_t);
    const size_t out_bytes = data_bytes + state_back_bytes;
    rc = AAD_APIRESULT_NG;
    if (!staging_reserve(ctx, ctx->in[b], in_bytes + 64) || !staging_reserve(ctx, ctx->out[b], out_bytes + 64)) break;
    if (carry && !state_reserve(ctx, (size_t)group_size * ch)) break;
    uint8_t *hin = static_cast<uint8_t *>(ctx->in[b].host), *din = static_cast<uint8_t *>(ctx->in[b].dev);
    uint8_t *dout = static_cast<uint8_t *>(ctx->out[b].dev);
    memcpy(hin, table.data(), sizeof(AADHipStreamDesc) * (size_t)n);
    if (state_in) /* every stream of a group is alive in its first tile: n == group_size */
      for (uint32_t k = 0; k < n; k++)
        memcpy(hin + table_bytes + sizeof(AADHipLaneState) * (size_t)k * ch, state + (size_t)order[k] * ch, sizeof(AADHipLaneState) * ch);
    {
      uint32_t piece_end[kMaxPieces];
      const uint32_t pieces = cut_pieces(fill_cost, n, !piped, piece_end);
      size_t sent = 0;
      bool ok = true;
      for (uint32_t p = 0, lo = 0; ok && p < pieces; lo = piece_end[p++]) {
        staged_span(ctx, lo, piece_end[p], fill_cost, [&](uint32_t x, uint32_t y) {
          for (uint32_t k = x; k < y; k++)
            fill(order[k], (uint32_t)(step.block0 * spb - lead), table[k].num_samples, reinterpret_cast<int16_t *>(hin + pcm_off) + table[k].pcm_offset);
        });
        const size_t upto = p + 1 == pieces ? in_bytes : pcm_off + (size_t)fill_cost[piece_end[p]];
        ok = hip_ok(ctx, hipMemcpyAsync(din + sent, hin + sent, upto - sent, hipMemcpyHostToDevice, route.up), "H2D block");
        sent = upto;
      }
      if (!ok) break;
    }
    if (!hop(ctx, ctx->uploaded[b], route.up, route.run)) break;
    aad::LaneStateRecord *d_state = static_cast<aad::LaneStateRecord *>(ctx->d_state);
    a.streams = reinterpret_cast<const aad::StreamDesc *>(din);
    a.pcm = reinterpret_cast<const int16_t *>(din + pcm_off);
    a.data = dout;
    a.state = state_in ? reinterpret_cast<const aad::LaneStateRecord *>(din + table_bytes) : (step.group_first ? nullptr : d_state);
    a.state_out = carry ? d_state : (state_back ? reinterpret_cast<aad::LaneStateRecord *>(dout + data_bytes) : nullptr);
    if (run_encode(ctx, a) != AAD_APIRESULT_OK) break;
    f.state_order.clear();
    if (state_back) {
      /* from the device-side records on the compute stream: the next group's first launch overwrites them */
      if (carry && !hip_ok(ctx, hipMemcpyAsync(static_cast<uint8_t *>(ctx->out[b].host) + data_bytes, d_state, state_back_bytes, hipMemcpyDeviceToHost, route.run), "D2H lane states")) break;
      f.state_order = order;
      f.state_src = data_bytes;
    }
    if (!hop(ctx, ctx->computed[b], route.run, route.down)) break;
    if (!hip_ok(ctx, hipMemcpyAsync(ctx->out[b].host, dout, lone ? out_bytes : data_bytes, hipMemcpyDeviceToHost, route.down), "D2H block")) break;
    if (!hip_ok(ctx, hipEventRecord(ctx->chunk_done[b], route.down), "hipEventRecord")) break;
    f.sequence = sequence++;
    f.active = true;
    rc = AAD_APIRESULT_OK;
  }
  /* drain what is still in flight, oldest first (also on failure: the buffers must be idle on return) */
  const int older = flight[0].active && flight[1].active && flight[1].sequence < flight[0].sequence ? 1 : 0;
  if (!finish(older) && rc == AAD_APIRESULT_OK) rc = AAD_APIRESULT_NG;
  if (!finish(older ^ 1) && rc == AAD_APIRESULT_OK) rc = AAD_APIRESULT_NG;
  if (rc != AAD_APIRESULT_OK) settle(route); /* a tile that failed half way may have left copies queued */
  if (rc == AAD_APIRESULT_OK && output_size) memcpy(output_size, sizes.data(), sizeof(uint64_t) * num_streams);
  return rc;
}

/*
 * Decode num_streams host images of one format.  fill(i, offset, size, dst) writes bytes
 * [offset, offset + size) of stream i's image, drain(i, frame0, src, frames) receives that range
 * of its interleaved int16 frames.  Tiles carry bare blocks (the file header stays on the host).
 */
template <class Fill, class Drain>
AADApiResult decode_host(AADHipContext *ctx, const struct AADHeaderInfo *format, int32_t has_file_header,
                         uint32_t num_streams, const uint64_t *data_size, const uint32_t *num_samples,
                         uint32_t *decoded_frames, Fill fill, Drain drain)
{
  const uint32_t ch = format->num_channels, head = has_file_header ? AAD_HEADER_SIZE : 0;
  const uint32_t spb = format->num_samples_per_block, bs = format->block_size;
  if (ch == 0 || bs == 0 || spb == 0) return AAD_APIRESULT_INVALID_FORMAT;
  std::vector<AADHipStreamDesc> table(num_streams);
  std::vector<uint64_t> prefix((size_t)num_streams + 1), blocks(num_streams), fill_cost, tile_blocks;
  uint64_t total = 0;
  for (uint32_t i = 0; i < num_streams; i++) {
    table[i] = {0, 0, data_size[i], num_samples[i], 0};
    total += data_size[i] + (uint64_t)num_samples[i] * ch * sizeof(int16_t);
  }
  {
    /* the whole batch is validated before the first tile runs: a bad stream fails the call with
     * nothing decoded, as it did when a batch was one launch */
    aad::DecodeArgs whole;
    const AADApiResult ok = decode_plan_init(format, has_file_header, num_streams, table.data(), prefix.data(), &whole);
    if (ok != AAD_APIRESULT_OK) return ok;
    for (uint32_t i = 0; i < num_streams; i++) blocks[i] = prefix[i + 1] - prefix[i];
  }
  DeviceGuard guard(ctx);
  if (!guard.ok || !ensure_events(ctx)) return AAD_APIRESULT_NG;
  if (decoded_frames) memset(decoded_frames, 0, sizeof(uint32_t) * num_streams);

  /* bytes a full block's decode touches beyond its own block_size (0 for every geometry an encoder writes) */
  const uint64_t unit_samples = format->bits_per_sample == 3 ? 8 : (format->bits_per_sample == 4 ? 2 : 4);
  const uint64_t unit_bytes = (uint64_t)(format->bits_per_sample == 3 ? 3 : 1) * ch;
  const uint64_t touched = (uint64_t)AAD_BLOCK_HEADER_BYTES_PER_CH * ch + (spb > 4 ? (spb - 4 + unit_samples - 1) / unit_samples * unit_bytes : 0);
  const uint64_t overreach = touched > bs ? touched - bs : 0;
  TilePlanner planner(blocks.data(), num_streams, (uint64_t)spb * ch * sizeof(int16_t) + bs + overreach, tile_budget(ctx, total));
  const bool piped = batch_is_cut(ctx, total);
  if (piped && !ensure_pipeline(ctx)) return AAD_APIRESULT_NG;
  const Route route = route_for(ctx, piped);
  Flight flight[2];
  uint64_t sequence = 0;

  auto finish = [&](int b) -> bool {
    Flight &f = flight[b];
    if (!f.active) return true;
    f.active = false;
    const int16_t *out = static_cast<const int16_t *>(ctx->out[b].host);
    bool ok = true;
    for (uint32_t p = 0, first = 0; p < f.pieces; first = f.piece_end[p++]) {
      /* every piece is waited for, also after a failure: the buffers must be idle on return */
      if (!hip_ok(ctx, hipEventSynchronize(p + 1 == f.pieces ? ctx->chunk_done[b] : ctx->piece_done[p]), "hipEventSynchronize")) ok = false;
      if (!ok) continue;
      staged_span(ctx, first, f.piece_end[p], f.cost, [&](uint32_t lo, uint32_t hi) {
        for (uint32_t k = lo; k < hi; k++) {
          const Delivery &d = f.items[k];
          if (d.count) drain(d.stream, (uint32_t)d.dst, out + d.src, (uint32_t)d.count);
          if (decoded_frames) decoded_frames[d.stream] += (uint32_t)d.count; /* one item per stream and tile */
        }
      });
    }
    return ok;
  };

  AADApiResult rc = AAD_APIRESULT_OK;
  TileStep step;
  while (rc == AAD_APIRESULT_OK && planner.next(&step)) {
    const int b = (int)(sequence & 1);
    if (!finish(b)) { rc = AAD_APIRESULT_NG; break; }
    const uint32_t n = step.alive;
    const std::vector<uint32_t> &order = planner.order;
    Flight &f = flight[b];
    table.resize(n);
    prefix.resize((size_t)n + 1);
    tile_blocks.resize(n);
    f.items.resize(n);
    f.cost.resize((size_t)n + 1);
    fill_cost.resize((size_t)n + 1);
    uint64_t pcm_elems = 0, data_bytes = 0;
    for (uint32_t k = 0; k < n; k++) {
      const uint32_t i = order[k];
      const uint64_t payload = data_size[i] - head; /* alive: it has a block, so more than `head` bytes */
      /* a block whose header asks for more samples than block_size holds reads on into the bytes behind it, as the reference's
       * unbounded code walk does (src/aad_decoder.c:396-451; the header checks relate samples_per_block and block_size to
       * nothing, :173-225): a tile carries that reach behind its last block, and a stream's last tile every byte that is left */
      const uint64_t byte0 = step.block0 * bs;
      const uint64_t upto = step.block1 >= blocks[i] ? payload : step.block1 * bs + overreach;
      const uint64_t byte1 = upto < payload ? upto : payload;
      const uint64_t frame0 = step.block0 * spb, frame1 = step.block1 * spb < num_samples[i] ? step.block1 * spb : num_samples[i];
      /* frames the reference's block walk produces: it stops when the bytes run out (src/aad_decoder.c:514) */
      tile_blocks[k] = (step.block1 < blocks[i] ? step.block1 : blocks[i]) - step.block0;
      const uint64_t by_bytes = tile_blocks[k] * spb;
      table[k].pcm_offset = pcm_elems;
      table[k].data_offset = data_bytes;
      table[k].data_size = byte1 - byte0;
      table[k].num_samples = (uint32_t)(frame1 - frame0);
      table[k].reserved = 0;
      f.items[k] = {i, pcm_elems, frame0, by_bytes < frame1 - frame0 ? by_bytes : frame1 - frame0};
      f.cost[k] = pcm_elems * sizeof(int16_t);
      fill_cost[k] = data_bytes;
      pcm_elems += round_up((frame1 - frame0) * ch, 8);
      data_bytes += round_up(byte1 - byte0, 16);
    }
    f.cost[n] = pcm_elems * sizeof(int16_t);
    fill_cost[n] = data_bytes;
    aad::DecodeArgs a;
    rc = decode_plan_init(format, 0, n, table.data(), prefix.data(), &a, tile_blocks.data());
    if (rc != AAD_APIRESULT_OK) break;
    const size_t table_bytes = round_up(sizeof(AADHipStreamDesc) * (size_t)n, 64);
    const size_t prefix_bytes = round_up(sizeof(uint64_t) * ((size_t)n + 1), 64);
    const size_t data_off = table_bytes + prefix_bytes;
    const size_t in_bytes = data_off + data_bytes, out_bytes = pcm_elems * sizeof(int16_t);
    rc = AAD_APIRESULT_NG;
    if (!staging_reserve(ctx, ctx->in[b], in_bytes + 64) || !staging_reserve(ctx, ctx->out[b], out_bytes + 64)) break;
    uint8_t *hin = static_cast<uint8_t *>(ctx->in[b].host), *din = static_cast<uint8_t *>(ctx->in[b].dev);
    memcpy(hin, table.data(), sizeof(AADHipStreamDesc) * (size_t)n);
    memcpy(hin + table_bytes, prefix.data(), sizeof(uint64_t) * ((size_t)n + 1));
    staged_ranges(ctx, n, fill_cost, [&](uint32_t lo, uint32_t hi) {
      for (uint32_t k = lo; k < hi; k++)
        fill(order[k], head + step.block0 * bs, table[k].data_size, hin + data_off + table[k].data_offset);
    });
    if (!hip_ok(ctx, hipMemcpyAsync(din, hin, in_bytes, hipMemcpyHostToDevice, route.up), "H2D block")) break;
    if (!hop(ctx, ctx->uploaded[b], route.up, route.run)) break;
    a.streams = reinterpret_cast<const aad::StreamDesc *>(din);
    a.block_prefix = reinterpret_cast<const uint64_t *>(din + table_bytes);
    a.data = din + data_off;
    a.pcm = static_cast<int16_t *>(ctx->out[b].dev);
    if ((reinterpret_cast<uintptr_t>(a.pcm) & 63u) != 0) a.stream_stores = 0;
    if (run_decode(ctx, a) != AAD_APIRESULT_OK) break;
    if (!hop(ctx, ctx->computed[b], route.run, route.down)) break;
    {
      f.pieces = cut_pieces(f.cost, n, !piped, f.piece_end);
      /* piece_done[] is ONE set of events for both flights: only a batch that travels as a single tile (nothing in
       * the other flight) may cut its copy into pieces */
      if (piped && f.pieces != 1) {
        snprintf(ctx->last_error, sizeof(ctx->last_error), "internal: a piped decode tile was cut into %u copy pieces", f.pieces);
        rc = AAD_APIRESULT_NG;
        break;
      }
      size_t got = 0;
      bool ok = true;
      for (uint32_t p = 0; ok && p < f.pieces; p++) {
        const size_t upto = p + 1 == f.pieces ? out_bytes : (size_t)f.cost[f.piece_end[p]];
        if (upto > got)
          ok = hip_ok(ctx, hipMemcpyAsync(static_cast<uint8_t *>(ctx->out[b].host) + got, static_cast<uint8_t *>(ctx->out[b].dev) + got,
                                         upto - got, hipMemcpyDeviceToHost, route.down), "D2H block");
        got = upto;
        ok = ok && hip_ok(ctx, hipEventRecord(p + 1 == f.pieces ? ctx->chunk_done[b] : ctx->piece_done[p], route.down), "hipEventRecord");
      }
      if (!ok) break;
    }
    f.sequence = sequence++;
    f.active = true;
    rc = AAD_APIRESULT_OK;
  }
  const int older = flight[0].active && flight[1].active && flight[1].sequence < flight[0].sequence ? 1 : 0;
  if (!finish(older) && rc == AAD_APIRESULT_OK) rc = AAD_APIRESULT_NG;
  if (!finish(older ^ 1) && rc == AAD_APIRESULT_OK) rc = AAD_APIRESULT_NG;
  if (rc != AAD_APIRESULT_OK) settle(route);
  return rc;
}

} /* namespace */

extern "C" {

AADApiResult AADHip_EncodeBatch(struct AADHipContext *ctx, const struct AADEncodeParameter *parameter,
                                uint32_t num_streams, const int16_t *const *pcm, const uint32_t *num_samples,
                                uint8_t *const *data, const uint64_t *data_capacity, uint64_t *output_size,
                                struct AADHipLaneState *state)
{
  if (ctx == nullptr || parameter == nullptr || (num_streams != 0 && (pcm == nullptr || num_samples == nullptr ||
      data == nullptr || data_capacity == nullptr)))
    return AAD_APIRESULT_INVALID_ARGUMENT;
  if (num_streams == 0) return AAD_APIRESULT_OK;
  for (uint32_t i = 0; i < num_streams; i++)
    if (pcm[i] == nullptr || data[i] == nullptr) return AAD_APIRESULT_INVALID_ARGUMENT;
  const size_t frame_bytes = sizeof(int16_t) * parameter->num_channels;
  return encode_host(ctx, parameter, num_streams, num_samples, data_capacity, output_size, state,
                     [&](uint32_t i, uint32_t frame0, uint32_t frames, int16_t *dst) {
                       memcpy(dst, pcm[i] + (size_t)frame0 * parameter->num_channels, (size_t)frames * frame_bytes);
                     },
                     [&](uint32_t i, uint64_t offset, const uint8_t *src, uint64_t size) { memcpy(data[i] + offset, src, size); });
}

/* AADEncoder_EncodeWhole's data path (src/aad_encoder.c:814-891): planar int32 rows in, one image
 * out.  The planar -> interleaved int16 conversion writes straight into the pinned block; samples
 * must already be in int16 range, which the reference only asserts (src/aad_encoder.c:612) -
 * out-of-range input saturates. */
AADApiResult AADHipInternal_EncodePlanar32(struct AADHipContext *ctx, const struct AADEncodeParameter *parameter,
                                           const int32_t *const *input, uint32_t num_samples, uint8_t *data,
                                           uint64_t data_capacity, uint64_t *output_size, struct AADHipLaneState *state)
{
  const uint32_t ch = parameter->num_channels;
  return encode_host(ctx, parameter, 1, &num_samples, &data_capacity, output_size, state,
                     [&](uint32_t, uint32_t frame0, uint32_t frames, int16_t *dst) {
                       for (uint32_t c = 0; c < ch; c++) {
                         const int32_t *x = input[c] + frame0;
                         int16_t *d = dst + c;
                         for (uint32_t s = 0; s < frames; s++, d += ch) {
                           const int32_t v = x[s];
                           *d = (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
                         }
                       }
                     },
                     [&](uint32_t, uint64_t offset, const uint8_t *src, uint64_t size) { memcpy(data + offset, src, size); });
}

/* shared by AADHip_DecodeBatch (file images) and the legacy AADDecoder_DecodeBlock (bare block) */
AADApiResult AADHipInternal_DecodeHost(struct AADHipContext *ctx, const struct AADHeaderInfo *format,
                                       int32_t has_file_header, uint32_t num_streams,
                                       const uint8_t *const *data, const uint64_t *data_size,
                                       const uint32_t *num_samples, int16_t *const *pcm, uint32_t *decoded_frames)
{
  const size_t frame_bytes = sizeof(int16_t) * format->num_channels;
  return decode_host(ctx, format, has_file_header, num_streams, data_size, num_samples, decoded_frames,
                     [&](uint32_t i, uint64_t offset, uint64_t size, uint8_t *dst) { memcpy(dst, data[i] + offset, size); },
                     [&](uint32_t i, uint32_t frame0, const int16_t *src, uint32_t frames) {
                       memcpy(pcm[i] + (size_t)frame0 * format->num_channels, src, (size_t)frames * frame_bytes);
                     });
}

/* AADDecoder_DecodeWhole / DecodeBlock's data path: one image (or bare block) in, planar int32
 * rows out, widened straight from the pinned block (src/aad_decoder.c:478-538, :321-475) */
AADApiResult AADHipInternal_DecodePlanar32(struct AADHipContext *ctx, const struct AADHeaderInfo *format,
                                           int32_t has_file_header, const uint8_t *data, uint64_t data_size,
                                           uint32_t want_frames, int32_t *const *buffer, uint32_t *decoded_frames)
{
  const uint32_t ch = format->num_channels;
  return decode_host(ctx, format, has_file_header, 1, &data_size, &want_frames, decoded_frames,
                     [&](uint32_t, uint64_t offset, uint64_t size, uint8_t *dst) { memcpy(dst, data + offset, size); },
                     [&](uint32_t, uint32_t frame0, const int16_t *src, uint32_t frames) {
                       for (uint32_t c = 0; c < ch; c++) {
                         int32_t *y = buffer[c] + frame0;
                         const int16_t *s = src + c;
                         for (uint32_t k = 0; k < frames; k++, s += ch) y[k] = *s;
                       }
                     });
}

AADApiResult AADHip_DecodeBatch(struct AADHipContext *ctx, uint32_t num_streams, const uint8_t *const *data,
                                const uint64_t *data_size, int16_t *const *pcm, const uint32_t *pcm_capacity_frames,
                                uint32_t *decoded_frames)
{
  if (ctx == nullptr || (num_streams != 0 && (data == nullptr || data_size == nullptr || pcm == nullptr ||
      pcm_capacity_frames == nullptr)))
    return AAD_APIRESULT_INVALID_ARGUMENT;
  if (num_streams == 0) return AAD_APIRESULT_OK;
  std::vector<uint32_t> frames(num_streams);
  AADHeaderInfo format;
  for (uint32_t i = 0; i < num_streams; i++) {
    AADHeaderInfo h;
    if (data[i] == nullptr || pcm[i] == nullptr) return AAD_APIRESULT_INVALID_ARGUMENT;
    if (data_size[i] < AAD_HEADER_SIZE) return AAD_APIRESULT_INSUFFICIENT_DATA;
    if (!AADFormat_GetHeader(data[i], &h)) return AAD_APIRESULT_INVALID_FORMAT;
    if (!AADFormat_HeaderAcceptedByDecoder(&h, AAD_HIP_MAX_NUM_CHANNELS)) return AAD_APIRESULT_INVALID_FORMAT;
    if (i == 0) {
      format = h;
    } else if (h.num_channels != format.num_channels || h.bits_per_sample != format.bits_per_sample ||
               h.block_size != format.block_size || h.num_samples_per_block != format.num_samples_per_block ||
               h.ch_process_method != format.ch_process_method) {
      return AAD_APIRESULT_INVALID_FORMAT; /* one format per batch */
    }
    if (pcm_capacity_frames[i] < h.num_samples) return AAD_APIRESULT_INSUFFICIENT_BUFFER;
    frames[i] = h.num_samples;
  }
  return AADHipInternal_DecodeHost(ctx, &format, 1, num_streams, data, data_size, frames.data(), pcm, decoded_frames);
}

/* ------------------------------------------------------------------- reconstruction modes -- */

AADApiResult AADHip_ReconstructPlanCreate(struct AADHipContext *ctx, const struct AADEncodeParameter *parameter,
                                          uint32_t num_streams, const struct AADHipStreamDesc *streams,
                                          struct AADHipReconstructPlan **plan)
{
  if (ctx == nullptr || parameter == nullptr || plan == nullptr || (num_streams != 0 && streams == nullptr))
    return AAD_APIRESULT_INVALID_ARGUMENT;
  *plan = nullptr;
  AADHipReconstructPlan *p = new (std::nothrow) AADHipReconstructPlan();
  if (p == nullptr) return AAD_APIRESULT_NG;
  memset(static_cast<void *>(p), 0, sizeof(*p));
  p->ctx = ctx;
  AADApiResult rc = AADHip_EncodePlanCreate(ctx, parameter, num_streams, streams, &p->encode);
  if (rc == AAD_APIRESULT_OK) {
    /* the decoder sees exactly the images the encoder writes */
    AADHeaderInfo h;
    std::vector<AADHipStreamDesc> images(streams, streams + num_streams);
    std::vector<uint64_t> prefix((size_t)num_streams + 1);
    uint64_t segments = 0;
    (void)AADFormat_ParameterToHeader(parameter, 1, AAD_HIP_MAX_NUM_CHANNELS, &h);
    for (uint32_t i = 0; i < num_streams; i++) {
      h.num_samples = streams[i].num_samples;
      images[i].data_size = AADFormat_EncodedSize(&h);
      prefix[i] = segments;
      segments += ((uint64_t)streams[i].num_samples * h.num_channels + aad::kCompareSegment - 1) / aad::kCompareSegment;
    }
    prefix[num_streams] = segments;
    rc = AADHip_DecodePlanCreate(ctx, &h, 1, num_streams, images.data(), &p->decode);
    if (rc == AAD_APIRESULT_OK) {
      DeviceGuard guard(ctx);
      if (!guard.ok || !upload(ctx, &p->d_segment_prefix, prefix.data(), prefix.size()) ||
          !hip_ok(ctx, hipMalloc((void **)&p->d_partials, sizeof(aad::ErrorPartial) * (segments ? segments : 1)), "hipMalloc partials"))
        rc = AAD_APIRESULT_NG;
    }
    p->args.streams = p->decode ? p->decode->d_streams : nullptr;
    p->args.segment_prefix = p->d_segment_prefix;
    p->args.total_segments = segments;
    p->args.num_streams = num_streams;
    p->args.channels = h.num_channels;
  }
  if (rc != AAD_APIRESULT_OK) {
    AADHip_ReconstructPlanDestroy(p);
    return rc;
  }
  *plan = p;
  return AAD_APIRESULT_OK;
}

void AADHip_ReconstructPlanDestroy(struct AADHipReconstructPlan *plan)
{
  if (plan == nullptr) return;
  AADHip_EncodePlanDestroy(plan->encode); /* each synchronises the stream first */
  AADHip_DecodePlanDestroy(plan->decode);
  {
    DeviceGuard guard(plan->ctx);
    if (guard.ok) {
      (void)hipStreamSynchronize(plan->ctx->stream);
      if (plan->d_segment_prefix) (void)hipFree(plan->d_segment_prefix);
      if (plan->d_partials) (void)hipFree(plan->d_partials);
    }
  }
  delete plan;
}

AADApiResult AADHip_ReconstructPlanRun(struct AADHipReconstructPlan *plan, const int16_t *device_pcm, uint8_t *device_data,
                                       int16_t *device_out, int32_t output_kind, struct AADHipErrorStats *device_stats)
{
  if (plan == nullptr || device_pcm == nullptr || device_data == nullptr || device_out == nullptr ||
      device_out == device_pcm)
    return AAD_APIRESULT_INVALID_ARGUMENT;
  if (output_kind != AAD_HIP_RECONSTRUCT_DECODED && output_kind != AAD_HIP_RECONSTRUCT_RESIDUAL)
    return AAD_APIRESULT_INVALID_ARGUMENT;
  AADHipContext *ctx = plan->ctx;
  /* a reconstruction is several kernels: pending AADHip_ContextSignalNextRun events are recorded in front of the first and
   * behind the last of them */
  const aad::LaunchSignal signal = take_signal(ctx);
  if (signal.start != nullptr && !hip_ok(ctx, hipEventRecord(signal.start, ctx->stream), "hipEventRecord")) return AAD_APIRESULT_NG;
  auto done = [&](AADApiResult r) -> AADApiResult {
    if (r == AAD_APIRESULT_OK && signal.stop != nullptr && !hip_ok(ctx, hipEventRecord(signal.stop, ctx->stream), "hipEventRecord")) return AAD_APIRESULT_NG;
    return r;
  };
  if (plan->args.num_streams == 0) return done(AAD_APIRESULT_OK);
  AADApiResult rc = AADHip_EncodePlanRun(plan->encode, device_pcm, device_data, nullptr);
  if (rc != AAD_APIRESULT_OK) return rc;
  rc = AADHip_DecodePlanRun(plan->decode, device_data, device_out);
  if (rc != AAD_APIRESULT_OK) return rc;
  if (device_stats == nullptr && output_kind == AAD_HIP_RECONSTRUCT_DECODED) return done(AAD_APIRESULT_OK);
  DeviceGuard guard(ctx);
  if (!guard.ok) return AAD_APIRESULT_NG;
  aad::CompareArgs a = plan->args;
  a.original = device_pcm;
  a.decoded = device_out;
  a.partials = device_stats ? plan->d_partials : nullptr;
  a.stats = reinterpret_cast<aad::ErrorStatsRecord *>(device_stats);
  a.write_residual = output_kind == AAD_HIP_RECONSTRUCT_RESIDUAL;
  a.sequential = ctx->compare_sequential ? 1u : 0u;
  if (a.total_segments > 0x7FFFFFFFull) return AAD_APIRESULT_INVALID_ARGUMENT;
  hipLaunchKernelGGL(aad::compare_segments_kernel, dim3((unsigned)a.total_segments), dim3(aad::kCompareThreads), 0, ctx->stream, a);
  if (device_stats)
    hipLaunchKernelGGL(aad::compare_finish_kernel, dim3(a.num_streams), dim3(64), 0, ctx->stream, a);
  return done(hip_ok(ctx, hipGetLastError(), "compare launch") ? AAD_APIRESULT_OK : AAD_APIRESULT_NG);
}

/*
 * Host-memory reconstruction.  The COMPUTE stays whole - every stream of a wave is encoded, decoded and compared by one
 * AADHip_ReconstructPlanRun over device-resident buffers, so that the encoder's block chains of all streams run side by side and
 * the statistics are summed per stream in the reference's order whatever the batch size - and only the STAGING is cut: the PCM goes
 * up and the output comes down through the context's two pinned blocks in chunks (the tile budget of the other host-memory entry
 * points: 16 MiB, or AAD_HIP_OPTION_TILE_KBYTES), chunk k + 1 being filled / chunk k - 1 being scattered by the host while chunk
 * k is on the bus.  Pinned memory is bounded by the chunk size; device memory holds 2 x PCM + images of a WAVE of whole streams,
 * and a batch that does not fit the device's free memory (288 GB on MI355X: ~60 G channel-samples) goes as several waves of
 * consecutive streams.  A forced tile size also forces small waves (64 tiles' worth), so that the tests walk every path.
 */
namespace {

struct ReconstructWave {
  uint32_t first, count;
};

/* device bytes a stream occupies in a wave: PCM in + PCM out (rows padded to 16 bytes), its image (padded to 16), its statistics */
uint64_t reconstruct_footprint(uint64_t samples, uint32_t ch, uint64_t image)
{
  return 2 * round_up(samples * ch, 8) * sizeof(int16_t) + round_up(image, 16) + sizeof(AADHipErrorStats);
}

bool device_block_reserve(AADHipContext *ctx, void **block, size_t *capacity, size_t bytes, const char *what)
{
  if (*capacity >= bytes) return true;
  if (*block) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(*block);
  }
  *block = nullptr;
  *capacity = 0;
  const size_t want = bytes + bytes / 8 + 4096; /* some slack, so that a slightly larger next wave does not reallocate */
  if (hipMalloc(block, want) == hipSuccess) {
    *capacity = want;
    return true;
  }
  (void)hipGetLastError();
  *block = nullptr;
  if (!hip_ok(ctx, hipMalloc(block, bytes), what)) {
    *block = nullptr;
    return false;
  }
  *capacity = bytes;
  return true;
}

/* one wave: streams [first, first + count) of the batch, device-resident compute, chunked staging */
AADApiResult reconstruct_wave(AADHipContext *ctx, const struct AADEncodeParameter *parameter, uint32_t count,
                              const int16_t *const *pcm, const uint32_t *num_samples, int32_t output_kind,
                              int16_t *const *out_pcm, struct AADHipErrorStats *stats, uint64_t chunk_bytes)
{
  const uint32_t ch = parameter->num_channels;
  std::vector<AADHipStreamDesc> table(count);
  std::vector<uint64_t> byte_prefix((size_t)count + 1); /* start of every stream's row in the flat PCM block, in bytes */
  uint64_t pcm_elems = 0, data_bytes = 0;
  for (uint32_t i = 0; i < count; i++) {
    const uint64_t size = AADHip_CalculateEncodedSize(parameter, num_samples[i]);
    table[i].pcm_offset = pcm_elems;
    table[i].data_offset = data_bytes;
    table[i].data_size = size;
    table[i].num_samples = num_samples[i];
    table[i].reserved = 0;
    byte_prefix[i] = pcm_elems * sizeof(int16_t);
    pcm_elems += round_up((uint64_t)num_samples[i] * ch, 8);
    data_bytes += round_up(size, 16);
  }
  byte_prefix[count] = pcm_elems * sizeof(int16_t);
  AADHipReconstructPlan *plan = nullptr;
  AADApiResult rc = AADHip_ReconstructPlanCreate(ctx, parameter, count, table.data(), &plan);
  if (rc != AAD_APIRESULT_OK) return rc;
  DeviceGuard guard(ctx);
  /* device: [pcm in] ; [pcm out | statistics] ; [images].  The statistics sit behind the output PCM so that one flat range
   * comes down. */
  const size_t pcm_bytes = pcm_elems * sizeof(int16_t), stats_bytes = stats ? sizeof(AADHipErrorStats) * (size_t)count : 0;
  const size_t stats_off = round_up(pcm_bytes, 64), out_bytes = stats_off + stats_bytes;
  const size_t chunk = (size_t)round_up(chunk_bytes < 4096 ? 4096 : chunk_bytes, 64);
  rc = AAD_APIRESULT_NG;
  do {
    if (!guard.ok || !ensure_events(ctx)) break;
    if (!device_block_reserve(ctx, &ctx->d_rc_in, &ctx->rc_in_capacity, pcm_bytes + 64, "hipMalloc reconstruction input") ||
        !device_block_reserve(ctx, &ctx->d_rc_out, &ctx->rc_out_capacity, out_bytes + 64, "hipMalloc reconstruction output") ||
        !device_block_reserve(ctx, &ctx->d_scratch, &ctx->scratch_capacity, data_bytes + 64, "hipMalloc image scratch"))
      break;
    const size_t stage = pcm_bytes < chunk ? pcm_bytes + 64 : chunk;
    if (!staging_reserve(ctx, ctx->in[0], stage) || (pcm_bytes > chunk && !staging_reserve(ctx, ctx->in[1], stage))) break;
    uint8_t *d_in = static_cast<uint8_t *>(ctx->d_rc_in), *d_out = static_cast<uint8_t *>(ctx->d_rc_out);

    /* the rows of streams [a, b) that fall into bytes [lo, hi) of the flat PCM block <-> host block `base` (which holds byte lo at 0) */
    auto rows = [&](uint32_t a, uint32_t b, uint64_t lo, uint64_t hi, uint8_t *base, bool up) {
      for (uint32_t i = a; i < b; i++) {
        const uint64_t r0 = byte_prefix[i], r1 = r0 + (uint64_t)num_samples[i] * ch * sizeof(int16_t);
        const uint64_t c0 = r0 > lo ? r0 : lo, c1 = r1 < hi ? r1 : hi;
        if (c1 <= c0) continue;
        if (up) memcpy(base + (c0 - lo), reinterpret_cast<const uint8_t *>(pcm[i]) + (c0 - r0), (size_t)(c1 - c0));
        else memcpy(reinterpret_cast<uint8_t *>(out_pcm[i]) + (c0 - r0), base + (c0 - lo), (size_t)(c1 - c0));
      }
    };
    auto span = [&](uint64_t lo, uint64_t hi, uint32_t *a, uint32_t *b) { /* streams whose rows meet [lo, hi) */
      *a = (uint32_t)(std::upper_bound(byte_prefix.begin(), byte_prefix.end(), lo) - byte_prefix.begin());
      *a = *a ? *a - 1 : 0;
      *b = (uint32_t)(std::lower_bound(byte_prefix.begin(), byte_prefix.end(), hi) - byte_prefix.begin());
      if (*b > count) *b = count;
    };

    /* ---- up: chunk k is filled while chunk k - 1 is on the bus ---- */
    bool ok = true;
    uint32_t k = 0;
    for (uint64_t lo = 0; lo < pcm_bytes && ok; lo += chunk, k++) {
      const uint64_t hi = lo + chunk < pcm_bytes ? lo + chunk : pcm_bytes;
      Staging &st = ctx->in[k & 1];
      if (k >= 2) ok = hip_ok(ctx, hipEventSynchronize(ctx->chunk_done[k & 1]), "hipEventSynchronize"); /* the block's last copy has left it */
      if (!ok) break;
      uint32_t a, b;
      span(lo, hi, &a, &b);
      uint8_t *host = static_cast<uint8_t *>(st.host);
      staged_span(ctx, a, b, byte_prefix, [&](uint32_t x, uint32_t y) { rows(x, y, lo, hi, host, true); });
      ok = hip_ok(ctx, hipMemcpyAsync(d_in + lo, host, (size_t)(hi - lo), hipMemcpyHostToDevice, ctx->stream), "H2D pcm") &&
           hip_ok(ctx, hipEventRecord(ctx->chunk_done[k & 1], ctx->stream), "hipEventRecord");
    }
    if (!ok) break;

    /* ---- compute: one run over the whole wave ---- */
    rc = AADHip_ReconstructPlanRun(plan, reinterpret_cast<const int16_t *>(d_in), static_cast<uint8_t *>(ctx->d_scratch),
                                   reinterpret_cast<int16_t *>(d_out), output_kind,
                                   stats ? reinterpret_cast<AADHipErrorStats *>(d_out + stats_off) : nullptr);
    if (rc != AAD_APIRESULT_OK) break;
    rc = AAD_APIRESULT_NG;

    /* ---- down: [PCM (if wanted) | statistics] as one flat range; chunk k - 1 is scattered while chunk k is on the bus.
     * Statistics only: nothing but 24 bytes per stream comes back. ---- */
    const uint64_t down_lo = out_pcm ? 0 : stats_off, down_hi = stats ? out_bytes : (out_pcm ? pcm_bytes : down_lo);
    if (down_hi > down_lo) {
      const size_t dstage = (size_t)(down_hi - down_lo) < chunk ? (size_t)(down_hi - down_lo) + 64 : chunk;
      if (!staging_reserve(ctx, ctx->out[0], dstage) || (down_hi - down_lo > chunk && !staging_reserve(ctx, ctx->out[1], dstage))) break;
      auto scatter = [&](uint32_t kk, uint64_t lo, uint64_t hi) { /* host block kk & 1 holds bytes [lo, hi) of the output block */
        uint8_t *host = static_cast<uint8_t *>(ctx->out[kk & 1].host);
        if (out_pcm && lo < pcm_bytes) {
          const uint64_t ph = hi < pcm_bytes ? hi : pcm_bytes;
          uint32_t a, b;
          span(lo, ph, &a, &b);
          staged_span(ctx, a, b, byte_prefix, [&](uint32_t x, uint32_t y) { rows(x, y, lo, ph, host, false); });
        }
        if (stats && hi > stats_off) {
          const uint64_t s0 = lo > stats_off ? lo : stats_off;
          memcpy(reinterpret_cast<uint8_t *>(stats) + (s0 - stats_off), host + (s0 - lo), (size_t)(hi - s0));
        }
      };
      uint64_t prev_lo = 0, prev_hi = 0;
      bool have_prev = false;
      k = 0;
      for (uint64_t lo = down_lo; lo < down_hi && ok; lo += chunk, k++) {
        const uint64_t hi = lo + chunk < down_hi ? lo + chunk : down_hi;
        ok = hip_ok(ctx, hipMemcpyAsync(ctx->out[k & 1].host, d_out + lo, (size_t)(hi - lo), hipMemcpyDeviceToHost, ctx->stream), "D2H block") &&
             hip_ok(ctx, hipEventRecord(ctx->chunk_done[k & 1], ctx->stream), "hipEventRecord");
        if (ok && have_prev) { /* the other block: its copy was queued one round ago */
          ok = hip_ok(ctx, hipEventSynchronize(ctx->chunk_done[(k - 1) & 1]), "hipEventSynchronize");
          if (ok) scatter(k - 1, prev_lo, prev_hi);
        }
        prev_lo = lo, prev_hi = hi, have_prev = true;
      }
      if (ok && have_prev) {
        ok = hip_ok(ctx, hipEventSynchronize(ctx->chunk_done[(k - 1) & 1]), "hipEventSynchronize");
        if (ok) scatter(k - 1, prev_lo, prev_hi);
      }
      if (!ok) break;
    }
    if (!hip_ok(ctx, hipStreamSynchronize(ctx->stream), "sync")) break;
    rc = AAD_APIRESULT_OK;
  } while (0);
  if (rc != AAD_APIRESULT_OK) (void)hipStreamSynchronize(ctx->stream); /* nothing of this call stays in flight */
  AADHip_ReconstructPlanDestroy(plan);
  return rc;
}

} /* namespace */

AADApiResult AADHip_ReconstructBatch(struct AADHipContext *ctx, const struct AADEncodeParameter *parameter,
                                     uint32_t num_streams, const int16_t *const *pcm, const uint32_t *num_samples,
                                     int32_t output_kind, int16_t *const *out_pcm, struct AADHipErrorStats *stats)
{
  if (ctx == nullptr || parameter == nullptr || (num_streams != 0 && (pcm == nullptr || num_samples == nullptr)))
    return AAD_APIRESULT_INVALID_ARGUMENT;
  if (num_streams == 0) return AAD_APIRESULT_OK;
  const uint32_t ch = parameter->num_channels;
  std::vector<uint64_t> footprint(num_streams);
  for (uint32_t i = 0; i < num_streams; i++) { /* the whole batch is validated before the first wave runs */
    if (pcm[i] == nullptr || (out_pcm != nullptr && out_pcm[i] == nullptr)) return AAD_APIRESULT_INVALID_ARGUMENT;
    const uint64_t size = AADHip_CalculateEncodedSize(parameter, num_samples[i]);
    if (size == 0) return AAD_APIRESULT_INVALID_FORMAT;
    footprint[i] = reconstruct_footprint(num_samples[i], ch, size);
  }
  const uint64_t chunk = ctx->tile_bytes > 0 ? (uint64_t)ctx->tile_bytes : kChunkBudget;
  uint64_t budget;
  if (ctx->tile_bytes > 0) {
    budget = (uint64_t)ctx->tile_bytes * 64u; /* a forced tile size: small waves too (tests) */
  } else {
    DeviceGuard guard(ctx);
    size_t free_bytes = 0, total_bytes = 0;
    if (!guard.ok || !hip_ok(ctx, hipMemGetInfo(&free_bytes, &total_bytes), "hipMemGetInfo")) return AAD_APIRESULT_NG;
    /* what is free now plus what this context's own grow-only blocks already hold, less a quarter for everybody else */
    const uint64_t own = ctx->rc_in_capacity + ctx->rc_out_capacity + ctx->scratch_capacity;
    budget = ((uint64_t)free_bytes + own) / 4 * 3;
  }
  AADApiResult rc = AAD_APIRESULT_OK;
  for (uint32_t first = 0; first < num_streams && rc == AAD_APIRESULT_OK;) {
    uint32_t n = 0;
    uint64_t sum = 0;
    while (first + n < num_streams && (n == 0 || sum + footprint[first + n] <= budget)) sum += footprint[first + n++]; /* a stream alone is always tried */
    rc = reconstruct_wave(ctx, parameter, n, pcm + first, num_samples + first, output_kind, out_pcm ? out_pcm + first : nullptr,
                          stats ? stats + first : nullptr, chunk);
    first += n;
  }
  return rc;
}

#if AAD_PHASE_TIMING
/* measurement builds only: copy out and reset the phase log of the encode kernel */
uint32_t AADHipDebug_ReadPhaseTimes(uint64_t *out, uint32_t capacity)
{
  uint32_t n = 0, zero = 0;
  uint64_t host[512];
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(aad::g_phase_count), sizeof(n));
  (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(aad::g_phase_times), sizeof(host));
  (void)hipMemcpyToSymbol(HIP_SYMBOL(aad::g_phase_count), &zero, sizeof(zero));
  if (n > capacity) n = capacity;
  memcpy(out, host, sizeof(uint64_t) * n);
  return n;
}
#endif

} /* extern "C" */
