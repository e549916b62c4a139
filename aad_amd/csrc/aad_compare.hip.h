/*
 * aad_compare.hip.h - device side of the reconstruction modes:
 * residual output and the RMSE / MSD / MaxAE figures of the reference CLI's -r / -g / -c
 * (src/main.c:275-503), computed where encode -> decode left the two PCM buffers: in HBM.
 *
 * The arithmetic per value is the CLI's, literally (src/main.c:470-491): with x the original and
 * y the reconstructed int16,
 *     g  = (x << 16) - (y << 16)            32-bit wrap, the WAV reader's domain
 *     e  = (double)g / INT32_MAX - (double)y / INT32_MAX
 *     RMSE = sqrt(sum e^2 / N), MSD = sum |e| / N, MaxAE = max |e|
 * (that the second operand is y and not y << 16 is how the reference is written; a drop-in
 * prints what it prints).  fp64 division on gfx950 is IEEE-exact, so every e is bit-identical to
 * the host's; only the ORDER of the fp64 sums differs (fixed tree here, channel-major walk there).
 * Both sums are sums of N non-negative terms, so any two orders agree to 2 N 2^-53 relative; the
 * printed line (%f: six decimals) is therefore THE SAME unless a rounding boundary (k + 0.5) 1e-6
 * lies inside that interval around the value.  compare_finish_kernel checks exactly that, and
 * when it does - or when the context asks for it - one lane walks the stream in the reference's
 * order (channel by channel, sample by sample, separately rounded multiply and add: no FMA
 * contraction), which gives the reference's doubles bit for bit (tests/test_gpu_reconstruct.py
 * runs every case that way too and compares with ==).
 *
 * HBM-bound by construction: 4 B read (+2 B written for the residual) per value, nothing else.
 */
#ifndef AAD_COMPARE_HIP_H_INCLUDED
#define AAD_COMPARE_HIP_H_INCLUDED

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "aad_decode.hip.h" /* StreamDesc, find_stream */

namespace aad {

constexpr uint32_t kCompareThreads = 256;
constexpr uint32_t kCompareValuesPerThread = 32;
constexpr uint32_t kCompareSegment = kCompareThreads * kCompareValuesPerThread; /* int16 values per workgroup */

struct ErrorPartial {
  double sum_sq, sum_abs, max_abs;
};

struct ErrorStatsRecord { /* == AADHipErrorStats */
  double rms_error, mean_abs_error, max_abs_error;
};

struct CompareArgs {
  const StreamDesc *streams;
  const uint64_t *segment_prefix; /* [num_streams + 1] exclusive prefix sum of segments per stream */
  const int16_t *original;
  int16_t *decoded;      /* read; overwritten with the residual when write_residual != 0 */
  ErrorPartial *partials; /* [total_segments]; null when no statistics are wanted */
  ErrorStatsRecord *stats; /* [num_streams] */
  uint64_t total_segments;
  uint32_t num_streams;
  uint32_t channels;
  uint32_t write_residual;
  uint32_t sequential; /* 1: every stream's sums are taken in the reference's order (AAD_HIP_OPTION_COMPARE_ORDER) */
};

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  return v;
}

/* one workgroup per kCompareSegment values of one stream */
__global__ __launch_bounds__(kCompareThreads) void compare_segments_kernel(CompareArgs a)
{
  __shared__ ErrorPartial wave_part[kCompareThreads / 64];
  const uint64_t g = blockIdx.x;
  if (g >= a.total_segments) return;
  const uint32_t s = find_stream(a.segment_prefix, a.num_streams, g);
  const StreamDesc sd = a.streams[s];
  const uint64_t count = (uint64_t)sd.num_samples * a.channels;
  const uint64_t first = (g - a.segment_prefix[s]) * kCompareSegment;
  const int16_t *x = a.original + sd.pcm_offset;
  int16_t *y = a.decoded + sd.pcm_offset;

  double sq = 0.0, ab = 0.0, mx = 0.0;
#pragma unroll 4
  for (uint32_t k = 0; k < kCompareValuesPerThread; k++) {
    const uint64_t i = first + (uint64_t)k * kCompareThreads + threadIdx.x;
    if (i >= count) break;
    const int32_t xv = x[i], yv = y[i];
    const int32_t gap = (int32_t)(((uint32_t)xv << 16) - ((uint32_t)yv << 16)); /* src/main.c:470-474 */
    if (a.partials != nullptr) {
      const double e = (double)gap / 2147483647.0 - (double)yv / 2147483647.0; /* src/main.c:483-486 */
      const double m = fabs(e);
      sq += e * e;
      ab += m;
      mx = fmax(mx, m);
    }
    if (a.write_residual) y[i] = (int16_t)(gap >> 16); /* the 16-bit writer keeps the top half, src/wav.c:429 */
  }
  if (a.partials == nullptr) return;
  sq = wave_sum(sq);
  ab = wave_sum(ab);
  mx = wave_max(mx);
  const uint32_t wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    wave_part[wave].sum_sq = sq;
    wave_part[wave].sum_abs = ab;
    wave_part[wave].max_abs = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    ErrorPartial p = wave_part[0];
    for (uint32_t w = 1; w < kCompareThreads / 64; w++) {
      p.sum_sq += wave_part[w].sum_sq;
      p.sum_abs += wave_part[w].sum_abs;
      p.max_abs = fmax(p.max_abs, wave_part[w].max_abs);
    }
    a.partials[g] = p;
  }
}

/* The reference's loop (src/main.c:478-491) for one stream: channel by channel, sample by sample, every operation
 * rounded by itself.  HIP compiles with -ffp-contract=fast and __dmul_rn / __dadd_rn are plain operators there: the
 * pragma is what keeps e * e and the addition from becoming one fused multiply-add (one rounding instead of two: the
 * last bit of the sum differed from the host's). */
__device__ __attribute__((noinline)) void sums_in_reference_order(const int16_t *x, const int16_t *y, uint32_t num_samples,
                                                                  uint32_t channels, double *sum_sq, double *sum_abs)
{
#pragma clang fp contract(off)
  double ssq = 0.0, sab = 0.0;
  for (uint32_t c = 0; c < channels; c++) {
    for (uint32_t i = 0; i < num_samples; i++) {
      const size_t at = (size_t)i * channels + c;
      const int32_t xv = x[at], yv = y[at];
      const int32_t gap = (int32_t)(((uint32_t)xv << 16) - ((uint32_t)yv << 16));
      const double p1 = __ddiv_rn((double)gap, 2147483647.0), p2 = __ddiv_rn((double)yv, 2147483647.0);
      const double e = p1 - p2;
      const double sq = e * e; /* pow(e, 2) == e * e, rounded once */
      ssq = ssq + sq;
      sab = sab + fabs(e);
    }
  }
  *sum_sq = ssq;
  *sum_abs = sab;
}

/* one wave per stream: fold its segment partials in a fixed order, then the CLI's final three lines */
__global__ __launch_bounds__(64) void compare_finish_kernel(CompareArgs a)
{
  const uint32_t s = blockIdx.x;
  if (s >= a.num_streams) return;
  const uint64_t lo = a.segment_prefix[s], hi = a.segment_prefix[s + 1];
  double sq = 0.0, ab = 0.0, mx = 0.0;
  for (uint64_t g = lo + threadIdx.x; g < hi; g += 64) {
    const ErrorPartial p = a.partials[g];
    sq += p.sum_sq;
    ab += p.sum_abs;
    mx = fmax(mx, p.max_abs);
  }
  sq = wave_sum(sq);
  ab = wave_sum(ab);
  mx = wave_max(mx);
  if (threadIdx.x == 0) {
    /* the CLI divides by the uint32 product (src/main.c:494-496) */
    const uint32_t count = a.channels * a.streams[s].num_samples;
    const double n = (double)count;
    ErrorStatsRecord r;
    r.rms_error = __dsqrt_rn(__ddiv_rn(sq, n)); /* correctly rounded forms: the host's sqrt and division are */
    r.mean_abs_error = __ddiv_rn(ab, n);
    r.max_abs_error = mx; /* a maximum does not depend on the order */
    /* Could the reference's summation order print another sixth decimal?  |tree sum - sequential sum| <= 2 N u |sum|
     * (u = 2^-53, non-negative terms); the square root halves a relative error, the division and the root add an ulp
     * each - rel covers both with room to spare. */
    const double rel = 2.5 * n * 1.1102230246251565e-16 + 1e-15;
    auto crosses = [](double v, double rel_err) { /* a %f rounding boundary inside v (1 +- rel_err)? */
      const double lo = v * (1.0 - rel_err) * 1e6, hi = v * (1.0 + rel_err) * 1e6;
      return floor(lo + 0.5 - 1e-9) != floor(hi + 0.5 + 1e-9); /* the 1e-9: the products above are rounded themselves */
    };
    const bool tie = crosses(r.rms_error, rel) || crosses(r.mean_abs_error, rel);
    if ((tie || a.sequential) && !a.write_residual) { /* (with the residual written over `decoded` the terms are gone) */
      const StreamDesc sd = a.streams[s];
      double ssq = 0.0, sab = 0.0;
      sums_in_reference_order(a.original + sd.pcm_offset, a.decoded + sd.pcm_offset, sd.num_samples, a.channels, &ssq, &sab);
      r.rms_error = __dsqrt_rn(__ddiv_rn(ssq, n));
      r.mean_abs_error = __ddiv_rn(sab, n);
    }
    a.stats[s] = r;
  }
}

} /* namespace aad */

#endif /* AAD_COMPARE_HIP_H_INCLUDED */
