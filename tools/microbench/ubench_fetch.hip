// HBM-traffic calibration for the access patterns of the AAD kernels (design input, not product code).
// MI355X_MICROARCH.md (HBM): "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ...
// Other access widths are uncalibrated: calibrate on a known byte count in your own access pattern before
// trusting an absolute."  This program IS that calibration: every kernel reads (or writes) every byte of a
// 2 GiB buffer exactly once, so the true byte count is known, in the patterns the codec's kernels use:
//   coalesced   lane i of a wave touches 16 B at  base + 16 i                  (1 KiB contiguous per instruction)
//   lane64      ONE lane walks its own 4 KiB row sector by sector: 4 x 16 B back to back per visit, the lanes of
//               a wave a row apart (the dense kernels today: lane = stream / block, 64-byte bursts)
//   lane64u     the same with every burst 8 bytes off the sector grid (a burst straddles two sectors: the mono
//               kernels, whose code bytes start 49 bytes into an image)
//   lane16      the same in single 16-byte pieces, one per visit (the chunk-by-chunk kernels of round 1)
//   quad64      FOUR adjacent lanes cover one 64-byte sector of a row, 16 rows per instruction (the sector-tiled
//               I/O planned for the dense kernels: rows staged through LDS)
//   oct128      EIGHT adjacent lanes cover one 128-byte line of a row, 8 rows per instruction
// Each pattern runs as a read kernel and as a write kernel.  Run once plainly (prints GB/s from HIP events) and
// once per counter group under rocprofv3 --pmc (FETCH_SIZE | WRITE_SIZE | TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
// TCC_BUBBLE_sum | TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum): counter / true bytes is the calibration factor.
// build: hipcc --offload-arch=gfx950 -O2 -o ubench_fetch ubench_fetch.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(1))) U4 { u4 v; };

constexpr uint64_t kBytes = 2ull << 30; /* far beyond L2 (32 MiB) and the Infinity Cache (256 MiB) */
constexpr uint32_t kRow = 4096;         /* bytes per lane-row: the pitch between the streams of the bench batch is 3968-4032 */
constexpr uint64_t kRows = kBytes / kRow;

enum { kCoalesced, kLane64, kLane64u, kLane16, kQuad64, kOct128 };

template <int PATTERN, bool WRITE>
__global__ void __launch_bounds__(256) sweep(uint8_t *buf, uint32_t *sink)
{
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  u4 acc = {0, 0, 0, 0};
  const u4 val = {(uint32_t)t, 1u, 2u, 3u};
  auto touch = [&](uint8_t *p) {
    if (WRITE) reinterpret_cast<U4 *>(p)->v = val;
    else {
      const u4 q = reinterpret_cast<const U4 *>(p)->v;
      acc.x ^= q.x; acc.y ^= q.y; acc.z ^= q.z; acc.w ^= q.w;
    }
  };
  if (PATTERN == kCoalesced) { /* every thread: 16 B per step, the grid strides over the buffer */
    const uint64_t total = kBytes / 16, step = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = t; i < total; i += step) touch(buf + 16 * i);
  } else if (PATTERN == kLane64 || PATTERN == kLane64u || PATTERN == kLane16) { /* thread = row */
    if (t < kRows) {
      uint8_t *row = buf + t * kRow;
      if (PATTERN == kLane16) {
        for (uint32_t o = 0; o < kRow; o += 16) touch(row + o);
      } else {
        const uint32_t off = PATTERN == kLane64u ? 8u : 0u; /* the last burst of a row runs 8 bytes into the next row: still every byte once */
        for (uint32_t o = 0; o < kRow; o += 64) {
#pragma unroll
          for (int k = 0; k < 4; k++) touch(row + off + o + 16 * k);
        }
      }
    }
  } else { /* a wave owns 64 rows; G adjacent lanes cover one sector (64 B) or line (128 B) of a row per instruction */
    constexpr uint32_t G = PATTERN == kQuad64 ? 4 : 8, kPiece = 16 * G, kRowsPerInst = 64 / G;
    const uint64_t wave = t >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    if (wave * 64 < kRows) {
      uint8_t *rows = buf + wave * 64 * kRow;
      for (uint32_t o = 0; o < kRow; o += kPiece) {
#pragma unroll
        for (uint32_t j = 0; j < 64 / kRowsPerInst; j++) touch(rows + (uint64_t)(j * kRowsPerInst + lane / G) * kRow + o + 16 * (lane % G));
      }
    }
  }
  if (!WRITE) sink[t & 0xFFFFF] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

template <int PATTERN, bool WRITE>
void run(const char *name, uint8_t *buf, uint32_t *sink)
{
  const uint32_t threads = PATTERN == kCoalesced ? 256 * 256 * 8 : (uint32_t)kRows;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((sweep<PATTERN, WRITE>), dim3(threads / 256), dim3(256), 0, 0, buf, sink);
    hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) { printf("%s: HIP ERROR\n", name); return; }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("%-10s %-5s true bytes %llu  %8.3f ms  %7.1f GB/s\n", name, WRITE ? "write" : "read", (unsigned long long)kBytes, best, kBytes / (best * 1e-3) / 1e9);
  fflush(stdout);
}

int main(int argc, char **argv)
{
  uint8_t *buf;
  uint32_t *sink;
  if (hipMalloc(&buf, kBytes + 4096) != hipSuccess || hipMalloc(&sink, 4u << 20) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  hipMemset(buf, 1, kBytes + 4096);
  const char *only = argc > 1 ? argv[1] : "";
  auto want = [&](const char *n) { return !*only || !strcmp(only, n); };
  if (want("coalesced")) { run<kCoalesced, false>("coalesced", buf, sink); run<kCoalesced, true>("coalesced", buf, sink); }
  if (want("lane64")) { run<kLane64, false>("lane64", buf, sink); run<kLane64, true>("lane64", buf, sink); }
  if (want("lane64u")) { run<kLane64u, false>("lane64u", buf, sink); run<kLane64u, true>("lane64u", buf, sink); }
  if (want("lane16")) { run<kLane16, false>("lane16", buf, sink); run<kLane16, true>("lane16", buf, sink); }
  if (want("quad64")) { run<kQuad64, false>("quad64", buf, sink); run<kQuad64, true>("quad64", buf, sink); }
  if (want("oct128")) { run<kOct128, false>("oct128", buf, sink); run<kOct128, true>("oct128", buf, sink); }
  return 0;
}
