// Does gfx950 serve LDS reads / writes at addresses that are not naturally aligned, and at what cost?
// (design input for the sector-tiled I/O of the dense kernels: a lane reads its code bytes / PCM out of
// an LDS ring at a byte phase that follows from the .aad layout - 49 or 67 bytes into an image.)
// For each instruction and byte offset: every lane reads at  base + 144 * lane + offset  (the ring pitch
// planned for the kernels), the result is checked against the bytes that were put there, and a dependent
// chain of 256 reads is timed with s_memtime (latency) plus an independent stream (issue cost).
// build: hipcc --offload-arch=gfx950 -O2 -o ubench_lds_unaligned ubench_lds_unaligned.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))

struct Out { uint64_t cycles; uint32_t got[4]; uint32_t pad[2]; };

template <int WIDTH, bool WRITE>
__global__ void probe(Out *out, int offset)
{
  __shared__ __attribute__((aligned(16))) uint8_t lds[64 * 144 + 64];
  for (int i = threadIdx.x; i < (int)sizeof(lds); i += blockDim.x) lds[i] = (uint8_t)(i * 7 + 3);
  __syncthreads();
  const uint32_t addr = (uint32_t)(uintptr_t)(lds) + 144u * threadIdx.x + (uint32_t)offset;
  uint32_t a = addr;
  uint32_t d0 = 0, d1 = 0, d2 = 0, d3 = 0;
  uint64_t t0, t1;
  if (WRITE) {
    d0 = 0x03020100u + threadIdx.x; d1 = 0x07060504u; d2 = 0x0b0a0908u; d3 = 0x0f0e0d0cu;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (WIDTH == 4) asm volatile(R64("ds_write_b32 %0, %1\n") "s_waitcnt lgkmcnt(0)" :: "v"(a), "v"(d0) : "memory");
    if (WIDTH == 8) asm volatile(R64("ds_write_b64 %0, %1\n") "s_waitcnt lgkmcnt(0)" :: "v"(a), "v"((uint64_t)d0 | ((uint64_t)d1 << 32)) : "memory");
    if (WIDTH == 16) {
      typedef uint32_t u4 __attribute__((ext_vector_type(4)));
      u4 v = {d0, d1, d2, d3};
      asm volatile(R64("ds_write_b128 %0, %1\n") "s_waitcnt lgkmcnt(0)" :: "v"(a), "v"(v) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    __syncthreads();
    // read the bytes back one by one (always legal)
    uint32_t g[4] = {0, 0, 0, 0};
    for (int i = 0; i < WIDTH; i++) g[i / 4] |= (uint32_t)lds[144u * threadIdx.x + offset + i] << (8 * (i % 4));
    d0 = g[0]; d1 = g[1]; d2 = g[2]; d3 = g[3];
  } else {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (WIDTH == 4) asm volatile(R64("ds_read_b32 %0, %1\n") "s_waitcnt lgkmcnt(0)" : "=v"(d0) : "v"(a) : "memory");
    if (WIDTH == 8) {
      uint64_t q;
      asm volatile(R64("ds_read_b64 %0, %1\n") "s_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(a) : "memory");
      d0 = (uint32_t)q; d1 = (uint32_t)(q >> 32);
    }
    if (WIDTH == 12) {
      typedef uint32_t u3 __attribute__((ext_vector_type(3)));
      u3 q;
      asm volatile(R64("ds_read_b96 %0, %1\n") "s_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(a) : "memory");
      d0 = q.x; d1 = q.y; d2 = q.z;
    }
    if (WIDTH == 16) {
      typedef uint32_t u4 __attribute__((ext_vector_type(4)));
      u4 q;
      asm volatile(R64("ds_read_b128 %0, %1\n") "s_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(a) : "memory");
      d0 = q.x; d1 = q.y; d2 = q.z; d3 = q.w;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  }
  Out o;
  o.cycles = t1 - t0;
  o.got[0] = d0; o.got[1] = d1; o.got[2] = d2; o.got[3] = d3;
  out[threadIdx.x] = o;
}

template <int WIDTH, bool WRITE>
void run(Out *d_out, const char *name)
{
  for (int offset : {0, 1, 2, 3, 4, 6, 8, 12, 13}) {
    hipLaunchKernelGGL((probe<WIDTH, WRITE>), dim3(1), dim3(64), 0, 0, d_out, offset);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s offset %d: HIP ERROR (%s)\n", name, offset, hipGetErrorString(hipGetLastError())); return; }
    std::vector<Out> h(64);
    hipMemcpy(h.data(), d_out, sizeof(Out) * 64, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) {
      for (int i = 0; i < WIDTH; i++) {
        uint8_t want;
        if (WRITE) {
          const uint32_t src[4] = {0x03020100u + (uint32_t)l, 0x07060504u, 0x0b0a0908u, 0x0f0e0d0cu};
          want = (uint8_t)(src[i / 4] >> (8 * (i % 4)));
        } else {
          want = (uint8_t)((144 * l + offset + i) * 7 + 3);
        }
        const uint8_t got = (uint8_t)(h[l].got[i / 4] >> (8 * (i % 4)));
        if (got != want) bad++;
      }
    }
    printf("%-14s offset %2d: %s  %6.1f cycles per instruction (64 back to back, one wave)\n", name, offset,
           bad ? "WRONG BYTES" : "bytes ok   ", (double)h[0].cycles / 64.0);
  }
}

int main()
{
  Out *d_out;
  hipMalloc(&d_out, sizeof(Out) * 64);
  run<4, false>(d_out, "ds_read_b32");
  run<8, false>(d_out, "ds_read_b64");
  run<12, false>(d_out, "ds_read_b96");
  run<16, false>(d_out, "ds_read_b128");
  run<4, true>(d_out, "ds_write_b32");
  run<8, true>(d_out, "ds_write_b64");
  run<16, true>(d_out, "ds_write_b128");
  return 0;
}
