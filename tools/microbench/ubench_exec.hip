// Does a gfx950 wave64 VALU instruction get cheaper when only part of EXEC is set?
// (If the SIMD skipped the 16-lane passes whose EXEC bits are all zero, a recurrence-bound
// kernel could run 16-lane waves up to 4x faster.)  Times 256 copies of an instruction pattern
// for a lone wave with EXEC = all 64, the low 32, the low 16 and the low 4 lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
#define R256(x) R4(R64(x))

#define DEFINE_TEST(NAME, BODY)                                                                  \
  __global__ void k_##NAME(uint64_t *out, int *sink, int seed, uint64_t mask) {                  \
    int a = seed + threadIdx.x, b = seed * 3 + 1, c = seed * 5 + 2, d = seed * 7 + 3;            \
    int e = seed + 11, f = seed + 13;                                                            \
    uint64_t t0, t1, best = ~0ull;                                                               \
    for (int it = 0; it < 6; it++) {                                                             \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_mov_b64 s[20:21], exec\n s_mov_b64 exec, %1\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) : "s"(mask) : "memory", "s20", "s21"); \
      asm volatile(R256(BODY) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) :: "memory", "vcc"); \
      asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)\n s_mov_b64 exec, s[20:21]" : "=s"(t1) :: "memory"); \
      if (t1 - t0 < best) best = t1 - t0;                                                        \
    }                                                                                            \
    if (threadIdx.x == 0) out[blockIdx.x] = best;                                                \
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f;                         \
  }

DEFINE_TEST(add_dep,   "v_add_u32 %0, %0, %1\n")
DEFINE_TEST(add_ind4,  "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n")
DEFINE_TEST(mad24_dep, "v_mad_i32_i24 %0, %0, %1, %2\n")
DEFINE_TEST(mullo_ind4,"v_mul_lo_u32 %0, %4, %5\n v_mul_lo_u32 %1, %4, %5\n v_mul_lo_u32 %2, %4, %5\n v_mul_lo_u32 %3, %4, %5\n")
DEFINE_TEST(dpp_add,   "v_add_u32_dpp %0, %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %4, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n")

struct T { const char *name; void (*fn)(uint64_t *, int *, int, uint64_t); int per; };

int main() {
  uint64_t *d_out; int *d_sink;
  if (hipMalloc(&d_out, 64) != hipSuccess || hipMalloc(&d_sink, sizeof(int) * 64) != hipSuccess) return 1;
#define E(n, per) {#n, k_##n, per}
  T tests[] = { E(add_dep,1), E(add_ind4,4), E(mad24_dep,1), E(mullo_ind4,4), E(dpp_add,2) };
  const uint64_t masks[4] = { ~0ull, 0xffffffffull, 0xffffull, 0xfull };
  printf("%-12s %10s %10s %10s %10s   (cycles per instruction, lone wave)\n", "test", "exec=64", "exec=32", "exec=16", "exec=4");
  for (auto &t : tests) {
    printf("%-12s", t.name);
    for (int m = 0; m < 4; m++) {
      hipLaunchKernelGGL(t.fn, dim3(1), dim3(64), 0, 0, d_out, d_sink, 7, masks[m]);
      if (hipDeviceSynchronize() != hipSuccess) { printf(" HIP ERROR\n"); return 1; }
      uint64_t h; hipMemcpy(&h, d_out, sizeof(h), hipMemcpyDeviceToHost);
      printf(" %10.2f", (double)h / (256.0 * t.per));
    }
    printf("\n"); fflush(stdout);
  }
  return 0;
}
