// Is a gfx950 wave slower per instruction when only part of EXEC is set?  Hot (instruction-cache
// resident) loops, lone wave, EXEC = all lanes / the low half of every 16-lane row / one quad per row.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define R4(x) x x x x
#define R8(x) R4(x) R4(x)
#define DEFINE_TEST(NAME, BODY) \
  __global__ void k_##NAME(uint64_t *out, int *sink, int seed, uint64_t mask) { \
    uint64_t t0, t1; \
    __shared__ int lds[2048]; lds[threadIdx.x] = seed; __syncthreads(); \
    asm volatile("v_mov_b32 v4, %0\n v_mov_b32 v5, %0\n v_mov_b32 v6, %0\n v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n v_mov_b32 v20, %0\n" :: "v"(seed) : "v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35"); \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_mov_b64 s[22:23], exec\n s_mov_b64 exec, %1\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) : "s"(mask) : "memory", "s22", "s23"); \
    asm volatile("s_movk_i32 s20, 400\n .p2align 8\n 1:\n" R8(BODY) "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n" ::: "memory", "vcc", "scc", "s20", "s6", "s7", "v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35"); \
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)\n s_mov_b64 exec, s[22:23]" : "=s"(t1) :: "memory"); \
    int r; asm volatile("v_add_u32 %0, v20, v21" : "=v"(r) :: "v20","v21"); \
    if (threadIdx.x == 0) out[0] = t1 - t0; \
    sink[threadIdx.x] = r + lds[threadIdx.x]; \
  }
DEFINE_TEST(add, "v_add_u32 v20, v4, v5\n v_add_u32 v21, v4, v5\n v_add_u32 v22, v4, v5\n v_add_u32 v23, v4, v5\n ")
DEFINE_TEST(dep_add, "v_add_u32 v20, v20, v5\n v_add_u32 v20, v20, v5\n v_add_u32 v20, v20, v5\n v_add_u32 v20, v20, v5\n ")
DEFINE_TEST(mad64, "v_mad_u64_u32 v[24:25], s[6:7], v4, v5, v[12:13]\n v_add_u32 v26, v4, v5\n v_add_u32 v20, v4, v5\n v_add_u32 v27, v4, v5\n ")
DEFINE_TEST(dpp_add, "v_add_u32 v20, v4, v5\n v_add_u32 v26, v4, v5\n v_add_u32 v27, v4, v5\n v_add_u32_dpp v21, v20, v20 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n ")
DEFINE_TEST(dpp_shift, "v_add_u32 v20, v4, v5\n v_add_u32 v26, v4, v5\n v_add_u32 v27, v4, v5\n v_and_b32_dpp v21, v20, v6 quad_perm:[0,0,1,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n ")
DEFINE_TEST(fma_cvt, "v_cvt_f32_i32 v20, v4\n v_fma_f32 v21, |v20|, v5, v6\n v_cvt_u32_f32 v22, v21\n v_min_u32 v23, 7, v22\n ")
DEFINE_TEST(lds96, "v_and_b32 v28, 0xff0, v4\n ds_read_b96 v[32:34], v28\n v_add_u32 v26, v4, v5\n v_add_u32 v27, v4, v5\n v_add_u32 v20, v4, v5\n v_add_u32 v21, v4, v5\n v_add_u32 v22, v4, v5\n s_waitcnt lgkmcnt(0)\n ")
DEFINE_TEST(mulhi, "v_mul_hi_u32 v20, v4, v5\n v_mul_hi_u32 v21, v4, v5\n v_mul_hi_u32 v22, v4, v5\n v_mul_hi_u32 v23, v4, v5\n ")
DEFINE_TEST(cmp_cnd, "v_cmp_lt_u32 vcc, v4, v5\n v_cndmask_b32 v20, v4, v5, vcc\n v_add_u32 v26, v4, v5\n v_add_u32 v27, v4, v5\n ")
struct T { const char *name; void (*fn)(uint64_t *, int *, int, uint64_t); int per; };
int main() {
  uint64_t *d_out; int *d_sink;
  if (hipMalloc(&d_out, 64) != hipSuccess || hipMalloc(&d_sink, sizeof(int) * 64) != hipSuccess) return 1;
  T tests[] = { {"add", k_add, 4}, {"dep_add", k_dep_add, 4}, {"mad64", k_mad64, 4}, {"dpp_add", k_dpp_add, 4}, {"dpp_shift", k_dpp_shift, 4}, {"fma_cvt", k_fma_cvt, 4}, {"lds96", k_lds96, 8}, {"mulhi", k_mulhi, 4}, {"cmp_cnd", k_cmp_cnd, 4} };
  const uint64_t masks[3] = { ~0ull, 0x00FF00FF00FF00FFull, 0x000F000F000F000Full };
  printf("%-12s %10s %10s %10s   (cycles per instruction)\n", "test", "all", "half rows", "one quad");
  for (auto &t : tests) {
    printf("%-12s", t.name);
    for (int m = 0; m < 3; m++) {
      uint64_t best = ~0ull;
      for (int rep = 0; rep < 4; rep++) {
        hipLaunchKernelGGL(t.fn, dim3(1), dim3(64), 0, 0, d_out, d_sink, 7, masks[m]);
        if (hipDeviceSynchronize() != hipSuccess) { printf(" HIP ERROR\n"); return 1; }
        uint64_t h = 0;
        if (hipMemcpy(&h, d_out, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
        if (h < best) best = h;
      }
      printf(" %10.2f", (double)best / (400.0 * 8 * t.per));
    }
    printf("\n"); fflush(stdout);
  }
  return 0;
}
