// Instruction-cost microbenchmarks for gfx950 (design input for the AAD kernels, not product code).
// Each test times a block of 256 copies of one instruction pattern with s_memtime, for a wave
// running ALONE on its SIMD (grid = 1 workgroup of 64 threads) and for 8 waves per SIMD
// (workgroup of 1024... here: 16 workgroups x 256 threads on one CU is not controllable, so
// the "busy" variant launches 256 CUs x 2048 threads and reports wave-cycles / instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
#define R256(x) R4(R64(x))

#define DEFINE_TEST(NAME, BODY)                                                        \
  __global__ void k_##NAME(uint64_t *out, int *sink, int seed) {                      \
    int a = seed + threadIdx.x, b = seed * 3 + 1, c = seed * 5 + 2, d = seed * 7 + 3;  \
    int e = seed + 11, f = seed + 13, g = seed + 17, h = seed + 19;                    \
    float fa = (float)a, fb = (float)b; long long l0 = (long long)(sink + blockIdx.x * blockDim.x + threadIdx.x), l1 = b;                                              \
    __shared__ int lds[1024];                                                          \
    lds[threadIdx.x & 1023] = (threadIdx.x * 8) & 1023;                                \
    __syncthreads();                                                                   \
    uint64_t t0, t1;                                                                   \
    uint64_t best = ~0ull;                                                             \
    for (int it = 0; it < 6; it++) {                                                   \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory"); \
      asm volatile(R256(BODY) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(fa), "+v"(fb), "+v"(l0), "+v"(l1) :: "memory", "vcc", "s20", "s21", "v200", "v201", "v202"); \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory"); \
      if (t1 - t0 < best) best = t1 - t0;                                              \
    }                                                                                  \
    if (threadIdx.x == 0) out[blockIdx.x] = best;                                      \
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h + (int)fa + (int)fb + (int)l1; \
  }

// operands: %0..%7 ints a..h, %8 %9 floats
DEFINE_TEST(add_dep,        "v_add_u32 %0, %0, %1\n")
DEFINE_TEST(add_ind4,       "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n")
DEFINE_TEST(mad24_dep,      "v_mad_i32_i24 %0, %0, %1, %2\n")
DEFINE_TEST(mad24_ind4,     "v_mad_i32_i24 %0, %4, %5, %0\n v_mad_i32_i24 %1, %4, %5, %1\n v_mad_i32_i24 %2, %4, %5, %2\n v_mad_i32_i24 %3, %4, %5, %3\n")
DEFINE_TEST(mullo_dep,      "v_mul_lo_u32 %0, %0, %1\n")
DEFINE_TEST(mullo_ind4,     "v_mul_lo_u32 %0, %4, %5\n v_mul_lo_u32 %1, %4, %6\n v_mul_lo_u32 %2, %4, %7\n v_mul_lo_u32 %3, %5, %6\n")
DEFINE_TEST(mulhi24_ind4,   "v_mul_hi_i32_i24 %0, %4, %5\n v_mul_hi_i32_i24 %1, %4, %6\n v_mul_hi_i32_i24 %2, %4, %7\n v_mul_hi_i32_i24 %3, %5, %6\n")
DEFINE_TEST(mulhi32_ind4,   "v_mul_hi_i32 %0, %4, %5\n v_mul_hi_i32 %1, %4, %6\n v_mul_hi_i32 %2, %4, %7\n v_mul_hi_i32 %3, %5, %6\n")
DEFINE_TEST(med3_dep,       "v_med3_i32 %0, %0, %1, %2\n")
DEFINE_TEST(add3_dep,       "v_add3_u32 %0, %0, %1, %2\n")
DEFINE_TEST(xad_dep,        "v_xad_u32 %0, %0, %1, %2\n")
DEFINE_TEST(ashr_add,       "v_ashrrev_i32 %1, 18, %0\n v_add_u32 %0, %1, %2\n")
DEFINE_TEST(cvt_mul_cvt,    "v_cvt_f32_u32 %8, %0\n v_mul_f32 %8, %8, %9\n v_cvt_u32_f32 %0, %8\n")
DEFINE_TEST(cvtf_ind,       "v_cvt_f32_u32 %8, %0\n v_cvt_f32_u32 %9, %1\n")
DEFINE_TEST(cmp_cndmask,    "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %2, %3, vcc\n")
DEFINE_TEST(bfe_ind,        "v_bfe_i32 %0, %4, 3, 4\n v_bfe_u32 %1, %4, 7, 3\n")
DEFINE_TEST(perm_ind,       "v_perm_b32 %0, %4, %5, %6\n v_perm_b32 %1, %4, %5, %7\n")
DEFINE_TEST(dpp_mov,        "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n")
DEFINE_TEST(dpp_add_dep,    "v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
DEFINE_TEST(dpp_add_ind,    "v_add_u32_dpp %0, %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %4, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n")
DEFINE_TEST(sdwa_sub,       "v_sub_u32_sdwa %0, sext(%4), %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_sub_u32_sdwa %1, sext(%4), %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n")
DEFINE_TEST(madu64_ind2,    "v_mad_u64_u32 %10, vcc, %4, %5, %11\n")
DEFINE_TEST(ds_read_dep,    "ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n")
DEFINE_TEST(ds_read_b64_ind,"ds_read_b32 %1, %0\n ds_read_b32 %2, %0 offset:4\n")
DEFINE_TEST(salu_mix,       "v_add_u32 %0, %0, %1\n s_add_u32 s20, s20, 1\n")
DEFINE_TEST(salu_only,      "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n")
DEFINE_TEST(snop,           "s_nop 0\n")
DEFINE_TEST(pk_add_i16,     "v_pk_add_i16 %0, %0, %1\n")
DEFINE_TEST(mad_i32_i16,    "v_mad_i32_i16 %0, %1, %2, %0\n")
DEFINE_TEST(dot2_i32_i16,   "v_dot2_i32_i16 %0, %1, %2, %0\n")
DEFINE_TEST(lshl_add,       "v_lshl_add_u32 %0, %0, 1, %1\n")
DEFINE_TEST(lshl_or,        "v_lshl_or_b32 %0, %0, 1, 1\n")
DEFINE_TEST(and_or,         "v_and_or_b32 %0, %0, %1, %2\n")
DEFINE_TEST(min_u32,        "v_min_u32 %0, %0, %1\n")
DEFINE_TEST(madu64_dep,     "v_mad_u64_u32 %10, vcc, %4, %5, %10\n")
DEFINE_TEST(pred_madu64,    "v_mad_u64_u32 %10, vcc, %0, %4, %11\n v_mad_u64_u32 %10, vcc, %1, %5, %10\n v_mad_u64_u32 %10, vcc, %2, %6, %10\n v_mad_u64_u32 %10, vcc, %3, %7, %10\n v_ashrrev_i32 %0, 15, %1\n")
DEFINE_TEST(pred_mullo,     "v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %5\n v_mul_lo_u32 %2, %2, %6\n v_mul_lo_u32 %3, %3, %7\n v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %0, %0, %3, %4\n v_ashrrev_i32 %0, 15, %0\n")
DEFINE_TEST(pred_mad24,     "v_mad_i32_i24 %1, %0, %4, %5\n v_mad_i32_i24 %1, %1, %5, %1\n v_mad_i32_i24 %1, %2, %6, %1\n v_mad_i32_i24 %1, %3, %7, %1\n v_ashrrev_i32 %0, 15, %1\n")
DEFINE_TEST(pred_quad_dpp,  "v_mad_i32_i24 %1, %0, %4, %5\n v_add_u32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_ashrrev_i32 %0, 15, %1\n")
DEFINE_TEST(cndmask_dpp,    "v_cndmask_b32_dpp %0, %0, %1, vcc quad_perm:[0,0,1,2] row_mask:0xf bank_mask:0xf\n")
DEFINE_TEST(fma_abs,        "v_cvt_f32_i32 %8, %0\n v_fma_f32 %8, |%8|, %9, %9\n v_cvt_u32_f32 %0, %8\n v_min_u32 %0, 7, %0\n")
DEFINE_TEST(lms3,           "v_mad_i32_i24 %1, %0, %4, %5\n v_ashrrev_i32 %1, 18, %1\n v_add_u32 %0, %0, %1\n")
DEFINE_TEST(ds_read_b128_dep, "ds_read_b64 %10, %0\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0xff0, %0\n")
DEFINE_TEST(gstore_short,   "global_store_short %10, %0, off\n global_store_short %10, %1, off offset:4\n")
DEFINE_TEST(gstore_dwordx2, "global_store_dwordx2 %10, %11, off\n")
DEFINE_TEST(ds_read_issue,  "ds_read_b32 %1, %0\n ds_read_b32 %2, %0 offset:16\n ds_read_b32 %3, %0 offset:32\n ds_read_b32 %4, %0 offset:48\n")
DEFINE_TEST(waitcnt_only,   "s_waitcnt lgkmcnt(0)\n")
DEFINE_TEST(lds_lat_0, "ds_read_b32 %5, %0\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0x3fc, %5\n")
DEFINE_TEST(lds_lat_4, "ds_read_b32 %5, %0\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0x3fc, %5\n")
DEFINE_TEST(lds_lat_8, "ds_read_b32 %5, %0\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0x3fc, %5\n")
DEFINE_TEST(lds_lat_12, "ds_read_b32 %5, %0\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0x3fc, %5\n")
DEFINE_TEST(lds_lat_16, "ds_read_b32 %5, %0\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0x3fc, %5\n")
DEFINE_TEST(lds_lat_20, "ds_read_b32 %5, %0\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0x3fc, %5\n")
DEFINE_TEST(lds_lat_24, "ds_read_b32 %5, %0\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0x3fc, %5\n")
DEFINE_TEST(lds_lat_32, "ds_read_b32 %5, %0\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0x3fc, %5\n")
DEFINE_TEST(lds96_lat_0, "ds_read_b96 v[200:202], %0\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0xf0, v200\n")
DEFINE_TEST(lds96_lat_8, "ds_read_b96 v[200:202], %0\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0xf0, v200\n")
DEFINE_TEST(lds96_lat_16, "ds_read_b96 v[200:202], %0\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0xf0, v200\n")
DEFINE_TEST(lds96_lat_24, "ds_read_b96 v[200:202], %0\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0xf0, v200\n")
DEFINE_TEST(add_f64,        "v_add_f64 %10, %10, %11\n")

__global__ void k_clock(uint64_t *out) {
  uint64_t c0, r0, c1, r1; int x = threadIdx.x;
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0) :: "memory");
  for (int i = 0; i < 200000; i++) { asm volatile("v_add_u32 %0, %0, 1" : "+v"(x)); }
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1) :: "memory");
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = x; }
}
struct T { const char *name; void (*fn)(uint64_t *, int *, int); int per; };

int main() {
  uint64_t *d_out; int *d_sink;
  hipMalloc(&d_out, sizeof(uint64_t) * 4096);
  hipMalloc(&d_sink, sizeof(int) * 4096 * 1024);
#define E(n, per) {#n, k_##n, per}
  std::vector<T> tests = { E(add_dep,1), E(add_ind4,4), E(mad24_dep,1), E(mad24_ind4,4), E(mullo_dep,1), E(mullo_ind4,4),
    E(mulhi24_ind4,4), E(mulhi32_ind4,4), E(med3_dep,1), E(add3_dep,1), E(xad_dep,1), E(ashr_add,2), E(cvt_mul_cvt,3), E(cvtf_ind,2),
    E(cmp_cndmask,2), E(bfe_ind,2), E(perm_ind,2), E(dpp_mov,1), E(dpp_add_dep,1), E(dpp_add_ind,2), E(sdwa_sub,2),
    E(madu64_ind2,1), E(madu64_dep,1), E(pred_madu64,5), E(pred_mullo,7), E(pred_mad24,5), E(pred_quad_dpp,4), E(cndmask_dpp,1), E(fma_abs,4), E(lms3,3), E(ds_read_b128_dep,2), E(lds_lat_0,3), E(lds_lat_4,7), E(lds_lat_8,11), E(lds_lat_12,15), E(lds_lat_16,19), E(lds_lat_20,23), E(lds_lat_24,27), E(lds_lat_32,35), E(lds96_lat_0,3), E(lds96_lat_8,11), E(lds96_lat_16,19), E(lds96_lat_24,27),  E(gstore_short,2), E(gstore_dwordx2,1), E(ds_read_issue,4), E(waitcnt_only,1), E(add_f64,1), E(ds_read_dep,1), E(ds_read_b64_ind,2), E(salu_mix,2), E(salu_only,2), E(snop,1), E(pk_add_i16,1), E(mad_i32_i16,1),
    E(dot2_i32_i16,1), E(lshl_add,1), E(lshl_or,1), E(and_or,1), E(min_u32,1) };
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, d_out);
    hipDeviceSynchronize();
    uint64_t h[3]; hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    printf("clock probe: %llu shader cycles in %llu x 10ns -> %.1f MHz\n", (unsigned long long)h[0], (unsigned long long)h[1], (double)h[0] / ((double)h[1] * 0.01));
  }
  printf("%-18s %12s %12s %12s\n", "test", "1wave cyc/ins", "1w/simd all CUs", "4w/simd all CUs"); fflush(stdout);
  for (auto &t : tests) {
    double res[3];
    printf("%-18s", t.name); fflush(stdout);
    int cfg_threads[3] = {64, 256, 256};   // 1 wave; 8 waves = 2/SIMD; 16 waves = 4/SIMD (x2 blocks -> 8/SIMD)
    int cfg_blocks[3] = {1, 256, 512};
    for (int c = 0; c < 3; c++) {
      hipMemset(d_out, 0, sizeof(uint64_t) * 4096);
      hipLaunchKernelGGL(t.fn, dim3(cfg_blocks[c]), dim3(cfg_threads[c]), 0, 0, d_out, d_sink, 7);
      if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { printf(" HIP ERROR\n"); fflush(stdout); return 1; }
      std::vector<uint64_t> h(cfg_blocks[c]);
      hipMemcpy(h.data(), d_out, sizeof(uint64_t) * cfg_blocks[c], hipMemcpyDeviceToHost);
      double s = 0; for (auto v : h) s += (double)v;
      res[c] = s / cfg_blocks[c] / (256.0 * t.per);
    }
    printf(" %12.2f %12.2f %12.2f\n", res[0], res[1], res[2]); fflush(stdout);
  }
  return 0;
}
