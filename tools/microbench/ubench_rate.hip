// VALU issue-RATE microbenchmark for gfx950 (design input, not product code): settles what one SIMD
// sustains per wave64 vector instruction when 1, 2, 4 or 8 waves share it.
//
// Question (round-2 verdict, item 1c): MI355X_MICROARCH.md says a wave64 VALU instruction occupies a
// SIMD-32 for 2 cycles, "one wave alone: 4"; round 1's ubench reported 4.05 cycles per instruction
// "at 4 waves/SIMD" - measured with s_memtime INSIDE one wave, i.e. the cadence one wave sees, not
// what the SIMD retires.  This program measures the aggregate: every wave runs ITERS x 64 independent
// instructions of one kind; k workgroups of 256 threads (4 waves, one per SIMD) are launched per CU;
// rate = instructions of all waves of a SIMD / cycles from the first wave's start to the last wave's
// end on that SIMD (s_memtime; cross-checked against the launch's wall time from HIP events).
// Each wave also records its HW_ID so that the waves-per-SIMD the run really had is printed, not assumed.
//
// build: hipcc --offload-arch=gfx950 -O2 -o ubench_rate ubench_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>

#define R8(x) x x x x x x x x
#define ITERS 1024

struct Rec { uint64_t t0, t1; uint32_t hw_id, xcc_id; };

#define DEFINE_RATE(NAME, BODY)                                                                   \
  __global__ void __launch_bounds__(256) r_##NAME(Rec *out, int *sink, int seed) {                \
    int a = seed + threadIdx.x, b = seed * 3 + 1, c = seed * 5 + 2, d = seed * 7 + 3;             \
    int e = seed + 11, f = seed + 13, g = seed + 17, h = seed + 19;                               \
    int s0 = seed * 9 + threadIdx.x, s1 = seed + 23, s2 = seed | 1;                               \
    float fa = (float)a, fb = 1.0001f;                                                            \
    long long l0 = a, l1 = b, l2 = c, l3 = d;                                                     \
    __shared__ int lds[4096];                                                                     \
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;                              \
    __syncthreads();                                                                              \
    uint64_t t0, t1;                                                                              \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory"); \
    for (int it = 0; it < ITERS; it++) {                                                          \
      asm volatile(R8(BODY) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), \
                   "+v"(fa), "+v"(fb), "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3)                     \
                   : "v"(s0), "v"(s1), "v"(s2) : "vcc", "memory", "s40", "s41", "s42", "s43");    \
    }                                                                                             \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory"); \
    if ((threadIdx.x & 63) == 0) {                                                                \
      uint32_t hw, xcc;                                                                           \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n s_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc)); \
      out[blockIdx.x * 4 + (threadIdx.x >> 6)] = Rec{t0, t1, hw, xcc};                            \
    }                                                                                             \
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h + (int)fa + (int)(l0 + l1 + l2 + l3); \
  }

// operands: %0..%7 ints a..h (destinations, one per chain), %8 %9 floats, %10..%13 64-bit, %14 %15 %16 read-only sources
// every BODY is 8 instructions on 8 different destinations: no instruction depends on one less than 8 back
DEFINE_RATE(add_u32,      "v_add_u32 %0, %0, %14\n v_add_u32 %1, %1, %14\n v_add_u32 %2, %2, %14\n v_add_u32 %3, %3, %14\n v_add_u32 %4, %4, %14\n v_add_u32 %5, %5, %14\n v_add_u32 %6, %6, %14\n v_add_u32 %7, %7, %14\n")
DEFINE_RATE(and_or,       "v_and_or_b32 %0, %0, %14, %15\n v_and_or_b32 %1, %1, %14, %15\n v_and_or_b32 %2, %2, %14, %15\n v_and_or_b32 %3, %3, %14, %15\n v_and_or_b32 %4, %4, %14, %15\n v_and_or_b32 %5, %5, %14, %15\n v_and_or_b32 %6, %6, %14, %15\n v_and_or_b32 %7, %7, %14, %15\n")
DEFINE_RATE(mad_i32_i24,  "v_mad_i32_i24 %0, %14, %15, %0\n v_mad_i32_i24 %1, %14, %15, %1\n v_mad_i32_i24 %2, %14, %15, %2\n v_mad_i32_i24 %3, %14, %15, %3\n v_mad_i32_i24 %4, %14, %15, %4\n v_mad_i32_i24 %5, %14, %15, %5\n v_mad_i32_i24 %6, %14, %15, %6\n v_mad_i32_i24 %7, %14, %15, %7\n")
DEFINE_RATE(mul_lo_u32,   "v_mul_lo_u32 %0, %0, %14\n v_mul_lo_u32 %1, %1, %14\n v_mul_lo_u32 %2, %2, %14\n v_mul_lo_u32 %3, %3, %14\n v_mul_lo_u32 %4, %4, %14\n v_mul_lo_u32 %5, %5, %14\n v_mul_lo_u32 %6, %6, %14\n v_mul_lo_u32 %7, %7, %14\n")
DEFINE_RATE(mul_hi_u24,   "v_mul_hi_u32_u24 %0, %0, %14\n v_mul_hi_u32_u24 %1, %1, %14\n v_mul_hi_u32_u24 %2, %2, %14\n v_mul_hi_u32_u24 %3, %3, %14\n v_mul_hi_u32_u24 %4, %4, %14\n v_mul_hi_u32_u24 %5, %5, %14\n v_mul_hi_u32_u24 %6, %6, %14\n v_mul_hi_u32_u24 %7, %7, %14\n")
DEFINE_RATE(mul_hi_u32,   "v_mul_hi_u32 %0, %0, %14\n v_mul_hi_u32 %1, %1, %14\n v_mul_hi_u32 %2, %2, %14\n v_mul_hi_u32 %3, %3, %14\n v_mul_hi_u32 %4, %4, %14\n v_mul_hi_u32 %5, %5, %14\n v_mul_hi_u32 %6, %6, %14\n v_mul_hi_u32 %7, %7, %14\n")
DEFINE_RATE(mad_u64_u32,  "v_mad_u64_u32 %10, vcc, %14, %15, %10\n v_mad_u64_u32 %11, vcc, %14, %15, %11\n v_mad_u64_u32 %12, vcc, %14, %15, %12\n v_mad_u64_u32 %13, vcc, %14, %15, %13\n v_mad_u64_u32 %10, vcc, %14, %16, %10\n v_mad_u64_u32 %11, vcc, %14, %16, %11\n v_mad_u64_u32 %12, vcc, %14, %16, %12\n v_mad_u64_u32 %13, vcc, %14, %16, %13\n")
DEFINE_RATE(mad_i64_i32,  "v_mad_i64_i32 %10, vcc, %14, %15, %10\n v_mad_i64_i32 %11, vcc, %14, %15, %11\n v_mad_i64_i32 %12, vcc, %14, %15, %12\n v_mad_i64_i32 %13, vcc, %14, %15, %13\n v_mad_i64_i32 %10, vcc, %14, %16, %10\n v_mad_i64_i32 %11, vcc, %14, %16, %11\n v_mad_i64_i32 %12, vcc, %14, %16, %12\n v_mad_i64_i32 %13, vcc, %14, %16, %13\n")
DEFINE_RATE(fma_f32,      "v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n")
DEFINE_RATE(cvt_f32_i32,  "v_cvt_f32_i32 %0, %14\n v_cvt_f32_i32 %1, %15\n v_cvt_f32_i32 %2, %14\n v_cvt_f32_i32 %3, %15\n v_cvt_f32_i32 %4, %14\n v_cvt_f32_i32 %5, %15\n v_cvt_f32_i32 %6, %14\n v_cvt_f32_i32 %7, %15\n")
DEFINE_RATE(cvt_u32_f32,  "v_cvt_u32_f32 %0, %8\n v_cvt_u32_f32 %1, %9\n v_cvt_u32_f32 %2, %8\n v_cvt_u32_f32 %3, %9\n v_cvt_u32_f32 %4, %8\n v_cvt_u32_f32 %5, %9\n v_cvt_u32_f32 %6, %8\n v_cvt_u32_f32 %7, %9\n")
DEFINE_RATE(perm_b32,     "v_perm_b32 %0, %0, %14, %15\n v_perm_b32 %1, %1, %14, %15\n v_perm_b32 %2, %2, %14, %15\n v_perm_b32 %3, %3, %14, %15\n v_perm_b32 %4, %4, %14, %15\n v_perm_b32 %5, %5, %14, %15\n v_perm_b32 %6, %6, %14, %15\n v_perm_b32 %7, %7, %14, %15\n")
DEFINE_RATE(lshl_or,      "v_lshl_or_b32 %0, %0, 4, %14\n v_lshl_or_b32 %1, %1, 4, %14\n v_lshl_or_b32 %2, %2, 4, %14\n v_lshl_or_b32 %3, %3, 4, %14\n v_lshl_or_b32 %4, %4, 4, %14\n v_lshl_or_b32 %5, %5, 4, %14\n v_lshl_or_b32 %6, %6, 4, %14\n v_lshl_or_b32 %7, %7, 4, %14\n")
DEFINE_RATE(med3_i32,     "v_med3_i32 %0, %0, %14, %15\n v_med3_i32 %1, %1, %14, %15\n v_med3_i32 %2, %2, %14, %15\n v_med3_i32 %3, %3, %14, %15\n v_med3_i32 %4, %4, %14, %15\n v_med3_i32 %5, %5, %14, %15\n v_med3_i32 %6, %6, %14, %15\n v_med3_i32 %7, %7, %14, %15\n")
DEFINE_RATE(min_u32,      "v_min_u32 %0, %0, %14\n v_min_u32 %1, %1, %14\n v_min_u32 %2, %2, %14\n v_min_u32 %3, %3, %14\n v_min_u32 %4, %4, %14\n v_min_u32 %5, %5, %14\n v_min_u32 %6, %6, %14\n v_min_u32 %7, %7, %14\n")
DEFINE_RATE(ashr_i32,     "v_ashrrev_i32 %0, 3, %14\n v_ashrrev_i32 %1, 3, %15\n v_ashrrev_i32 %2, 3, %14\n v_ashrrev_i32 %3, 3, %15\n v_ashrrev_i32 %4, 3, %14\n v_ashrrev_i32 %5, 3, %15\n v_ashrrev_i32 %6, 3, %14\n v_ashrrev_i32 %7, 3, %15\n")
DEFINE_RATE(add_dpp_ror4, "v_add_u32_dpp %0, %14, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %14, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %14, %2 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %14, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %4, %14, %4 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %5, %14, %5 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %6, %14, %6 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %7, %14, %7 row_ror:4 row_mask:0xf bank_mask:0xf\n")
DEFINE_RATE(mov_dpp_shr4, "v_mov_b32_dpp %0, %14 row_shr:4 row_mask:0xf bank_mask:0xe\n v_mov_b32_dpp %1, %14 row_shr:4 row_mask:0xf bank_mask:0xe\n v_mov_b32_dpp %2, %14 row_shr:4 row_mask:0xf bank_mask:0xe\n v_mov_b32_dpp %3, %14 row_shr:4 row_mask:0xf bank_mask:0xe\n v_mov_b32_dpp %4, %14 row_shr:4 row_mask:0xf bank_mask:0xe\n v_mov_b32_dpp %5, %14 row_shr:4 row_mask:0xf bank_mask:0xe\n v_mov_b32_dpp %6, %14 row_shr:4 row_mask:0xf bank_mask:0xe\n v_mov_b32_dpp %7, %14 row_shr:4 row_mask:0xf bank_mask:0xe\n")
DEFINE_RATE(sub_sdwa,     "v_sub_u32_sdwa %0, sext(%14), %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n v_sub_u32_sdwa %1, sext(%14), %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_sub_u32_sdwa %2, sext(%14), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n v_sub_u32_sdwa %3, sext(%14), %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_sub_u32_sdwa %4, sext(%14), %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n v_sub_u32_sdwa %5, sext(%14), %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_sub_u32_sdwa %6, sext(%14), %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n v_sub_u32_sdwa %7, sext(%14), %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n")
DEFINE_RATE(add_f64,      "v_add_f64 %10, %10, %12\n v_add_f64 %11, %11, %12\n v_add_f64 %10, %10, %13\n v_add_f64 %11, %11, %13\n v_add_f64 %10, %10, %12\n v_add_f64 %11, %11, %12\n v_add_f64 %10, %10, %13\n v_add_f64 %11, %11, %13\n")
DEFINE_RATE(pk_add_i16,   "v_pk_add_i16 %0, %0, %14\n v_pk_add_i16 %1, %1, %14\n v_pk_add_i16 %2, %2, %14\n v_pk_add_i16 %3, %3, %14\n v_pk_add_i16 %4, %4, %14\n v_pk_add_i16 %5, %5, %14\n v_pk_add_i16 %6, %6, %14\n v_pk_add_i16 %7, %7, %14\n")
// the dense encoder's instruction mix for one sample, roughly (cvt, fma, cvt, min, lookups excluded, mul_hi, xor/sub,
// med3, 64-bit multiply-adds, adds): what "VALU peak" means for THIS kernel
DEFINE_RATE(encoder_mix,  "v_cvt_f32_i32 %0, %14\n v_fma_f32 %1, |%8|, %9, %9\n v_mul_hi_u32_u24 %2, %14, %15\n v_med3_i32 %3, %14, %15, %16\n v_mad_i64_i32 %10, vcc, %14, %15, %10\n v_add_u32 %4, %4, %14\n v_mad_u64_u32 %11, vcc, %14, %16, %11\n v_and_or_b32 %5, %5, %14, %15\n")

DEFINE_RATE(sub_u32, "v_sub_u32 %0, %0, %14\n v_sub_u32 %1, %1, %14\n v_sub_u32 %2, %2, %14\n v_sub_u32 %3, %3, %14\n v_sub_u32 %4, %4, %14\n v_sub_u32 %5, %5, %14\n v_sub_u32 %6, %6, %14\n v_sub_u32 %7, %7, %14\n")
DEFINE_RATE(and_b32, "v_and_b32 %0, %0, %14\n v_and_b32 %1, %1, %14\n v_and_b32 %2, %2, %14\n v_and_b32 %3, %3, %14\n v_and_b32 %4, %4, %14\n v_and_b32 %5, %5, %14\n v_and_b32 %6, %6, %14\n v_and_b32 %7, %7, %14\n")
DEFINE_RATE(or_b32, "v_or_b32 %0, %0, %14\n v_or_b32 %1, %1, %14\n v_or_b32 %2, %2, %14\n v_or_b32 %3, %3, %14\n v_or_b32 %4, %4, %14\n v_or_b32 %5, %5, %14\n v_or_b32 %6, %6, %14\n v_or_b32 %7, %7, %14\n")
DEFINE_RATE(xor_b32, "v_xor_b32 %0, %0, %14\n v_xor_b32 %1, %1, %14\n v_xor_b32 %2, %2, %14\n v_xor_b32 %3, %3, %14\n v_xor_b32 %4, %4, %14\n v_xor_b32 %5, %5, %14\n v_xor_b32 %6, %6, %14\n v_xor_b32 %7, %7, %14\n")
DEFINE_RATE(lshlrev, "v_lshlrev_b32 %0, 3, %14\n v_lshlrev_b32 %1, 3, %15\n v_lshlrev_b32 %2, 3, %14\n v_lshlrev_b32 %3, 3, %15\n v_lshlrev_b32 %4, 3, %14\n v_lshlrev_b32 %5, 3, %15\n v_lshlrev_b32 %6, 3, %14\n v_lshlrev_b32 %7, 3, %15\n")
DEFINE_RATE(lshrrev, "v_lshrrev_b32 %0, 3, %14\n v_lshrrev_b32 %1, 3, %15\n v_lshrrev_b32 %2, 3, %14\n v_lshrrev_b32 %3, 3, %15\n v_lshrrev_b32 %4, 3, %14\n v_lshrrev_b32 %5, 3, %15\n v_lshrrev_b32 %6, 3, %14\n v_lshrrev_b32 %7, 3, %15\n")
DEFINE_RATE(lshlrev_v, "v_lshlrev_b32 %0, %16, %14\n v_lshlrev_b32 %1, %16, %15\n v_lshlrev_b32 %2, %16, %14\n v_lshlrev_b32 %3, %16, %15\n v_lshlrev_b32 %4, %16, %14\n v_lshlrev_b32 %5, %16, %15\n v_lshlrev_b32 %6, %16, %14\n v_lshlrev_b32 %7, %16, %15\n")
DEFINE_RATE(mov_b32, "v_mov_b32 %0, %14\n v_mov_b32 %1, %15\n v_mov_b32 %2, %14\n v_mov_b32 %3, %15\n v_mov_b32 %4, %14\n v_mov_b32 %5, %15\n v_mov_b32 %6, %14\n v_mov_b32 %7, %15\n")
DEFINE_RATE(max_i32, "v_max_i32 %0, %0, %14\n v_max_i32 %1, %1, %14\n v_max_i32 %2, %2, %14\n v_max_i32 %3, %3, %14\n v_max_i32 %4, %4, %14\n v_max_i32 %5, %5, %14\n v_max_i32 %6, %6, %14\n v_max_i32 %7, %7, %14\n")
DEFINE_RATE(min_i32, "v_min_i32 %0, %0, %14\n v_min_i32 %1, %1, %14\n v_min_i32 %2, %2, %14\n v_min_i32 %3, %3, %14\n v_min_i32 %4, %4, %14\n v_min_i32 %5, %5, %14\n v_min_i32 %6, %6, %14\n v_min_i32 %7, %7, %14\n")
DEFINE_RATE(add3_u32, "v_add3_u32 %0, %0, %14, %15\n v_add3_u32 %1, %1, %14, %15\n v_add3_u32 %2, %2, %14, %15\n v_add3_u32 %3, %3, %14, %15\n v_add3_u32 %4, %4, %14, %15\n v_add3_u32 %5, %5, %14, %15\n v_add3_u32 %6, %6, %14, %15\n v_add3_u32 %7, %7, %14, %15\n")
DEFINE_RATE(lshl_add, "v_lshl_add_u32 %0, %0, 2, %14\n v_lshl_add_u32 %1, %1, 2, %14\n v_lshl_add_u32 %2, %2, 2, %14\n v_lshl_add_u32 %3, %3, 2, %14\n v_lshl_add_u32 %4, %4, 2, %14\n v_lshl_add_u32 %5, %5, 2, %14\n v_lshl_add_u32 %6, %6, 2, %14\n v_lshl_add_u32 %7, %7, 2, %14\n")
DEFINE_RATE(xad_u32, "v_xad_u32 %0, %0, %14, %15\n v_xad_u32 %1, %1, %14, %15\n v_xad_u32 %2, %2, %14, %15\n v_xad_u32 %3, %3, %14, %15\n v_xad_u32 %4, %4, %14, %15\n v_xad_u32 %5, %5, %14, %15\n v_xad_u32 %6, %6, %14, %15\n v_xad_u32 %7, %7, %14, %15\n")
DEFINE_RATE(bfe_i32, "v_bfe_i32 %0, %14, 3, 4\n v_bfe_i32 %1, %15, 3, 4\n v_bfe_i32 %2, %14, 3, 4\n v_bfe_i32 %3, %15, 3, 4\n v_bfe_i32 %4, %14, 3, 4\n v_bfe_i32 %5, %15, 3, 4\n v_bfe_i32 %6, %14, 3, 4\n v_bfe_i32 %7, %15, 3, 4\n")
DEFINE_RATE(cndmask, "v_cndmask_b32 %0, %0, %14, vcc\n v_cndmask_b32 %1, %1, %14, vcc\n v_cndmask_b32 %2, %2, %14, vcc\n v_cndmask_b32 %3, %3, %14, vcc\n v_cndmask_b32 %4, %4, %14, vcc\n v_cndmask_b32 %5, %5, %14, vcc\n v_cndmask_b32 %6, %6, %14, vcc\n v_cndmask_b32 %7, %7, %14, vcc\n")
DEFINE_RATE(cmp_lt_i32, "v_cmp_lt_i32 vcc, %0, %14\n v_cmp_lt_i32 vcc, %1, %14\n v_cmp_lt_i32 vcc, %2, %14\n v_cmp_lt_i32 vcc, %3, %14\n v_cmp_lt_i32 vcc, %4, %14\n v_cmp_lt_i32 vcc, %5, %14\n v_cmp_lt_i32 vcc, %6, %14\n v_cmp_lt_i32 vcc, %7, %14\n")
DEFINE_RATE(add_co_u32, "v_add_co_u32 %0, vcc, %0, %14\n v_add_co_u32 %1, vcc, %1, %14\n v_add_co_u32 %2, vcc, %2, %14\n v_add_co_u32 %3, vcc, %3, %14\n v_add_co_u32 %4, vcc, %4, %14\n v_add_co_u32 %5, vcc, %5, %14\n v_add_co_u32 %6, vcc, %6, %14\n v_add_co_u32 %7, vcc, %7, %14\n")
DEFINE_RATE(add_f32, "v_add_f32 %0, %0, %14\n v_add_f32 %1, %1, %14\n v_add_f32 %2, %2, %14\n v_add_f32 %3, %3, %14\n v_add_f32 %4, %4, %14\n v_add_f32 %5, %5, %14\n v_add_f32 %6, %6, %14\n v_add_f32 %7, %7, %14\n")
DEFINE_RATE(mul_f32, "v_mul_f32 %0, %0, %14\n v_mul_f32 %1, %1, %14\n v_mul_f32 %2, %2, %14\n v_mul_f32 %3, %3, %14\n v_mul_f32 %4, %4, %14\n v_mul_f32 %5, %5, %14\n v_mul_f32 %6, %6, %14\n v_mul_f32 %7, %7, %14\n")
DEFINE_RATE(mul_u32_u24, "v_mul_u32_u24 %0, %0, %14\n v_mul_u32_u24 %1, %1, %14\n v_mul_u32_u24 %2, %2, %14\n v_mul_u32_u24 %3, %3, %14\n v_mul_u32_u24 %4, %4, %14\n v_mul_u32_u24 %5, %5, %14\n v_mul_u32_u24 %6, %6, %14\n v_mul_u32_u24 %7, %7, %14\n")
DEFINE_RATE(mul_i32_i24, "v_mul_i32_i24 %0, %0, %14\n v_mul_i32_i24 %1, %1, %14\n v_mul_i32_i24 %2, %2, %14\n v_mul_i32_i24 %3, %3, %14\n v_mul_i32_i24 %4, %4, %14\n v_mul_i32_i24 %5, %5, %14\n v_mul_i32_i24 %6, %6, %14\n v_mul_i32_i24 %7, %7, %14\n")
DEFINE_RATE(pk_fma_f32, "v_pk_fma_f32 %10, %12, %13, %10\n v_pk_fma_f32 %11, %12, %13, %11\n v_pk_fma_f32 %10, %12, %13, %10\n v_pk_fma_f32 %11, %12, %13, %11\n v_pk_fma_f32 %10, %12, %13, %10\n v_pk_fma_f32 %11, %12, %13, %11\n v_pk_fma_f32 %10, %12, %13, %10\n v_pk_fma_f32 %11, %12, %13, %11\n")
DEFINE_RATE(decoder_mix, "v_mad_i64_i32 %10, vcc, %14, %15, %10\n v_add_u32 %0, %0, %14\n v_med3_i32 %1, %14, %15, %16\n v_mad_i64_i32 %11, vcc, %14, %16, %11\n v_add_u32 %2, %2, %14\n v_mad_u64_u32 %12, vcc, %14, %15, %12\n v_ashrrev_i32 %3, 15, %14\n v_add_u32_sdwa %4, %14, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n")

struct T { const char *name; void (*fn)(Rec *, int *, int); };

int main()
{
  Rec *d_out;
  int *d_sink;
  const int max_blocks = 256 * 8;
  hipMalloc(&d_out, sizeof(Rec) * max_blocks * 4);
  hipMalloc(&d_sink, sizeof(int) * max_blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
#define E(n) {#n, r_##n}
  std::vector<T> tests = {E(add_u32), E(and_or), E(mad_i32_i24), E(mul_lo_u32), E(mul_hi_u24), E(mul_hi_u32), E(mad_u64_u32), E(mad_i64_i32),
                          E(fma_f32), E(cvt_f32_i32), E(cvt_u32_f32), E(perm_b32), E(lshl_or), E(med3_i32), E(min_u32), E(ashr_i32),
                          E(add_dpp_ror4), E(mov_dpp_shr4), E(sub_sdwa), E(add_f64), E(pk_add_i16), E(encoder_mix), E(sub_u32), E(and_b32), E(or_b32), E(xor_b32), E(lshlrev), E(lshrrev), E(lshlrev_v), E(mov_b32), E(max_i32), E(min_i32), E(add3_u32), E(lshl_add), E(xad_u32), E(bfe_i32), E(cndmask), E(cmp_lt_i32), E(add_co_u32), E(add_f32), E(mul_f32), E(mul_u32_u24), E(mul_i32_i24), E(pk_fma_f32), E(decoder_mix)};
  const double insts_per_wave = (double)ITERS * 64.0;
  printf("# every wave issues %d x 64 independent instructions of one kind; k workgroups of 256 threads per CU (256 k in all)\n", ITERS);
  printf("# cyc/inst seen by a wave = mean over waves of (its own s_memtime span) / instructions\n");
  printf("# cyc/inst per SIMD       = (first start .. last end of the waves that shared a SIMD) / (their instructions), median over SIMDs\n");
  printf("# G wave-inst/s           = all instructions / launch wall time (HIP events), best of 5\n");
  printf("%-14s %4s %10s %12s %14s %14s %12s\n", "instruction", "k", "waves/SIMD", "cyc/inst(wave)", "cyc/inst(SIMD)", "G wave-inst/s", "wall us");
  for (auto &t : tests) {
    for (int k : {1, 2, 4, 8}) {
      const int blocks = 256 * k;
      float best_ms = 1e9f;
      for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), 0, 0, d_out, d_sink, 7);
        hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess || hipGetLastError() != hipSuccess) {
          printf("HIP ERROR in %s\n", t.name);
          return 1;
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best_ms = std::min(best_ms, ms);
      }
      std::vector<Rec> h(blocks * 4);
      hipMemcpy(h.data(), d_out, sizeof(Rec) * blocks * 4, hipMemcpyDeviceToHost);
      // group the waves of the LAST launch by the SIMD they ran on: xcc | se | sh | cu | simd out of HW_ID
      std::map<uint32_t, std::vector<Rec>> by_simd;
      double wave_cyc = 0;
      for (auto &r : h) {
        const uint32_t simd = (r.hw_id >> 4) & 3u, cu = (r.hw_id >> 8) & 15u, sh = (r.hw_id >> 12) & 1u, se = (r.hw_id >> 13) & 7u;
        by_simd[(r.xcc_id & 15u) << 16 | se << 12 | sh << 8 | cu << 4 | simd].push_back(r);
        wave_cyc += (double)(r.t1 - r.t0);
      }
      std::vector<double> per_simd;
      double waves_per_simd = 0;
      for (auto &kv : by_simd) {
        uint64_t lo = ~0ull, hi = 0;
        for (auto &r : kv.second) {
          lo = std::min(lo, r.t0);
          hi = std::max(hi, r.t1);
        }
        per_simd.push_back((double)(hi - lo) / (insts_per_wave * kv.second.size()));
        waves_per_simd += kv.second.size();
      }
      std::sort(per_simd.begin(), per_simd.end());
      printf("%-14s %4d %10.2f %12.2f %14.2f %14.1f %12.1f\n", t.name, k, waves_per_simd / by_simd.size(),
             wave_cyc / h.size() / insts_per_wave, per_simd[per_simd.size() / 2],
             insts_per_wave * h.size() / (best_ms * 1e-3) / 1e9, best_ms * 1e3);
      fflush(stdout);
    }
  }
  return 0;
}
