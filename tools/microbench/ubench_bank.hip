// VGPR operand placement vs issue cost for a lone gfx950 wave, measured on I$-resident loops
// (64 instructions per iteration, 400 iterations).  Generated table of register choices.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define R4(x) x x x x
#define R8(x) R4(x) R4(x)
#define DEFINE_TEST(NAME, BODY) \
  __global__ void k_##NAME(uint64_t *out, int *sink, int seed) { \
    uint64_t t0, t1; \
    asm volatile("v_mov_b32 v4, %0\n v_mov_b32 v5, %0\n v_mov_b32 v6, %0\n v_mov_b32 v7, %0\n v_mov_b32 v8, %0\n v_mov_b32 v9, %0\n v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n v_mov_b32 v14, %0\n v_mov_b32 v15, %0\n v_mov_b32 v16, %0\n v_mov_b32 v20, %0\n v_mov_b32 v21, %0\n v_mov_b32 v22, %0\n" :: "v"(seed) : "v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44"); \
    asm volatile("s_mov_b32 s8, 0x11111111\n s_mov_b32 s9, 0x11111111\n s_mov_b64 vcc, s[8:9]\n" ::: "s8", "s9", "vcc"); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory"); \
    asm volatile("s_movk_i32 s20, 400\n .p2align 8\n 1:\n" R8(BODY) "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n" ::: "memory", "vcc", "scc", "s20", "s6", "s7", "v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44"); \
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory"); \
    int r; asm volatile("v_add_u32 %0, v20, v21\n v_add_u32 %0, %0, v22" : "=v"(r) :: "v20","v21","v22"); \
    if (threadIdx.x == 0) out[0] = t1 - t0; \
    sink[threadIdx.x] = r; \
  }
DEFINE_TEST(mad_4_8_4, "v_mad_i32_i24 v20, v4, v8, v4\n v_mad_i32_i24 v21, v4, v8, v4\n v_mad_i32_i24 v22, v4, v8, v4\n v_mad_i32_i24 v23, v4, v8, v4\n v_mad_i32_i24 v24, v4, v8, v4\n v_mad_i32_i24 v25, v4, v8, v4\n v_mad_i32_i24 v26, v4, v8, v4\n v_mad_i32_i24 v27, v4, v8, v4\n ")
DEFINE_TEST(mad_4_8_5, "v_mad_i32_i24 v20, v4, v8, v5\n v_mad_i32_i24 v21, v4, v8, v5\n v_mad_i32_i24 v22, v4, v8, v5\n v_mad_i32_i24 v23, v4, v8, v5\n v_mad_i32_i24 v24, v4, v8, v5\n v_mad_i32_i24 v25, v4, v8, v5\n v_mad_i32_i24 v26, v4, v8, v5\n v_mad_i32_i24 v27, v4, v8, v5\n ")
DEFINE_TEST(mad_4_8_6, "v_mad_i32_i24 v20, v4, v8, v6\n v_mad_i32_i24 v21, v4, v8, v6\n v_mad_i32_i24 v22, v4, v8, v6\n v_mad_i32_i24 v23, v4, v8, v6\n v_mad_i32_i24 v24, v4, v8, v6\n v_mad_i32_i24 v25, v4, v8, v6\n v_mad_i32_i24 v26, v4, v8, v6\n v_mad_i32_i24 v27, v4, v8, v6\n ")
DEFINE_TEST(mad_4_8_7, "v_mad_i32_i24 v20, v4, v8, v7\n v_mad_i32_i24 v21, v4, v8, v7\n v_mad_i32_i24 v22, v4, v8, v7\n v_mad_i32_i24 v23, v4, v8, v7\n v_mad_i32_i24 v24, v4, v8, v7\n v_mad_i32_i24 v25, v4, v8, v7\n v_mad_i32_i24 v26, v4, v8, v7\n v_mad_i32_i24 v27, v4, v8, v7\n ")
DEFINE_TEST(mad_4_8_8, "v_mad_i32_i24 v20, v4, v8, v8\n v_mad_i32_i24 v21, v4, v8, v8\n v_mad_i32_i24 v22, v4, v8, v8\n v_mad_i32_i24 v23, v4, v8, v8\n v_mad_i32_i24 v24, v4, v8, v8\n v_mad_i32_i24 v25, v4, v8, v8\n v_mad_i32_i24 v26, v4, v8, v8\n v_mad_i32_i24 v27, v4, v8, v8\n ")
DEFINE_TEST(mad_4_8_9, "v_mad_i32_i24 v20, v4, v8, v9\n v_mad_i32_i24 v21, v4, v8, v9\n v_mad_i32_i24 v22, v4, v8, v9\n v_mad_i32_i24 v23, v4, v8, v9\n v_mad_i32_i24 v24, v4, v8, v9\n v_mad_i32_i24 v25, v4, v8, v9\n v_mad_i32_i24 v26, v4, v8, v9\n v_mad_i32_i24 v27, v4, v8, v9\n ")
DEFINE_TEST(mad_4_8_10, "v_mad_i32_i24 v20, v4, v8, v10\n v_mad_i32_i24 v21, v4, v8, v10\n v_mad_i32_i24 v22, v4, v8, v10\n v_mad_i32_i24 v23, v4, v8, v10\n v_mad_i32_i24 v24, v4, v8, v10\n v_mad_i32_i24 v25, v4, v8, v10\n v_mad_i32_i24 v26, v4, v8, v10\n v_mad_i32_i24 v27, v4, v8, v10\n ")
DEFINE_TEST(mad_4_8_11, "v_mad_i32_i24 v20, v4, v8, v11\n v_mad_i32_i24 v21, v4, v8, v11\n v_mad_i32_i24 v22, v4, v8, v11\n v_mad_i32_i24 v23, v4, v8, v11\n v_mad_i32_i24 v24, v4, v8, v11\n v_mad_i32_i24 v25, v4, v8, v11\n v_mad_i32_i24 v26, v4, v8, v11\n v_mad_i32_i24 v27, v4, v8, v11\n ")
DEFINE_TEST(mad_4_8_12, "v_mad_i32_i24 v20, v4, v8, v12\n v_mad_i32_i24 v21, v4, v8, v12\n v_mad_i32_i24 v22, v4, v8, v12\n v_mad_i32_i24 v23, v4, v8, v12\n v_mad_i32_i24 v24, v4, v8, v12\n v_mad_i32_i24 v25, v4, v8, v12\n v_mad_i32_i24 v26, v4, v8, v12\n v_mad_i32_i24 v27, v4, v8, v12\n ")
DEFINE_TEST(mad_4_8_13, "v_mad_i32_i24 v20, v4, v8, v13\n v_mad_i32_i24 v21, v4, v8, v13\n v_mad_i32_i24 v22, v4, v8, v13\n v_mad_i32_i24 v23, v4, v8, v13\n v_mad_i32_i24 v24, v4, v8, v13\n v_mad_i32_i24 v25, v4, v8, v13\n v_mad_i32_i24 v26, v4, v8, v13\n v_mad_i32_i24 v27, v4, v8, v13\n ")
DEFINE_TEST(mad_4_8_14, "v_mad_i32_i24 v20, v4, v8, v14\n v_mad_i32_i24 v21, v4, v8, v14\n v_mad_i32_i24 v22, v4, v8, v14\n v_mad_i32_i24 v23, v4, v8, v14\n v_mad_i32_i24 v24, v4, v8, v14\n v_mad_i32_i24 v25, v4, v8, v14\n v_mad_i32_i24 v26, v4, v8, v14\n v_mad_i32_i24 v27, v4, v8, v14\n ")
DEFINE_TEST(mad_4_8_15, "v_mad_i32_i24 v20, v4, v8, v15\n v_mad_i32_i24 v21, v4, v8, v15\n v_mad_i32_i24 v22, v4, v8, v15\n v_mad_i32_i24 v23, v4, v8, v15\n v_mad_i32_i24 v24, v4, v8, v15\n v_mad_i32_i24 v25, v4, v8, v15\n v_mad_i32_i24 v26, v4, v8, v15\n v_mad_i32_i24 v27, v4, v8, v15\n ")
DEFINE_TEST(mad_4_8_16, "v_mad_i32_i24 v20, v4, v8, v16\n v_mad_i32_i24 v21, v4, v8, v16\n v_mad_i32_i24 v22, v4, v8, v16\n v_mad_i32_i24 v23, v4, v8, v16\n v_mad_i32_i24 v24, v4, v8, v16\n v_mad_i32_i24 v25, v4, v8, v16\n v_mad_i32_i24 v26, v4, v8, v16\n v_mad_i32_i24 v27, v4, v8, v16\n ")
DEFINE_TEST(mad_4_4_12, "v_mad_i32_i24 v20, v4, v4, v12\n v_mad_i32_i24 v21, v4, v4, v12\n v_mad_i32_i24 v22, v4, v4, v12\n v_mad_i32_i24 v23, v4, v4, v12\n v_mad_i32_i24 v24, v4, v4, v12\n v_mad_i32_i24 v25, v4, v4, v12\n v_mad_i32_i24 v26, v4, v4, v12\n v_mad_i32_i24 v27, v4, v4, v12\n ")
DEFINE_TEST(mad_4_5_12, "v_mad_i32_i24 v20, v4, v5, v12\n v_mad_i32_i24 v21, v4, v5, v12\n v_mad_i32_i24 v22, v4, v5, v12\n v_mad_i32_i24 v23, v4, v5, v12\n v_mad_i32_i24 v24, v4, v5, v12\n v_mad_i32_i24 v25, v4, v5, v12\n v_mad_i32_i24 v26, v4, v5, v12\n v_mad_i32_i24 v27, v4, v5, v12\n ")
DEFINE_TEST(mad_4_6_12, "v_mad_i32_i24 v20, v4, v6, v12\n v_mad_i32_i24 v21, v4, v6, v12\n v_mad_i32_i24 v22, v4, v6, v12\n v_mad_i32_i24 v23, v4, v6, v12\n v_mad_i32_i24 v24, v4, v6, v12\n v_mad_i32_i24 v25, v4, v6, v12\n v_mad_i32_i24 v26, v4, v6, v12\n v_mad_i32_i24 v27, v4, v6, v12\n ")
DEFINE_TEST(mad_4_7_12, "v_mad_i32_i24 v20, v4, v7, v12\n v_mad_i32_i24 v21, v4, v7, v12\n v_mad_i32_i24 v22, v4, v7, v12\n v_mad_i32_i24 v23, v4, v7, v12\n v_mad_i32_i24 v24, v4, v7, v12\n v_mad_i32_i24 v25, v4, v7, v12\n v_mad_i32_i24 v26, v4, v7, v12\n v_mad_i32_i24 v27, v4, v7, v12\n ")
DEFINE_TEST(mad_4_8_12b, "v_mad_i32_i24 v20, v4, v8, v12\n v_mad_i32_i24 v21, v4, v8, v12\n v_mad_i32_i24 v22, v4, v8, v12\n v_mad_i32_i24 v23, v4, v8, v12\n v_mad_i32_i24 v24, v4, v8, v12\n v_mad_i32_i24 v25, v4, v8, v12\n v_mad_i32_i24 v26, v4, v8, v12\n v_mad_i32_i24 v27, v4, v8, v12\n ")
DEFINE_TEST(mad_d20, "v_mad_i32_i24 v20, v4, v5, v6\n v_mad_i32_i24 v24, v4, v5, v6\n v_mad_i32_i24 v28, v4, v5, v6\n v_mad_i32_i24 v32, v4, v5, v6\n v_mad_i32_i24 v20, v4, v5, v6\n v_mad_i32_i24 v24, v4, v5, v6\n v_mad_i32_i24 v28, v4, v5, v6\n v_mad_i32_i24 v32, v4, v5, v6\n ")
DEFINE_TEST(mad_d21, "v_mad_i32_i24 v21, v4, v5, v6\n v_mad_i32_i24 v25, v4, v5, v6\n v_mad_i32_i24 v29, v4, v5, v6\n v_mad_i32_i24 v33, v4, v5, v6\n v_mad_i32_i24 v21, v4, v5, v6\n v_mad_i32_i24 v25, v4, v5, v6\n v_mad_i32_i24 v29, v4, v5, v6\n v_mad_i32_i24 v33, v4, v5, v6\n ")
DEFINE_TEST(mad_d22, "v_mad_i32_i24 v22, v4, v5, v6\n v_mad_i32_i24 v26, v4, v5, v6\n v_mad_i32_i24 v30, v4, v5, v6\n v_mad_i32_i24 v34, v4, v5, v6\n v_mad_i32_i24 v22, v4, v5, v6\n v_mad_i32_i24 v26, v4, v5, v6\n v_mad_i32_i24 v30, v4, v5, v6\n v_mad_i32_i24 v34, v4, v5, v6\n ")
DEFINE_TEST(mad_d23, "v_mad_i32_i24 v23, v4, v5, v6\n v_mad_i32_i24 v27, v4, v5, v6\n v_mad_i32_i24 v31, v4, v5, v6\n v_mad_i32_i24 v35, v4, v5, v6\n v_mad_i32_i24 v23, v4, v5, v6\n v_mad_i32_i24 v27, v4, v5, v6\n v_mad_i32_i24 v31, v4, v5, v6\n v_mad_i32_i24 v35, v4, v5, v6\n ")
DEFINE_TEST(mad_d24, "v_mad_i32_i24 v24, v4, v5, v6\n v_mad_i32_i24 v28, v4, v5, v6\n v_mad_i32_i24 v32, v4, v5, v6\n v_mad_i32_i24 v36, v4, v5, v6\n v_mad_i32_i24 v24, v4, v5, v6\n v_mad_i32_i24 v28, v4, v5, v6\n v_mad_i32_i24 v32, v4, v5, v6\n v_mad_i32_i24 v36, v4, v5, v6\n ")
DEFINE_TEST(mad_d28, "v_mad_i32_i24 v28, v4, v5, v6\n v_mad_i32_i24 v32, v4, v5, v6\n v_mad_i32_i24 v36, v4, v5, v6\n v_mad_i32_i24 v40, v4, v5, v6\n v_mad_i32_i24 v28, v4, v5, v6\n v_mad_i32_i24 v32, v4, v5, v6\n v_mad_i32_i24 v36, v4, v5, v6\n v_mad_i32_i24 v40, v4, v5, v6\n ")
DEFINE_TEST(mad_d32, "v_mad_i32_i24 v32, v4, v5, v6\n v_mad_i32_i24 v36, v4, v5, v6\n v_mad_i32_i24 v40, v4, v5, v6\n v_mad_i32_i24 v44, v4, v5, v6\n v_mad_i32_i24 v32, v4, v5, v6\n v_mad_i32_i24 v36, v4, v5, v6\n v_mad_i32_i24 v40, v4, v5, v6\n v_mad_i32_i24 v44, v4, v5, v6\n ")
DEFINE_TEST(med3_4_8_12, "v_med3_i32 v20, v4, v8, v12\n v_med3_i32 v21, v4, v8, v12\n v_med3_i32 v22, v4, v8, v12\n v_med3_i32 v23, v4, v8, v12\n v_med3_i32 v24, v4, v8, v12\n v_med3_i32 v25, v4, v8, v12\n v_med3_i32 v26, v4, v8, v12\n v_med3_i32 v27, v4, v8, v12\n ")
DEFINE_TEST(med3_4_5_6, "v_med3_i32 v20, v4, v5, v6\n v_med3_i32 v21, v4, v5, v6\n v_med3_i32 v22, v4, v5, v6\n v_med3_i32 v23, v4, v5, v6\n v_med3_i32 v24, v4, v5, v6\n v_med3_i32 v25, v4, v5, v6\n v_med3_i32 v26, v4, v5, v6\n v_med3_i32 v27, v4, v5, v6\n ")
DEFINE_TEST(add_4_8, "v_add_u32 v20, v4, v8\n v_add_u32 v21, v4, v8\n v_add_u32 v22, v4, v8\n v_add_u32 v23, v4, v8\n v_add_u32 v24, v4, v8\n v_add_u32 v25, v4, v8\n v_add_u32 v26, v4, v8\n v_add_u32 v27, v4, v8\n ")
DEFINE_TEST(add_4_5, "v_add_u32 v20, v4, v5\n v_add_u32 v21, v4, v5\n v_add_u32 v22, v4, v5\n v_add_u32 v23, v4, v5\n v_add_u32 v24, v4, v5\n v_add_u32 v25, v4, v5\n v_add_u32 v26, v4, v5\n v_add_u32 v27, v4, v5\n ")
DEFINE_TEST(perm_4_8_12, "v_perm_b32 v20, v4, v8, v12\n v_perm_b32 v21, v4, v8, v12\n v_perm_b32 v22, v4, v8, v12\n v_perm_b32 v23, v4, v8, v12\n v_perm_b32 v24, v4, v8, v12\n v_perm_b32 v25, v4, v8, v12\n v_perm_b32 v26, v4, v8, v12\n v_perm_b32 v27, v4, v8, v12\n ")
DEFINE_TEST(perm_4_5_6, "v_perm_b32 v20, v4, v5, v6\n v_perm_b32 v21, v4, v5, v6\n v_perm_b32 v22, v4, v5, v6\n v_perm_b32 v23, v4, v5, v6\n v_perm_b32 v24, v4, v5, v6\n v_perm_b32 v25, v4, v5, v6\n v_perm_b32 v26, v4, v5, v6\n v_perm_b32 v27, v4, v5, v6\n ")
DEFINE_TEST(dep_chain_a, "v_mad_i32_i24 v20, v20, v8, v12\n v_mad_i32_i24 v21, v20, v8, v12\n v_mad_i32_i24 v22, v21, v8, v12\n v_mad_i32_i24 v20, v22, v8, v12\n ")
DEFINE_TEST(dep_chain_b, "v_mad_i32_i24 v20, v20, v5, v6\n v_mad_i32_i24 v21, v20, v5, v6\n v_mad_i32_i24 v22, v21, v5, v6\n v_mad_i32_i24 v20, v22, v5, v6\n ")
DEFINE_TEST(cnd_e32, "v_cndmask_b32_e32 v20, v4, v5, vcc\n v_cndmask_b32_e32 v21, v4, v5, vcc\n v_cndmask_b32_e32 v22, v4, v5, vcc\n v_cndmask_b32_e32 v23, v4, v5, vcc\n ")
DEFINE_TEST(cnd_e64, "v_cndmask_b32_e64 v20, v4, v5, s[8:9]\n v_cndmask_b32_e64 v21, v4, v5, s[8:9]\n v_cndmask_b32_e64 v22, v4, v5, s[8:9]\n v_cndmask_b32_e64 v23, v4, v5, s[8:9]\n ")
DEFINE_TEST(mad64_cnd_e32, "v_mad_u64_u32 v[24:25], s[6:7], v4, v5, v[12:13]\n v_add_u32 v26, v4, v5\n v_cndmask_b32_e32 v20, v4, v5, vcc\n v_add_u32 v27, v4, v5\n ")
DEFINE_TEST(mad64_cnd_e64, "v_mad_u64_u32 v[24:25], s[6:7], v4, v5, v[12:13]\n v_add_u32 v26, v4, v5\n v_cndmask_b32_e64 v20, v4, v5, s[8:9]\n v_add_u32 v27, v4, v5\n ")
DEFINE_TEST(mad64_only, "v_mad_u64_u32 v[24:25], s[6:7], v4, v5, v[12:13]\n v_add_u32 v26, v4, v5\n v_add_u32 v20, v4, v5\n v_add_u32 v27, v4, v5\n ")
DEFINE_TEST(cmp_cnd_e64, "v_cmp_lt_u32_e64 s[6:7], v4, v5\n v_add_u32 v26, v4, v5\n v_cndmask_b32_e64 v20, v4, v5, s[8:9]\n v_add_u32 v27, v4, v5\n ")
DEFINE_TEST(bfi, "v_bfi_b32 v20, v4, v5, v6\n v_bfi_b32 v21, v4, v5, v6\n v_bfi_b32 v22, v4, v5, v6\n v_bfi_b32 v23, v4, v5, v6\n ")
DEFINE_TEST(mad64_bfi, "v_mad_u64_u32 v[24:25], s[6:7], v4, v5, v[12:13]\n v_add_u32 v26, v4, v5\n v_bfi_b32 v20, v4, v5, v6\n v_add_u32 v27, v4, v5\n ")
DEFINE_TEST(dppmov_cnd_e32, "v_mov_b32_dpp v20, v4 quad_perm:[0,0,1,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32 v26, v4, v5\n v_cndmask_b32_e32 v21, v20, v5, vcc\n v_add_u32 v27, v4, v5\n ")
DEFINE_TEST(cnd_dpp, "v_cndmask_b32_dpp v20, v4, v5, vcc quad_perm:[0,0,1,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32 v26, v4, v5\n v_add_u32 v21, v20, v5\n v_add_u32 v27, v4, v5\n ")

struct T { const char *name; void (*fn)(uint64_t *, int *, int); int per; };
int main() {
  uint64_t *d_out; int *d_sink;
  if (hipMalloc(&d_out, 64) != hipSuccess || hipMalloc(&d_sink, sizeof(int) * 64) != hipSuccess) return 1;
  T tests[] = { {"cnd_e32", k_cnd_e32, 4}, {"cnd_e64", k_cnd_e64, 4}, {"mad64_cnd_e32", k_mad64_cnd_e32, 4}, {"mad64_cnd_e64", k_mad64_cnd_e64, 4}, {"mad64_only", k_mad64_only, 4}, {"cmp_cnd_e64", k_cmp_cnd_e64, 4}, {"bfi", k_bfi, 4}, {"mad64_bfi", k_mad64_bfi, 4}, {"dppmov_cnd_e32", k_dppmov_cnd_e32, 4}, {"cnd_dpp", k_cnd_dpp, 4}, {"mad_4_8_4", k_mad_4_8_4, 8}, {"mad_4_8_5", k_mad_4_8_5, 8}, {"mad_4_8_6", k_mad_4_8_6, 8}, {"mad_4_8_7", k_mad_4_8_7, 8}, {"mad_4_8_8", k_mad_4_8_8, 8}, {"mad_4_8_9", k_mad_4_8_9, 8}, {"mad_4_8_10", k_mad_4_8_10, 8}, {"mad_4_8_11", k_mad_4_8_11, 8}, {"mad_4_8_12", k_mad_4_8_12, 8}, {"mad_4_8_13", k_mad_4_8_13, 8}, {"mad_4_8_14", k_mad_4_8_14, 8}, {"mad_4_8_15", k_mad_4_8_15, 8}, {"mad_4_8_16", k_mad_4_8_16, 8}, {"mad_4_4_12", k_mad_4_4_12, 8}, {"mad_4_5_12", k_mad_4_5_12, 8}, {"mad_4_6_12", k_mad_4_6_12, 8}, {"mad_4_7_12", k_mad_4_7_12, 8}, {"mad_4_8_12b", k_mad_4_8_12b, 8}, {"mad_d20", k_mad_d20, 8}, {"mad_d21", k_mad_d21, 8}, {"mad_d22", k_mad_d22, 8}, {"mad_d23", k_mad_d23, 8}, {"mad_d24", k_mad_d24, 8}, {"mad_d28", k_mad_d28, 8}, {"mad_d32", k_mad_d32, 8}, {"med3_4_8_12", k_med3_4_8_12, 8}, {"med3_4_5_6", k_med3_4_5_6, 8}, {"add_4_8", k_add_4_8, 8}, {"add_4_5", k_add_4_5, 8}, {"perm_4_8_12", k_perm_4_8_12, 8}, {"perm_4_5_6", k_perm_4_5_6, 8}, {"dep_chain_a", k_dep_chain_a, 4}, {"dep_chain_b", k_dep_chain_b, 4} };
  for (auto &t : tests) {
    uint64_t best = ~0ull;
    for (int rep = 0; rep < 4; rep++) {
      hipLaunchKernelGGL(t.fn, dim3(1), dim3(64), 0, 0, d_out, d_sink, 7);
      if (hipDeviceSynchronize() != hipSuccess) { printf("%s HIP ERROR\n", t.name); return 1; }
      uint64_t h = 0;
      if (hipMemcpy(&h, d_out, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
      if (h < best) best = h;
    }
    printf("%-18s %8.2f\n", t.name, (double)best / (400.0 * 8 * t.per)); fflush(stdout);
  }
  return 0;
}
