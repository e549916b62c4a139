#!/usr/bin/env python3
"""Generate aad_amd/csrc/aad_encode_chunk_asm.hip.h: the 4-bit quad-mapping encoder's 16-sample
chunk body as ONE hand-scheduled gfx950 instruction sequence (inline asm).

Why not leave it to the compiler: a lone wave pays 4 cycles per issued instruction plus one for
every instruction that reads the result of the instruction just before it, s_nop and s_waitcnt
included.  The C++ body (encode_chunk16_quad) comes out at 36 slots per sample with ~17 such
back-to-back pairs; the schedule below has 32 slots and 8 pairs:

  * after the quantiser (fma, cvt, min) the sample splits into two chains that need nothing but
    the magnitude - step index -> LDS record of the next sample, and dequantise -> reconstruct ->
    LMS - which are emitted strictly alternating, the index chain one step ahead so that its
    ds_read_b96 goes out as early as possible;
  * both code-packing instructions sit in the two wait states the first DPP butterfly add
    needs, the NEXT sample's history shift (v_and_b32_dpp) in those of the second, leaving one
    s_nop per sample.

The arithmetic is instruction for instruction what encode_chunk16_quad<4, true> computes (same
operations, same operand order); parity is enforced by the GPU suite (tests/test_gpu_parity.py
runs every golden case through the quad mapping).
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "aad_amd", "csrc", "aad_encode_chunk_asm.hip.h")

DPP_SWAP1 = "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"
DPP_SWAP2 = "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1"
DPP_SHIFT = "quad_perm:[0,0,1,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"


def sample(j):
    acc = "acc0" if j < 8 else "acc1"
    xnext = "x%d" % (j + 1)  # x16 = first sample of the next chunk
    last = j == 15
    L = [
        "s_waitcnt lgkmcnt(0)",
        "v_fma_f32 %[t0], |%[f]|, %[e2], %[e1]",
        "v_cvt_u32_f32 %[t0], %[t0]",
        "v_min_u32 %[mag], 7, %[t0]",
        "v_or_b32 %[t1], 0x0c0c0c00, %[mag]",
        "v_lshl_add_u32 %[t3], 2, %[mag], %[idxb]",
        "v_perm_b32 %[t1], 0, %[lut], %[t1]",
        "v_lshl_or_b32 %[t2], %[mag], 29, %[k1shl28]",
        "v_sub_u32 %[t3], %[t3], %[t1]",
        "v_med3_i32 %[idxb], %[t3], 8, %[kidxmax]",
        "v_and_b32 %[t4], 0xff0, %[idxb]",
        "v_mul_hi_u32 %[q], %[t2], %[e0]",
        "ds_read_b96 %[e], %[t4] offset:%[wide]",
        "v_xor_b32 %[q], %[m], %[q]",
        "v_sub_u32 %[qd], %[q], %[m]",
        "v_add_u32 %[t0], %[qd], %[p]",
        "v_mad_i32_i24 %[lm], %[qd], %[h], %[k16384]",
        "v_med3_i32 %[y], %[t0], %[km32768], %[k32767]",
        "v_ashrrev_i32 %[lm], 18, %[lm]",
        "v_and_or_b32 %[h], %[y], %[newest], %[hs]",
        "v_add_u32 %[w], %[lm], %[w]",
        "v_mad_u64_u32 %[s], vcc, %[h], %[w], %[round]",
        "v_and_or_b32 %[t1], %[m], 8, %[mag]",
        "v_lshl_or_b32 %%[%s], %%[%s], 4, %%[t1]" % (acc, acc),
        "v_add_u32_dpp %[slo], %[slo], %[slo] " + DPP_SWAP1,
    ]
    if not last:
        L += ["v_and_b32_dpp %[hs], %[h], %[notnewest] " + DPP_SHIFT, "s_nop 0"]
    else:
        L += ["s_nop 1"]
    L += [
        "v_add_u32_dpp %[slo], %[slo], %[slo] " + DPP_SWAP2,
        "v_ashrrev_i32 %[p], 15, %[slo]",
        "v_sub_u32 %%[d], %%[%s], %%[p]" % xnext,
        "v_cvt_f32_i32 %[f], %[d]",
        "v_ashrrev_i32 %[m], 31, %[d]",
    ]
    return L


E0, E1, E2, S0 = "v250", "v251", "v252", "v248"   # fixed homes of the LDS record and the 64-bit product


def main():
    lines = ["v_mov_b32 %s, %%[e0]" % E0, "v_mov_b32 %s, %%[e1]" % E1, "v_mov_b32 %s, %%[e2]" % E2,
             "v_and_b32_dpp %[hs], %[h], %[notnewest] " + DPP_SHIFT]
    for j in range(16):
        lines += sample(j)
    lines += ["s_waitcnt lgkmcnt(0)", "v_mov_b32 %%[e0], %s" % E0, "v_mov_b32 %%[e1], %s" % E1, "v_mov_b32 %%[e2], %s" % E2]
    exp = os.environ.get("AAD_ASM_EXPERIMENT", "")   # destructive timing experiments (wrong output!)
    if exp == "nolds":
        lines = [l for l in lines if not l.startswith("ds_read") and not (l.startswith("s_waitcnt") )] + ["s_waitcnt lgkmcnt(0)"]
    if exp == "nonop":
        lines = [l for l in lines if not l.startswith("s_nop")]
    if exp == "nodpp":
        lines = [l for l in lines if "v_add_u32_dpp" not in l and not l.startswith("s_nop")]
    if exp == "nowait":
        lines = [l for l in lines if not l.startswith("s_waitcnt")] + ["s_waitcnt lgkmcnt(0)"]
    if exp == "half":
        lines = [l for k, l in enumerate(lines) if k % 2 == 0 or l.startswith("v_mov_b32 v25") or l.startswith("v_mov_b32 %[e") or l.startswith("ds_read") or l.startswith("s_waitcnt")]
    lines = [l.replace("%[e0]", E0).replace("%[e1]", E1).replace("%[e2]", E2) if not l.startswith("v_mov_b32") else l for l in lines]
    lines = [l.replace("%[e],", "v[250:252],").replace("%[slo]", S0).replace("%[s],", "v[248:249],") for l in lines]
    body = "\n".join('      "%s\\n"' % l for l in lines)
    xs_in = ", ".join('[x%d] "v"(x[%d])' % (j, j) for j in range(1, 16)) + ', [x16] "v"(xn0)'
    text = '''/* GENERATED by tools/gen_encode_chunk_asm.py - do not edit. */
#ifndef AAD_ENCODE_CHUNK_ASM_HIP_H
#define AAD_ENCODE_CHUNK_ASM_HIP_H

/* Sixteen steps of the 4-bit quad-mapping encoder (EMIT form), hand-scheduled: see the generator
 * for the why and the how.  Same contract as encode_chunk16_quad<4, true>: x[1..15] are this
 * chunk's samples after the one already in flight in C, xn0 the first sample of the next chunk,
 * w[0..1] the two big-endian code words, qd_out the last dequantised difference.  The LDS table
 * block must start at LDS address 0 (checked once by the kernel). */
__device__ __forceinline__ void encode_chunk16_quad_emit4_asm(QuadLane &L, EncodeCarry &C, const int32_t *x, int32_t xn0,
                                                              uint32_t *w, int32_t &qd_out)
{
  uint32_t t0, t1, t2, t3, t4, mag, hs;
  int32_t q, qd, lm, y;
  uint32_t e0 = C.e.x, e1 = C.e.y, e2 = C.e.z;
  const uint64_t round64 = (uint64_t)L.round;
  const uint32_t notnewest = ~L.newest;
  asm volatile(
%s
      : [w] "+v"(L.w), [h] "+v"(L.h), [idxb] "+v"(L.idxb), [e0] "+v"(e0), [e1] "+v"(e1), [e2] "+v"(e2), [p] "+v"(C.p),
        [d] "+v"(C.d), [m] "+v"(C.m), [f] "+v"(C.f), [acc0] "+v"(w[0]), [acc1] "+v"(w[1]), [t0] "=&v"(t0), [t1] "=&v"(t1),
        [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [mag] "=&v"(mag), [hs] "=&v"(hs), [q] "=&v"(q), [qd] "=&v"(qd),
        [lm] "=&v"(lm), [y] "=&v"(y)
      : %s,
        [lut] "s"(0x00161514u), [k1shl28] "s"(1u << 28), [kidxmax] "v"((int32_t)kIdxMax), [k16384] "s"(16384),
        [km32768] "s"(-32768), [k32767] "v"(32767), [newest] "v"(L.newest), [notnewest] "v"(notnewest),
        [round] "v"(round64), [wide] "n"(kLdsWideOff)
      : "vcc", "memory", "v248", "v249", "v250", "v251", "v252");
  C.e.x = e0;
  C.e.y = e1;
  C.e.z = e2;
  qd_out = qd;
}

#endif /* AAD_ENCODE_CHUNK_ASM_HIP_H */
''' % (body, xs_in)
    # sub-register names of multi-dword operands: e0/e1/e2 of the 96-bit record, low half of s
    open(OUT, "w").write(text)
    print("wrote", OUT, len(lines), "instructions")


if __name__ == "__main__":
    main()
