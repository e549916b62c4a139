/*
 * aad_decode_split.hip - translation unit of the split (two-strand) quad decoder.
 *
 * Separate from aad_hip_engine.hip only because of one compiler flag: with LLVM's
 * "iterative-ilp" machine-scheduling strategy the prediction recurrence's chunk body comes out
 * 7 % faster on gfx950 (0.0452 -> 0.0420 ms for the bench batch, same box, bit-identical output;
 * the strategy alternates the two independent strands of a sample instead of emitting them one
 * after the other, and a lone wave pays an extra cycle for every instruction that reads the
 * result of the one just before it).  The same flag makes no difference to the encoder and is
 * not applied to the other kernels.
 */
#include <cstring>

#include "aad_decode_split.hip.h"
#include "aad_decode_split_launch.h"
#include "aad_launch.h"

namespace aad {

template <int BITS, bool LDSRES>
static void launch_bits(const SplitDecodeArgs &sa, dim3 grid, dim3 block, hipStream_t stream)
{
  if (sa.d.channels == 1)
    AAD_LAUNCH((decode_split_kernel<BITS, 1, false, LDSRES>), grid, block, 0, stream, sa);
  else if (sa.d.mid_side)
    AAD_LAUNCH((decode_split_kernel<BITS, 2, true, LDSRES>), grid, block, 0, stream, sa);
  else
    AAD_LAUNCH((decode_split_kernel<BITS, 2, false, LDSRES>), grid, block, 0, stream, sa);
}

bool decode_split_fits_lds(const DecodeArgs &args)
{
  const uint32_t coded = args.samples_per_block > 4 ? args.samples_per_block - 4 : 0;
  return coded <= kLdsResidualMax && args.total_blocks * args.channels <= 4096; /* one workgroup per CU */
}

bool launch_decode_split(const DecodeArgs &args, int32_t *residual, uint32_t residual_stride, hipStream_t stream)
{
  if (args.channels < 1 || args.channels > 2) return false;
  SplitDecodeArgs sa;
  sa.d = args;
  sa.residual = residual;
  sa.residual_stride = residual_stride;
  sa.reserved = 0;
  const uint64_t recurrences = args.total_blocks * args.channels;
  const dim3 grid((unsigned)((recurrences + 15) / 16)), block(1024); /* 16 recurrences per workgroup */
  const bool lds = residual == nullptr;
  if (lds && !decode_split_fits_lds(args)) return false;
  switch (args.bits) {
    case 4: lds ? launch_bits<4, true>(sa, grid, block, stream) : launch_bits<4, false>(sa, grid, block, stream); return true;
    case 3: lds ? launch_bits<3, true>(sa, grid, block, stream) : launch_bits<3, false>(sa, grid, block, stream); return true;
    case 2: lds ? launch_bits<2, true>(sa, grid, block, stream) : launch_bits<2, false>(sa, grid, block, stream); return true;
    default: return false;
  }
}

} /* namespace aad */

#if AAD_PHASE_TIMING
/* measurement builds only: copy out and reset the phase log of this unit's kernels
 * (kernel entry | tables written | barrier | strand 1 done | barrier | header parsed, first frames out |
 *  first loads + prime | chunk loop | tail) */
extern "C" uint32_t AADHipDebug_ReadSplitPhaseTimes(uint64_t *out, uint32_t capacity)
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
