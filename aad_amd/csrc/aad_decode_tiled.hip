/*
 * aad_decode_tiled.hip - translation unit of the sector-tiled dense decoder (aad_decode_tiled.hip.h).
 */
#include "aad_decode_tiled.hip.h"
#include "aad_decode_tiled_launch.h"
#include "aad_launch.h"

namespace aad {

bool decode_tiled_applicable(const DecodeArgs &a)
{
  if (a.channels < 1 || a.channels > 2) return false;
  if (a.bits < 2 || a.bits > 4) return false;
  if (!a.pcm_aligned16) return false;
  {
    /* A block whose header asks for more samples than its block_size holds reads on into the bytes behind it (the reference's
     * code walk has no bound, src/aad_decoder.c:396-451; no encoder writes such a header).  The rows' rings are laid out for
     * blocks that keep to themselves: those streams take the per-lane kernel, which reads through to the end of the stream. */
    const uint64_t us = a.bits == 3 ? 8u : (a.bits == 4 ? 2u : 4u), ub = (uint64_t)(a.bits == 3 ? 3u : 1u) * a.channels;
    const uint64_t coded = a.samples_per_block > 4 ? a.samples_per_block - 4 : 0;
    if ((uint64_t)kBlockHeaderBytesPerCh * a.channels + (coded + us - 1) / us * ub > a.block_size) return false;
  }
  /* every block of a stream starts on a piece boundary: the block length in PCM bytes is a multiple of 16 (mono 2-bit
   * blocks of 1024 bytes hold 4028 samples = 8056 bytes: 8 mod 16, which the kernel takes with a short lead chunk) - or no stream has a second block */
  const uint64_t block_pcm_bytes = (uint64_t)a.samples_per_block * a.channels * 2u;
  const bool odd8_ok = a.bits == 2 && a.channels == 1 && block_pcm_bytes % 16u == 8u; /* DecodeTile::kOdd8: a short lead chunk for every second block */
  if (block_pcm_bytes % 16u != 0 && !odd8_ok && a.total_blocks > a.num_streams) return false;
  if ((reinterpret_cast<uintptr_t>(a.pcm) & 15u) != 0) return false;
  /* 3-bit rows: the code bytes of every block at the same offset inside their granule (aad_decode_tiled.hip.h "3-bit rows") */
  if (a.bits == 3 && !(a.code_phase_uniform == 1 || (a.code_phase_uniform == 2 && a.total_blocks <= a.num_streams))) return false;
  return true;
}

template <int BITS>
static void launch_bits(const DecodeArgs &a, dim3 grid, dim3 block, hipStream_t stream)
{
  if (a.channels == 1)
    AAD_LAUNCH((decode_tiled_kernel<BITS, 1, false>), grid, block, 0, stream, a);
  else if (a.mid_side)
    AAD_LAUNCH((decode_tiled_kernel<BITS, 2, true>), grid, block, 0, stream, a);
  else
    AAD_LAUNCH((decode_tiled_kernel<BITS, 2, false>), grid, block, 0, stream, a);
}

bool launch_decode_tiled(const DecodeArgs &a, hipStream_t stream)
{
  if (!decode_tiled_applicable(a)) return false;
  const uint64_t lanes = a.total_blocks * a.channels;
  const unsigned wg = 64u * DecodeTile<4, 1>::kWaves;
  const dim3 grid((unsigned)((lanes + wg - 1) / wg)), block(wg);
  if (a.bits == 4) launch_bits<4>(a, grid, block, stream);
  else if (a.bits == 3) launch_bits<3>(a, grid, block, stream);
  else launch_bits<2>(a, grid, block, stream);
  return true;
}

} /* namespace aad */
