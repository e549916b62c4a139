/*
 * aad_decode_tiled_launch.h - host entry of the sector-tiled dense decoder's translation unit
 * (aad_decode_tiled.hip; a unit of its own so that the kernels compile beside aad_hip_engine.hip's).
 */
#ifndef AAD_DECODE_TILED_LAUNCH_H
#define AAD_DECODE_TILED_LAUNCH_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace aad {
struct DecodeArgs;
/* true when decode_tiled_kernel can decode this plan: mono / stereo, 3-bit codes only at a batch-wide code phase (DecodeArgs::code_phase_uniform), every block's PCM 16-byte
 * aligned (DecodeArgs::pcm_aligned16 from the stream table, the base pointer checked here) */
bool decode_tiled_applicable(const DecodeArgs &args);
/* decode_tiled_kernel<bits, channels, mid_side> over ceil(recurrences / 256) workgroups of 256 threads.
 * false when nothing was launched (not applicable). */
bool launch_decode_tiled(const DecodeArgs &args, hipStream_t stream);
} /* namespace aad */

#endif /* AAD_DECODE_TILED_LAUNCH_H */
