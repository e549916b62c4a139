/*
 * aad_decode_split_launch.h - host entry of the split decoder's translation unit
 * (aad_decode_split.hip).  That unit is compiled with its own instruction-scheduling strategy
 * (see the Makefile), which is why its kernels are not launched from aad_hip_engine.hip directly.
 */
#ifndef AAD_DECODE_SPLIT_LAUNCH_H
#define AAD_DECODE_SPLIT_LAUNCH_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace aad {
struct DecodeArgs;
/* true when the batch is small enough, and its blocks short enough, for the residuals to stay in LDS */
bool decode_split_fits_lds(const DecodeArgs &args);
/* decode_split_kernel<bits, channels (1 or 2), mid_side, residuals in LDS> over ceil(recurrences / 16)
 * workgroups of 1024 threads.  residual == nullptr selects the LDS form (decode_split_fits_lds must
 * hold), otherwise the rows go through that device buffer.  false when nothing was launched. */
bool launch_decode_split(const DecodeArgs &args, int32_t *residual, uint32_t residual_stride, hipStream_t stream);
} /* namespace aad */

#endif /* AAD_DECODE_SPLIT_LAUNCH_H */
