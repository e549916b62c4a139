/* aad_hip_internal.h - library-private entry points shared between the engine and the legacy API. */
#ifndef AAD_HIP_INTERNAL_H
#define AAD_HIP_INTERNAL_H

#include "../../include/aad_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Host-memory decode of num_streams images of one `format`; has_file_header = 0 decodes bare
 * blocks (AADDecoder_DecodeBlock).  num_samples[i] frames are requested per stream; decoded_frames
 * (may be NULL) receives how many were produced. */
AADApiResult AADHipInternal_DecodeHost(struct AADHipContext *context, const struct AADHeaderInfo *format,
                                       int32_t has_file_header, uint32_t num_streams,
                                       const uint8_t *const *data, const uint64_t *data_size,
                                       const uint32_t *num_samples, int16_t *const *pcm, uint32_t *decoded_frames);

#ifdef __cplusplus
}
#endif
#endif
