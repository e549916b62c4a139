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


/* context pool of the legacy API (aad_legacy_api.c) */
void AADHipInternal_ContextOptionsFromEnvironment(struct AADHipContext *context);
int32_t AADHipInternal_ContextDevice(const struct AADHipContext *context);

/* The legacy API's two data paths (one stream per call, planar int32 rows as the reference's
 * callers hold them, src/main.c:175-179 and :122-126): the int16 <-> int32 conversion reads / writes
 * the pinned staging block directly.  `state`: the handle's carried lanes (never NULL). */
AADApiResult AADHipInternal_EncodePlanar32(struct AADHipContext *context, const struct AADEncodeParameter *parameter,
                                           const int32_t *const *input, uint32_t num_samples, uint8_t *data,
                                           uint64_t data_capacity, uint64_t *output_size, struct AADHipLaneState *state);
AADApiResult AADHipInternal_DecodePlanar32(struct AADHipContext *context, const struct AADHeaderInfo *format,
                                           int32_t has_file_header, const uint8_t *data, uint64_t data_size,
                                           uint32_t want_frames, int32_t *const *buffer, uint32_t *decoded_frames);

#ifdef __cplusplus
}
#endif
#endif
