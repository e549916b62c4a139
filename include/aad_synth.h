/*
 * aad_synth.h - the synthetic PCM corpus generator of SURVEY.md section 8d, C form.
 *
 * Not part of the reference's API (the reference has no generator: its tests use sin/rand in
 * test/test_aad_encode_decode.c:430-470 and WAV files); measurement and test input only.  Integer
 * arithmetic throughout, so the GPU box rebuilds exactly the inputs the reference was run on in
 * the build container.  aad_amd/synth.py is the specification.
 */
#ifndef AAD_SYNTH_H_INCLUDED
#define AAD_SYNTH_H_INCLUDED

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum AADSynthKind {
  AAD_SYNTH_MUSIC = 0,  /* two triangle partials + noise, |x| <~ 0.63 full scale */
  AAD_SYNTH_NOISE = 1,  /* full-scale white noise   (reference test/test_aad_encode_decode.c:447-451) */
  AAD_SYNTH_NYQUIST = 2 /* full-scale square at fs/2 (reference test/test_aad_encode_decode.c:467-470) */
};

/* pcm: num_streams * num_samples * channels int16, stream-major, channel-interleaved frames.
 * Stream i of the call is corpus stream first_stream + i.  Returns 0, or -1 on a bad argument. */
int32_t AADSynth_Generate(int16_t *pcm, uint64_t num_streams, uint64_t num_samples, uint32_t channels,
                          uint64_t seed, uint32_t rate, int32_t kind, uint64_t first_stream);

#ifdef __cplusplus
}
#endif
#endif
