/*
 * aad_synth.c - the integer-only synthetic PCM corpus of SURVEY.md section 8d ("Synthetic PCM
 * generator") in C.  Same arithmetic, sample for sample, as aad_amd/synth.py (which stays the
 * specification and the fallback; tests/test_host_api.py holds the two against each other): every
 * (stream, channel) owns a xorshift64 generator, a sample is two triangle partials plus
 * triangular-ish noise clipped to int16.  Exists because a single long stream (BASELINE config 2
 * form iii: 1 stream x 1000 blocks) costs the numpy version one interpreter round per sample.
 * Host C, no GPU; measurement / test input only - nothing on the codec path calls it.
 */
#include "../../include/aad_synth.h"

static uint64_t xorshift(uint64_t x)
{
  x ^= x << 13;
  x ^= x >> 7;
  x ^= x << 17;
  return x;
}

static int64_t triangle(uint64_t phase) /* 16-bit triangle from a 32-bit phase accumulator */
{
  const int64_t t = (int64_t)(phase >> 16);
  return 2 * (t < 32768 ? t : 65535 - t) - 32767;
}

static int64_t asr(int64_t v, int s) { return v >= 0 ? v >> s : -((-v + (((int64_t)1 << s) - 1)) >> s); } /* floor */

int32_t AADSynth_Generate(int16_t *pcm, uint64_t num_streams, uint64_t num_samples, uint32_t channels,
                          uint64_t seed, uint32_t rate, int32_t kind, uint64_t first_stream)
{
  uint64_t s, n;
  uint32_t c;
  if (pcm == 0 || channels == 0 || rate == 0 || kind < AAD_SYNTH_MUSIC || kind > AAD_SYNTH_NYQUIST) return -1;
  for (s = 0; s < num_streams; s++) {
    for (c = 0; c < channels; c++) {
      uint64_t x = (0x9E3779B97F4A7C15ull * (seed + 1)) ^ (((first_stream + s) << 8) | c);
      uint64_t f1, f2, ph1, ph2, inc1, inc2;
      int16_t *out = pcm + s * num_samples * channels + c;
      int i;
      if (x == 0) x = 1;
      for (i = 0; i < 5; i++) x = xorshift(x);
      f1 = 100 + x % 901;
      x = xorshift(x);
      f2 = 1000 + x % 5001;
      x = xorshift(x);
      ph1 = x & 0xFFFFFFFFull;
      x = xorshift(x);
      ph2 = x & 0xFFFFFFFFull;
      inc1 = (f1 << 32) / rate;
      inc2 = (f2 << 32) / rate;
      for (n = 0; n < num_samples; n++, out += channels) {
        int64_t v;
        x = xorshift(x);
        if (kind == AAD_SYNTH_MUSIC) {
          const int64_t u = (int64_t)((x & 0xFFFF) + ((x >> 16) & 0xFFFF) + ((x >> 32) & 0xFFFF));
          v = asr(triangle(ph1) * 11469, 15) + asr(triangle(ph2) * 6554, 15) + asr((u - 98304) * 437, 14);
          ph1 = (ph1 + inc1) & 0xFFFFFFFFull;
          ph2 = (ph2 + inc2) & 0xFFFFFFFFull;
        } else if (kind == AAD_SYNTH_NOISE) {
          v = (int64_t)(x & 0xFFFF) - 32768;
        } else {
          v = (n & 1) == 0 ? 32767 : -32768;
        }
        *out = (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
      }
    }
  }
  return 0;
}
