/* Exhaustive proof that the device quantiser equals the reference's integer division
 * (reference src/aad_encoder.c:372):  min((|d| << (bits-2)) / step, magmax)
 *   ==  min(trunc(fmaf((float)|d|, 2^(bits-1) * hr, hr)), magmax),  hr = fl32(0.5 / step)
 * for every table step, every bit width and every reachable |d| (|x - p| <= 32768 + 65536).
 * fmaf is correctly rounded in C (like v_fma_f32), the cast truncates (like v_cvt_u32_f32). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../aad_amd/csrc/aad_tables_data.h"

static const uint16_t T[256] = {AAD_STEP_TABLE_VALUES};
static const uint32_t HB[256] = {AAD_HALF_RECIP_BITS};

int main(void)
{
  long bad = 0, checked = 0;
  for (int i = 0; i < 256; i++) { /* the table itself: hr must be the correctly rounded 0.5/step */
    float want = (float)(0.5 / (double)T[i]), got;
    memcpy(&got, &HB[i], 4);
    if (want != got) bad++;
  }
  for (int bits = 2; bits <= 4; bits++) {
    const uint32_t magmax = (1u << (bits - 1)) - 1u;
    for (int i = 0; i < 256; i++) {
      float hr;
      memcpy(&hr, &HB[i], 4);
      const float hs = hr * (float)(1 << (bits - 1));
      for (uint32_t a = 0; a <= 98304u + 16u; a++) {
        uint32_t want = (a << (bits - 2)) / T[i];
        if (want > magmax) want = magmax;
        uint32_t got = (uint32_t)fmaf((float)a, hs, hr);
        if (got > magmax) got = magmax;
        bad += want != got;
        checked++;
      }
    }
  }
  printf("%ld %ld\n", checked, bad);
  return bad != 0;
}
