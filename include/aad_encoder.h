/* aad_encoder.h - kept so that `#include "aad_encoder.h"` of code written against the reference (src/aad_encoder.h) keeps
 * working; everything is declared in aad_api.h. */
#ifndef AAD_ENCODER_H_INCLDED
#define AAD_ENCODER_H_INCLDED
#include "aad_api.h"
#endif
