/* aad_decoder.h - kept so that `#include "aad_decoder.h"` of code written against the reference (src/aad_decoder.h) keeps
 * working; everything is declared in aad_api.h. */
#ifndef AAD_DECODER_H_INCLDED
#define AAD_DECODER_H_INCLDED
#include "aad_api.h"
#endif
