/* aad.h - kept so that `#include "aad.h"` of code written against the reference (src/aad.h) keeps
 * working; everything is declared in aad_api.h. */
#ifndef AAD_H_INCLDED
#define AAD_H_INCLDED
#include "aad_api.h"
#endif
