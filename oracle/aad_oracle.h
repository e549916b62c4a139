/*
 * aad_oracle.h - CPU restatement of the AAD (codec v18 / format v4) encode/decode path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle: nothing in the product
 * (aad_amd/, include/) may include, link or call it.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
 * checker / as the timed CPU baseline - never as the thing shipped.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file
 * byte-for-byte against the reference's own fixtures (sin300Hz*.aad,
 * sin300Hz*_decoded.wav, geometry known-answers) and against golden vectors
 * generated from the compiled reference (oracle/_ref, tests/golden/make_golden.py).
 *
 * Every function names the reference lines it restates.  The layout differs
 * from the reference on purpose: PCM is int16, channel-interleaved (the device
 * layout), channel state is a flat "lane" record, and any channel count up to
 * AADO_MAX_CHANNELS is accepted (the reference stops at 2, src/aad.h:13).
 */
#ifndef AAD_ORACLE_H
#define AAD_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AADO_MAX_CHANNELS 8
#define AADO_FILE_HEADER_BYTES 31
#define AADO_TAPS 4

/* return codes: same numbering as the reference's AADApiResult (src/aad.h:25-33) */
enum {
  AADO_OK = 0,
  AADO_INVALID_ARGUMENT = 1,
  AADO_INVALID_FORMAT = 2,
  AADO_INSUFFICIENT_BUFFER = 3,
  AADO_INSUFFICIENT_DATA = 4,
  AADO_PARAMETER_NOT_SET = 5,
  AADO_NG = 6
};

/* Per-channel predictor state (reference src/aad_encoder.c:10-15, src/aad_decoder.c:9-13). */
typedef struct {
  int32_t w[AADO_TAPS]; /* Q15 LMS weights */
  int32_t h[AADO_TAPS]; /* history, h[0] newest, int16 range */
  int32_t idx;          /* Q4 step index, 0..4080 */
  int32_t qerr;         /* last dequantised difference (encoder only) */
} AadoLane;

typedef struct {
  uint32_t format_version, codec_version;
  uint32_t num_channels, num_samples, sampling_rate, bits_per_sample;
  uint32_t block_size, samples_per_block, ch_process_method;
} AadoHeader;

/* the 256-entry step table and the per-bit-width index deltas (read-only views) */
const uint16_t *aado_step_table(void);
const int16_t *aado_index_deltas(uint32_t bits); /* 1 << (bits-1) entries */

/* geometry: reference src/aad_encoder.c:85-131 */
int aado_block_geometry(uint32_t max_block_size, uint32_t channels, uint32_t bits,
                        uint32_t *block_size, uint32_t *samples_per_block);

/* 31-byte big-endian file header: reference src/aad_encoder.c:134-221, src/aad_decoder.c:99-225 */
int aado_put_header(const AadoHeader *hd, uint8_t *out, size_t cap);
int aado_get_header(const uint8_t *data, size_t size, AadoHeader *hd);
int aado_check_header(const AadoHeader *hd, uint32_t max_channels);

/* bytes an encoded stream occupies (file header + all blocks, short tail included) */
size_t aado_encoded_size(uint32_t num_samples, uint32_t channels, uint32_t bits, uint32_t max_block_size);

/*
 * Encode one stream.  pcm: num_samples frames of `channels` interleaved int16.
 * lanes[channels] carries the predictor state in and out (zero it for a fresh
 * encoder; the reference keeps it across EncodeWhole calls, src/aad_encoder.c:853-886).
 * reset_idx != 0 zeroes the step index first (what SetEncodeParameter does,
 * src/aad_encoder.c:797-799).  Restates src/aad_encoder.c:343-727, 814-891.
 */
int aado_encode_stream(const int16_t *pcm, uint32_t num_samples, uint32_t channels,
                       uint32_t sampling_rate, uint32_t bits, uint32_t max_block_size,
                       uint32_t ch_process_method, uint32_t trials, AadoLane *lanes,
                       uint8_t *out, size_t cap, size_t *out_size);

/*
 * Decode one stream into interleaved int16 (values the reference returns as
 * int32 are always in int16 range, src/aad_decoder.c:298-300, 467-468).
 * pcm_frames is the capacity in frames; the decode covers hd.num_samples.
 * Bytes past `size` read as zero (the reference reads out of bounds there).
 * Restates src/aad_decoder.c:269-538.
 */
int aado_decode_stream(const uint8_t *data, size_t size, uint32_t max_channels,
                       int16_t *pcm, uint32_t pcm_frames, AadoHeader *hd_out);

/* Decode a single block given a header (reference src/aad_decoder.c:321-475). */
int aado_decode_block(const AadoHeader *hd, const uint8_t *block, size_t size,
                      int16_t *pcm, uint32_t want_frames, uint32_t *got_frames);

/* single steps, exported for white-box tests and the quantiser-equivalence check */
uint32_t aado_encode_step(AadoLane *lane, int32_t x, uint32_t bits);
int32_t aado_decode_step(AadoLane *lane, uint32_t code, uint32_t bits);

/* batch helpers used by bench.py's cpu_baseline leg: n equal-shaped streams back to back */
int aado_encode_batch(const int16_t *pcm, uint32_t num_streams, uint32_t num_samples,
                      uint32_t channels, uint32_t sampling_rate, uint32_t bits,
                      uint32_t max_block_size, uint32_t ch_process_method, uint32_t trials,
                      uint8_t *out, size_t out_stride);
int aado_decode_batch(const uint8_t *data, uint32_t num_streams, size_t stride, size_t size,
                      int16_t *pcm, uint32_t pcm_frames);

/*
 * The reference CLI's reconstruction modes (src/main.c:275-503): x = original, y = decode(encode(x)),
 * both interleaved int16, `count` values in total.
 *   aado_residual     what `aad -g` writes: the CLI subtracts in the WAV reader's 32-bit domain
 *                     ((x << 16) - (y << 16), wrapping) and the 16-bit writer keeps the top half
 *                     (src/wav.c:429), i.e. the int16 wrap of x - y  (src/main.c:419-423).
 *   aado_error_stats  the three numbers `aad -c` prints as "RMSE:%f MSD:%f MaxAE:%f"
 *                     (src/main.c:476-497).  As written there, the first operand is the
 *                     RESIDUAL (the WAV buffer was overwritten at :470-474) scaled by 1/INT32_MAX
 *                     and the second the decoded int16 value scaled by 1/INT32_MAX without the
 *                     << 16; the oracle restates that arithmetic literally, in the reference's
 *                     summation order (channel-major) when channels is given.
 */
void aado_residual(const int16_t *x, const int16_t *y, size_t count, int16_t *out);
void aado_error_stats(const int16_t *x, const int16_t *y, uint32_t num_samples, uint32_t channels,
                      double out_rmse_msd_maxae[3]);

#ifdef __cplusplus
}
#endif
#endif
