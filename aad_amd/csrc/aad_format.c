/* aad_format.c - see aad_format.h.  Host C only. */
#include "aad_format.h"

static uint32_t gcd32(uint32_t a, uint32_t b)
{
  while (b != 0) {
    uint32_t r = a % b;
    a = b;
    b = r;
  }
  return a;
}

struct AADPackUnit AADFormat_PackUnit(uint32_t bits)
{
  struct AADPackUnit u;
  const uint32_t unit_bits = 8u * bits / gcd32(8u, bits);
  u.bytes_per_channel = unit_bits / 8u;
  u.samples = unit_bits / bits;
  return u;
}

AADApiResult AADFormat_BlockGeometry(uint32_t max_block_size, uint32_t num_channels, uint32_t bits,
                                     uint32_t max_channels, uint16_t *block_size, uint32_t *samples_per_block)
{
  struct AADPackUnit u;
  uint32_t head, stride, units;
  if (block_size == NULL) return AAD_APIRESULT_INVALID_ARGUMENT;
  if (num_channels == 0 || num_channels > max_channels || bits == 0 || bits > AAD_MAX_BITS_PER_SAMPLE)
    return AAD_APIRESULT_INVALID_FORMAT;
  head = AAD_BLOCK_HEADER_BYTES_PER_CH * num_channels;
  if (max_block_size < head) return AAD_APIRESULT_INVALID_FORMAT;
  u = AADFormat_PackUnit(bits);
  stride = u.bytes_per_channel * num_channels;
  units = (max_block_size - head) / stride;
  *block_size = (uint16_t)(head + units * stride);
  if (samples_per_block != NULL) *samples_per_block = AAD_NUM_TAPS + units * u.samples;
  return AAD_APIRESULT_OK;
}

int AADFormat_HeaderFieldsValid(const struct AADHeaderInfo *h, uint32_t max_channels)
{
  if (h->num_channels == 0 || h->num_channels > max_channels) return 0;
  if (h->num_samples == 0) return 0;
  if (h->sampling_rate == 0) return 0;
  if (h->bits_per_sample < AAD_MIN_BITS_PER_SAMPLE || h->bits_per_sample > AAD_MAX_BITS_PER_SAMPLE) return 0;
  if (h->block_size <= AAD_BLOCK_HEADER_BYTES_PER_CH * (uint32_t)h->num_channels) return 0;
  if (h->num_samples_per_block == 0) return 0;
  if ((uint32_t)h->ch_process_method >= (uint32_t)AAD_CH_PROCESS_METHOD_INVALID) return 0;
  if (h->ch_process_method == AAD_CH_PROCESS_METHOD_MS && h->num_channels == 1) return 0;
  return 1;
}

int AADFormat_HeaderAcceptedByDecoder(const struct AADHeaderInfo *h, uint32_t max_channels)
{
  if (h->format_version != AAD_FORMAT_VERSION) return 0;
  if (h->codec_version != AAD_CODEC_VERSION) return 0;
  return AADFormat_HeaderFieldsValid(h, max_channels);
}

static uint8_t *store_be(uint8_t *p, uint32_t v, uint32_t nbytes)
{
  while (nbytes-- > 0) *p++ = (uint8_t)(v >> (8u * nbytes));
  return p;
}

static uint32_t load_be(const uint8_t *p, uint32_t nbytes)
{
  uint32_t v = 0;
  while (nbytes-- > 0) v = (v << 8) | *p++;
  return v;
}

void AADFormat_PutHeader(const struct AADHeaderInfo *h, uint8_t *d)
{
  d[0] = 'A';
  d[1] = 'A';
  d[2] = 'D';
  d[3] = 0;
  d = store_be(d + 4, AAD_FORMAT_VERSION, 4); /* macros, not the struct fields: src/aad_encoder.c:195-200 */
  d = store_be(d, AAD_CODEC_VERSION, 4);
  d = store_be(d, h->num_channels, 2);
  d = store_be(d, h->num_samples, 4);
  d = store_be(d, h->sampling_rate, 4);
  d = store_be(d, h->bits_per_sample, 2);
  d = store_be(d, h->block_size, 2);
  d = store_be(d, h->num_samples_per_block, 4);
  (void)store_be(d, (uint32_t)h->ch_process_method, 1);
}

int AADFormat_GetHeader(const uint8_t *d, struct AADHeaderInfo *h)
{
  if (d[0] != 'A' || d[1] != 'A' || d[2] != 'D' || d[3] != 0) return 0;
  h->format_version = load_be(d + 4, 4);
  h->codec_version = load_be(d + 8, 4);
  h->num_channels = (uint16_t)load_be(d + 12, 2);
  h->num_samples = load_be(d + 14, 4);
  h->sampling_rate = load_be(d + 18, 4);
  h->bits_per_sample = (uint16_t)load_be(d + 22, 2);
  h->block_size = (uint16_t)load_be(d + 24, 2);
  h->num_samples_per_block = load_be(d + 26, 4);
  h->ch_process_method = (AADChannelProcessMethod)d[30];
  return 1;
}

AADApiResult AADFormat_ParameterToHeader(const struct AADEncodeParameter *p, uint32_t num_samples,
                                         uint32_t max_channels, struct AADHeaderInfo *header)
{
  struct AADHeaderInfo h;
  if (p == NULL || header == NULL) return AAD_APIRESULT_INVALID_ARGUMENT;
  if (p->bits_per_sample == 0 || p->bits_per_sample > AAD_MAX_BITS_PER_SAMPLE) return AAD_APIRESULT_INVALID_FORMAT;
  if (p->max_block_size < AAD_BLOCK_HEADER_BYTES_PER_CH * (uint32_t)p->num_channels) return AAD_APIRESULT_INVALID_FORMAT;
  if ((uint32_t)p->ch_process_method >= (uint32_t)AAD_CH_PROCESS_METHOD_INVALID) return AAD_APIRESULT_INVALID_FORMAT;
  h.format_version = AAD_FORMAT_VERSION; /* what the encoder writes, src/aad_encoder.c:195-200 */
  h.codec_version = AAD_CODEC_VERSION;
  h.num_channels = p->num_channels;
  h.num_samples = num_samples;
  h.sampling_rate = p->sampling_rate;
  h.bits_per_sample = p->bits_per_sample;
  h.ch_process_method = p->ch_process_method;
  if (AADFormat_BlockGeometry(p->max_block_size, p->num_channels, p->bits_per_sample, max_channels,
                              &h.block_size, &h.num_samples_per_block) != AAD_APIRESULT_OK)
    return AAD_APIRESULT_INVALID_FORMAT;
  *header = h;
  return AAD_APIRESULT_OK;
}

uint32_t AADFormat_BlockBytes(uint32_t n, uint32_t num_channels, uint32_t bits)
{
  const struct AADPackUnit u = AADFormat_PackUnit(bits);
  const uint32_t units = n > AAD_NUM_TAPS ? (n - AAD_NUM_TAPS + u.samples - 1) / u.samples : 0;
  return (AAD_BLOCK_HEADER_BYTES_PER_CH + units * u.bytes_per_channel) * num_channels;
}

uint64_t AADFormat_EncodedSize(const struct AADHeaderInfo *h)
{
  const uint32_t spb = h->num_samples_per_block;
  const uint64_t full = h->num_samples / spb;
  const uint32_t tail = h->num_samples % spb;
  uint64_t size = AAD_HEADER_SIZE + full * AADFormat_BlockBytes(spb, h->num_channels, h->bits_per_sample);
  if (tail != 0) size += AADFormat_BlockBytes(tail, h->num_channels, h->bits_per_sample);
  return size;
}

int AADFormat_DecodeWorkBounded(const struct AADHeaderInfo *h, uint32_t num_samples)
{
  const uint32_t per_block = num_samples < h->num_samples_per_block ? num_samples : h->num_samples_per_block;
  return per_block < 0x80000000u;
}
