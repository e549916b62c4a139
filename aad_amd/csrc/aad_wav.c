/* aad_wav.c - see include/aad_wav.h.  Host C, no GPU. */
#include "../../include/aad_wav.h"

#include <string.h>

static uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint16_t le16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static void put32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static void put16(uint8_t *p, uint16_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }

AADApiResult AADWav_ParseHeader(const uint8_t *data, uint64_t data_size, struct AADWavInfo *info)
{
  uint64_t pos = 12;
  int have_fmt = 0;
  if (data == NULL || info == NULL) return AAD_APIRESULT_INVALID_ARGUMENT;
  if (data_size < 12) return AAD_APIRESULT_INSUFFICIENT_DATA;
  if (memcmp(data, "RIFF", 4) != 0 || memcmp(data + 8, "WAVE", 4) != 0) return AAD_APIRESULT_INVALID_FORMAT;
  memset(info, 0, sizeof(*info));
  while (pos + 8 <= data_size) {
    const uint8_t *ck = data + pos;
    const uint64_t size = le32(ck + 4);
    if (memcmp(ck, "fmt ", 4) == 0) {
      if (size < 16 || pos + 8 + 16 > data_size) return AAD_APIRESULT_INSUFFICIENT_DATA;
      info->format_tag = le16(ck + 8);
      info->num_channels = le16(ck + 10);
      info->sampling_rate = le32(ck + 12);
      info->bits_per_sample = le16(ck + 22);
      have_fmt = 1;
    } else if (memcmp(ck, "data", 4) == 0) {
      uint64_t avail = data_size - (pos + 8);
      if (!have_fmt) return AAD_APIRESULT_INVALID_FORMAT;
      if (info->num_channels == 0 || info->bits_per_sample == 0 || (info->bits_per_sample & 7) != 0)
        return AAD_APIRESULT_INVALID_FORMAT;
      info->data_offset = pos + 8;
      info->data_size = size < avail ? size : avail; /* a short file keeps what is there */
      info->num_samples = (uint32_t)(info->data_size / ((uint64_t)info->num_channels * (info->bits_per_sample / 8)));
      return AAD_APIRESULT_OK;
    }
    pos += 8 + size; /* like the reference reader, no odd-size padding step (src/wav.c:176-193) */
  }
  return have_fmt ? AAD_APIRESULT_INSUFFICIENT_DATA : AAD_APIRESULT_INVALID_FORMAT;
}

AADApiResult AADWav_WriteHeader(uint8_t *d, uint32_t data_size, uint16_t num_channels, uint32_t sampling_rate,
                                uint32_t num_samples)
{
  const uint32_t bytes = num_samples * 2u * num_channels;
  if (d == NULL) return AAD_APIRESULT_INVALID_ARGUMENT;
  if (data_size < AAD_WAV_HEADER_SIZE) return AAD_APIRESULT_INSUFFICIENT_BUFFER;
  if (num_channels == 0 || sampling_rate == 0) return AAD_APIRESULT_INVALID_FORMAT;
  memcpy(d, "RIFF", 4);
  put32(d + 4, 36u + bytes);
  memcpy(d + 8, "WAVEfmt ", 8);
  put32(d + 16, 16);
  put16(d + 20, 1);
  put16(d + 22, num_channels);
  put32(d + 24, sampling_rate);
  put32(d + 28, sampling_rate * 2u * num_channels);
  put16(d + 32, (uint16_t)(2u * num_channels));
  put16(d + 34, 16);
  memcpy(d + 36, "data", 4);
  put32(d + 40, bytes);
  return AAD_APIRESULT_OK;
}

AADApiResult AADWav_ConvertToPcm16(const uint8_t *p, uint16_t bits, uint64_t count, int16_t *pcm)
{
  uint64_t i;
  if (p == NULL || pcm == NULL) return AAD_APIRESULT_INVALID_ARGUMENT;
  switch (bits) {
    case 8:
      for (i = 0; i < count; i++) pcm[i] = (int16_t)(((int32_t)p[i] - 128) * 256);
      return AAD_APIRESULT_OK;
    case 16:
    case 24:
    case 32: {
      const uint32_t step = bits / 8u;
      p += step - 2; /* the two most significant bytes of every little-endian sample */
      for (i = 0; i < count; i++, p += step) pcm[i] = (int16_t)le16(p);
      return AAD_APIRESULT_OK;
    }
    default:
      return AAD_APIRESULT_INVALID_FORMAT;
  }
}
