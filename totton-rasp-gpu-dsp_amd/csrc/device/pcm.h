// PCM <-> float conversion, usable on device and host (the host uses the same
// inline functions for the single-block drop-in API's error checks and the
// streamer's pass-through mode).
//
// Arithmetic follows the reference exactly (src/alsa/alsa_common.cpp:42-127):
//   to float : (float)int * 2^-15 | 2^-23 | 2^-31      (s16 | s24_3le | s32)
//   to PCM   : clamp to [-1, 0.9999695] (s16) or [-1, 0.9999999] (s24, s32),
//              multiply by 2^15 | 2^23 | 2^31 in fp32, truncate toward zero.
#pragma once

#include "common.h"

namespace miups {

MI_HD float pcm_load(const void *base, int fmt, long long idx) {
  const unsigned char *p = static_cast<const unsigned char *>(base);
  if (fmt == kF32) {
    return reinterpret_cast<const float *>(p)[idx];
  }
  if (fmt == kS32) {
    return static_cast<float>(reinterpret_cast<const int32_t *>(p)[idx]) * (1.0f / 2147483648.0f);
  }
  if (fmt == kS16) {
    return static_cast<float>(reinterpret_cast<const int16_t *>(p)[idx]) * (1.0f / 32768.0f);
  }
  const unsigned char *q = p + 3 * idx;
  int32_t v = static_cast<int32_t>(q[0]) | (static_cast<int32_t>(q[1]) << 8) | (static_cast<int32_t>(q[2]) << 16);
  if (v & 0x00800000) {
    v |= static_cast<int32_t>(0xFF000000u);
  }
  return static_cast<float>(v) * (1.0f / 8388608.0f);
}

MI_HD float pcm_clamp(float x, float hi) {
  // std::max(-1.0f, std::min(hi, x)) with the reference's NaN behaviour
  const float m = (x < hi) ? x : hi;
  return (m < -1.0f) ? -1.0f : m;
}

MI_HD void pcm_store(void *base, int fmt, long long idx, float x) {
  unsigned char *p = static_cast<unsigned char *>(base);
  if (fmt == kF32) {
    reinterpret_cast<float *>(p)[idx] = x;
    return;
  }
  if (fmt == kS32) {
    reinterpret_cast<int32_t *>(p)[idx] = static_cast<int32_t>(pcm_clamp(x, 0.9999999f) * 2147483648.0f);
    return;
  }
  if (fmt == kS16) {
    reinterpret_cast<int16_t *>(p)[idx] = static_cast<int16_t>(pcm_clamp(x, 0.9999695f) * 32768.0f);
    return;
  }
  const int32_t v = static_cast<int32_t>(pcm_clamp(x, 0.9999999f) * 8388608.0f);
  unsigned char *q = p + 3 * idx;
  q[0] = static_cast<unsigned char>(v & 0xFF);
  q[1] = static_cast<unsigned char>((v >> 8) & 0xFF);
  q[2] = static_cast<unsigned char>((v >> 16) & 0xFF);
}

}  // namespace miups
