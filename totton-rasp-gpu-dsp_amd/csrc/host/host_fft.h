// Host-side, load-time-only transforms used to build the device tables.
//  * Fft64: fp64 radix-2 with directly evaluated twiddles (accurate to ~1e-16);
//    runs a handful of times per LoadFilter / EQ reload, never per block.
//  * FftRefCompat32: fp32 radix-2 whose twiddles come from the recurrence
//    w *= wlen -- the arithmetic the reference uses for its filter spectrum on
//    BOTH of its paths (src/vulkan/fft_utils.h:30-61, called from
//    vulkan_streaming_upsampler.cpp:734-739). Only the optional
//    "reference-compatible spectrum" load flag uses it.
#pragma once

#include <cmath>
#include <complex>
#include <cstddef>
#include <utility>
#include <vector>

namespace miups {

template <typename C>
inline void BitReversePermute(std::vector<C> &a) {
  const std::size_t n = a.size();
  for (std::size_t i = 1, j = 0; i < n; ++i) {
    std::size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) {
      j ^= bit;
    }
    j ^= bit;
    if (i < j) {
      std::swap(a[i], a[j]);
    }
  }
}

inline void Fft64(std::vector<std::complex<double>> &a, bool inverse) {
  const std::size_t n = a.size();
  if (n <= 1) {
    return;
  }
  BitReversePermute(a);
  const double pi = 3.14159265358979323846264338327950288;
  std::vector<std::complex<double>> w(n / 2);
  for (std::size_t k = 0; k < n / 2; ++k) {
    const double ang = (inverse ? 2.0 : -2.0) * pi * static_cast<double>(k) / static_cast<double>(n);
    w[k] = std::complex<double>(std::cos(ang), std::sin(ang));
  }
  for (std::size_t len = 2; len <= n; len <<= 1) {
    const std::size_t step = n / len;
    for (std::size_t i = 0; i < n; i += len) {
      for (std::size_t j = 0; j < len / 2; ++j) {
        const std::complex<double> u = a[i + j];
        const std::complex<double> v = a[i + j + len / 2] * w[j * step];
        a[i + j] = u + v;
        a[i + j + len / 2] = u - v;
      }
    }
  }
  if (inverse) {
    const double s = 1.0 / static_cast<double>(n);
    for (auto &x : a) {
      x *= s;
    }
  }
}

inline void FftRefCompat32(std::vector<std::complex<float>> &a, bool inverse) {
  const std::size_t n = a.size();
  if (n <= 1) {
    return;
  }
  BitReversePermute(a);
  const float pi = 3.14159265358979323846f;
  for (std::size_t len = 2; len <= n; len <<= 1) {
    const float ang = (inverse ? 2.0f : -2.0f) * pi / static_cast<float>(len);
    const float wlr = std::cos(ang), wli = std::sin(ang);
    for (std::size_t i = 0; i < n; i += len) {
      float wr = 1.0f, wi = 0.0f;
      for (std::size_t j = 0; j < len / 2; ++j) {
        const std::complex<float> u = a[i + j];
        const std::complex<float> x = a[i + j + len / 2];
        const std::complex<float> v(x.real() * wr - x.imag() * wi, x.real() * wi + x.imag() * wr);
        a[i + j] = u + v;
        a[i + j + len / 2] = u - v;
        const float nr = wr * wlr - wi * wli;
        wi = wr * wli + wi * wlr;
        wr = nr;
      }
    }
  }
  if (inverse) {
    const float s = 1.0f / static_cast<float>(n);
    for (auto &x : a) {
      x *= s;
    }
  }
}

}  // namespace miups
