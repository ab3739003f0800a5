// Filter sidecar (.json) + coefficient (.bin, float32 LE) loading.
// Same accepted files, same validation order and the same error strings as the
// reference's LoadFilterConfig / LoadCoefficients / PrepareSpectrum guards
// (src/vulkan/vulkan_streaming_upsampler.cpp:606-732; format: docs/filter_format.md).
#pragma once

#include <cstddef>
#include <string>
#include <vector>

namespace miups {

// include/vulkan/vulkan_streaming_upsampler.h:12-18
struct FilterConfig {
  std::string coefficientsPath;
  std::size_t taps = 0;
  std::size_t fftSize = 0;
  std::size_t blockSize = 0;
  std::size_t upsampleFactor = 1;
};

bool ReadFilterConfig(const std::string &jsonPath, FilterConfig *config, std::string *errorMessage);
bool ReadCoefficients(const FilterConfig &config, std::vector<float> *coefficients, std::string *errorMessage);
// config + coefficients + the taps <= fft_size guard, in the reference's order.
bool ReadFilter(const std::string &jsonPath, FilterConfig *config, std::vector<float> *coefficients,
                std::string *errorMessage);

}  // namespace miups
