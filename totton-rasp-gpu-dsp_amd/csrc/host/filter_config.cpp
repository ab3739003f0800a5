#include "filter_config.h"

#include <cctype>
#include <cstdio>
#include <filesystem>
#include <fstream>
#include <iterator>
#include <system_error>

namespace miups {
namespace {

// The reference does not parse JSON: it looks for the FIRST occurrence of the
// quoted key anywhere in the file, then the next ':' (vulkan_streaming_upsampler.cpp:24-73).
// Nested objects, duplicate keys and non-integer numbers therefore behave in
// specific ways (tests/golden/g6_load_errors.json pins them), so the same
// scan is used here rather than a real JSON parser.
std::size_t ValueStart(const std::string &text, const std::string &key) {
  const std::string quoted = '"' + key + '"';
  const std::size_t at = text.find(quoted);
  if (at == std::string::npos) {
    return std::string::npos;
  }
  return text.find(':', at + quoted.size());
}

bool ScanString(const std::string &text, const std::string &key, std::string *out) {
  std::size_t colon = ValueStart(text, key);
  if (colon == std::string::npos) {
    return false;
  }
  const std::size_t open = text.find('"', colon);
  if (open == std::string::npos) {
    return false;
  }
  const std::size_t close = text.find('"', open + 1);
  if (close == std::string::npos) {
    return false;
  }
  out->assign(text, open + 1, close - open - 1);
  return true;
}

bool ScanUnsigned(const std::string &text, const std::string &key, std::size_t *out) {
  std::size_t i = ValueStart(text, key);
  if (i == std::string::npos) {
    return false;
  }
  for (++i; i < text.size() && std::isspace(static_cast<unsigned char>(text[i])); ++i) {
  }
  std::size_t value = 0;
  std::size_t digits = 0;
  for (; i < text.size() && std::isdigit(static_cast<unsigned char>(text[i])); ++i, ++digits) {
    value = value * 10 + static_cast<std::size_t>(text[i] - '0');
  }
  if (digits == 0) {
    return false;
  }
  *out = value;
  return true;
}

bool Fail(std::string *errorMessage, const std::string &message) {
  if (errorMessage) {
    *errorMessage = message;
  }
  return false;
}

bool IsPow2(std::size_t v) { return v != 0 && (v & (v - 1)) == 0; }

}  // namespace

bool ReadFilterConfig(const std::string &jsonPath, FilterConfig *config, std::string *errorMessage) {
  std::string text;
  {
    std::ifstream file(jsonPath);
    if (file) {
      text.assign(std::istreambuf_iterator<char>(file), std::istreambuf_iterator<char>());
    }
  }
  if (text.empty()) {
    return Fail(errorMessage, "Failed to read filter config: " + jsonPath);
  }

  std::string bin;
  if (!ScanString(text, "coefficients_bin", &bin)) {
    return Fail(errorMessage, "Missing coefficients_bin in filter config");
  }
  std::size_t taps = 0, fft = 0, block = 0;
  ScanUnsigned(text, "taps", &taps);
  ScanUnsigned(text, "fft_size", &fft);
  ScanUnsigned(text, "block_size", &block);
  // the factor lands in *config even when a later check fails (as in the reference)
  ScanUnsigned(text, "upsample_factor", &config->upsampleFactor);

  if (taps == 0 || fft == 0 || block == 0) {
    return Fail(errorMessage, "taps/fft_size/block_size must be set and non-zero");
  }
  if (!IsPow2(fft)) {
    return Fail(errorMessage, "fft_size must be power of two");
  }
  if (block >= fft) {
    return Fail(errorMessage, "block_size must be smaller than fft_size");
  }
  if (fft - block != taps - 1) {
    return Fail(errorMessage, "block_size must satisfy fft_size - block_size == taps - 1");
  }

  std::filesystem::path binPath = bin;
  if (!binPath.is_absolute()) {
    binPath = std::filesystem::path(jsonPath).parent_path() / binPath;
  }
  config->coefficientsPath = binPath.string();
  config->taps = taps;
  config->fftSize = fft;
  config->blockSize = block;
  if (config->upsampleFactor == 0) {
    config->upsampleFactor = 1;
  }
  if (config->upsampleFactor > 1 && block % config->upsampleFactor != 0) {
    return Fail(errorMessage, "block_size must be divisible by upsample_factor");
  }
  return true;
}

bool ReadCoefficients(const FilterConfig &config, std::vector<float> *coefficients, std::string *errorMessage) {
  std::error_code ec;
  const auto size = std::filesystem::file_size(config.coefficientsPath, ec);
  if (ec) {
    return Fail(errorMessage, "Failed to stat coefficients: " + config.coefficientsPath);
  }
  std::ifstream file(config.coefficientsPath, std::ios::binary);
  if (!file) {
    return Fail(errorMessage, "Failed to open coefficients: " + config.coefficientsPath);
  }
  const std::size_t expected = config.taps * sizeof(float);
  if (size != expected) {
    return Fail(errorMessage, "Coefficient file size does not match taps");
  }
  std::vector<float> taps(config.taps, 0.0f);
  file.read(reinterpret_cast<char *>(taps.data()), static_cast<std::streamsize>(expected));
  if (static_cast<std::size_t>(file.gcount()) != expected) {
    return Fail(errorMessage, "Coefficient file size does not match taps");
  }
  coefficients->swap(taps);
  return true;
}

bool ReadFilter(const std::string &jsonPath, FilterConfig *config, std::vector<float> *coefficients,
                std::string *errorMessage) {
  FilterConfig parsed;
  if (!ReadFilterConfig(jsonPath, &parsed, errorMessage)) {
    return false;
  }
  if (!ReadCoefficients(parsed, coefficients, errorMessage)) {
    return false;
  }
  if (parsed.taps > parsed.fftSize) {
    return Fail(errorMessage, "taps must be <= fft_size for minimal overlap-save");
  }
  *config = parsed;
  return true;
}

}  // namespace miups
