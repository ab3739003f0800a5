// C++ face of the drop-in boundary: the same public surface as the reference's
// totton::vulkan::VulkanStreamingUpsampler
// (include/vulkan/vulkan_streaming_upsampler.h:12-49), implemented on top of the
// C ABI in mi_upsampler.h. A caller written against the reference class
// (src/alsa/alsa_streamer_main.cpp:198-252,321,543) compiles against this one by
// changing the include and the namespace alias:
//
//   #include "hip_streaming_upsampler.h"
//   namespace totton::vulkan { using VulkanStreamingUpsampler = totton::hip::HipStreamingUpsampler;
//                              using FilterConfig = totton::hip::FilterConfig; }
//
// Semantics kept: LoadFilter -> bool + message; ProcessBlock -> vector of
// exactly blockSize samples, or an EMPTY vector on every guard/backend failure;
// Reset zeroes the carried history; copies are deep (own history) and movable.
// Not kept: the silent CPU fallback -- without a usable HIP device LoadFilter
// fails with a message.
#pragma once

#include <cstddef>
#include <string>
#include <utility>
#include <vector>

#include "mi_upsampler.h"

namespace totton::hip {

struct FilterConfig {
  std::string coefficientsPath;
  std::size_t taps = 0;
  std::size_t fftSize = 0;
  std::size_t blockSize = 0;
  std::size_t upsampleFactor = 1;
};

class HipStreamingUpsampler {
 public:
  explicit HipStreamingUpsampler(int device = 0) : device_(device), handle_(mi_ups_create(device)) {}
  HipStreamingUpsampler(const HipStreamingUpsampler &other)
      : device_(other.device_), handle_(other.handle_ ? mi_ups_clone(other.handle_) : nullptr), config_(other.config_) {}
  HipStreamingUpsampler &operator=(const HipStreamingUpsampler &other) {
    if (this != &other) {
      HipStreamingUpsampler copy(other);
      swap(copy);
    }
    return *this;
  }
  HipStreamingUpsampler(HipStreamingUpsampler &&other) noexcept { swap(other); }
  HipStreamingUpsampler &operator=(HipStreamingUpsampler &&other) noexcept {
    if (this != &other) {
      HipStreamingUpsampler moved(std::move(other));
      swap(moved);
    }
    return *this;
  }
  ~HipStreamingUpsampler() {
    if (handle_) {
      mi_ups_destroy(handle_);
    }
  }

  bool LoadFilter(const std::string &jsonPath, std::string *errorMessage, int flags = MI_LOAD_DEFAULT) {
    char message[1280] = {0};
    if (!handle_) {
      handle_ = mi_ups_create(device_);
    }
    const int rc = handle_ ? mi_ups_load_filter(handle_, jsonPath.c_str(), flags, message, sizeof(message)) : MI_ERR_DEVICE;
    if (rc != MI_OK) {
      if (errorMessage) {
        *errorMessage = message[0] ? message : "HIP upsampler unavailable";
      }
      return false;
    }
    mi_ups_config c;
    mi_ups_get_config(handle_, &c);
    config_.coefficientsPath = c.coefficients_path;
    config_.taps = c.taps;
    config_.fftSize = c.fft_size;
    config_.blockSize = c.block_size;
    config_.upsampleFactor = c.upsample_factor;
    return true;
  }

  std::vector<float> ProcessBlock(const float *input, std::size_t count) {
    if (!handle_ || config_.blockSize == 0) {
      return {};
    }
    std::vector<float> out(config_.blockSize, 0.0f);
    const long n = mi_ups_process_block(handle_, input, count, out.data(), out.size());
    if (n <= 0) {
      return {};
    }
    out.resize(static_cast<std::size_t>(n));
    return out;
  }

  void Reset() {
    if (handle_) {
      mi_ups_reset(handle_);
    }
  }

  const FilterConfig &GetConfig() const { return config_; }

  // extension: fold an Equalizer-APO profile into the filter spectrum
  bool SetEq(const std::string &apoText, double outputSampleRate) {
    return handle_ && mi_ups_set_eq(handle_, apoText.c_str(), outputSampleRate) == MI_OK;
  }

 private:
  void swap(HipStreamingUpsampler &o) noexcept {
    std::swap(device_, o.device_);
    std::swap(handle_, o.handle_);
    std::swap(config_, o.config_);
  }

  int device_ = 0;
  mi_ups *handle_ = nullptr;
  FilterConfig config_{};
};

}  // namespace totton::hip
