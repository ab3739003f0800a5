// Compile-and-run check of the C++ face (include/hip_streaming_upsampler.h) used the way a
// caller of the reference class uses it: through the totton::vulkan names.
//
// The three GPU cases restate the known-answer checks of the reference's
// tests/cpp/test_vulkan_upsampler.cpp:124-195 (5-tap filter {1,2,3,2,1}, fft 16, block 12:
// impulse response, two streamed blocks = the overlap carry, 2x zero-stuffed impulse; abs 1e-3,
// its own eps :68-70), then add what the drop-in promises on top: deep copies, moves,
// LoadFilter failure strings.
//
//   test_hip_upsampler <tmpdir>            GPU run (pytest -m gpu)
//   test_hip_upsampler <tmpdir> --no-gpu   host-only part (error strings, copy/move of an unloaded object)
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "hip_streaming_upsampler.h"

namespace totton::vulkan {
using VulkanStreamingUpsampler = totton::hip::HipStreamingUpsampler;
using FilterConfig = totton::hip::FilterConfig;
}  // namespace totton::vulkan

namespace {

int g_failures = 0;

#define EXPECT(cond)                                                        \
  do {                                                                      \
    if (!(cond)) {                                                          \
      std::fprintf(stderr, "%s:%d: EXPECT(%s) failed\n", __FILE__, __LINE__, #cond); \
      ++g_failures;                                                         \
    }                                                                       \
  } while (0)

const float kTaps[5] = {1.0f, 2.0f, 3.0f, 2.0f, 1.0f};

std::string WriteFilter(const std::string &dir, const std::string &name, unsigned factor) {
  {
    std::ofstream bin(dir + "/" + name + ".bin", std::ios::binary);
    bin.write(reinterpret_cast<const char *>(kTaps), sizeof(kTaps));
  }
  std::ofstream js(dir + "/" + name + ".json");
  js << "{\"coefficients_bin\": \"" << name << ".bin\", \"taps\": 5, \"fft_size\": 16, \"block_size\": 12";
  if (factor > 0) {
    js << ", \"upsample_factor\": " << factor;
  }
  js << "}\n";
  return dir + "/" + name + ".json";
}

// first `count` samples of the direct convolution of zero-stuffed x with the taps
std::vector<float> DirectFir(const std::vector<float> &x, unsigned factor, std::size_t count) {
  std::vector<double> up(x.size() * factor, 0.0);
  for (std::size_t i = 0; i < x.size(); ++i) {
    up[i * factor] = x[i];
  }
  std::vector<float> y(count, 0.0f);
  for (std::size_t n = 0; n < count; ++n) {
    double acc = 0.0;
    for (std::size_t k = 0; k < 5 && k <= n; ++k) {
      if (n - k < up.size()) {
        acc += kTaps[k] * up[n - k];
      }
    }
    y[n] = static_cast<float>(acc);
  }
  return y;
}

bool Near(const std::vector<float> &a, const std::vector<float> &b, float eps = 1e-3f) {
  if (a.size() != b.size()) {
    return false;
  }
  for (std::size_t i = 0; i < a.size(); ++i) {
    if (!(std::fabs(a[i] - b[i]) <= eps)) {
      return false;
    }
  }
  return true;
}

void HostOnly(const std::string &dir) {
  totton::vulkan::VulkanStreamingUpsampler u;
  std::string err;
  // failure strings come from the sidecar validation, before any device work
  EXPECT(!u.LoadFilter(dir + "/does_not_exist.json", &err));
  EXPECT(err == "Failed to read filter config: " + dir + "/does_not_exist.json");
  {
    std::ofstream js(dir + "/bad.json");
    js << "{\"coefficients_bin\": \"k1.bin\", \"taps\": 5, \"fft_size\": 18, \"block_size\": 14}";
  }
  EXPECT(!u.LoadFilter(dir + "/bad.json", &err));
  EXPECT(err == "fft_size must be power of two");
  EXPECT(!u.LoadFilter(dir + "/bad.json", nullptr));  // a null message pointer is allowed
  // an object without a filter: ProcessBlock -> empty vector, config all zero / factor 1
  const float x[12] = {0};
  EXPECT(u.ProcessBlock(x, 12).empty());
  EXPECT(u.ProcessBlock(nullptr, 12).empty());
  EXPECT(u.GetConfig().blockSize == 0 && u.GetConfig().upsampleFactor == 1);
  // copies and moves of it are well-formed objects
  totton::vulkan::VulkanStreamingUpsampler c(u);
  totton::vulkan::VulkanStreamingUpsampler m(std::move(c));
  c = m;
  m = std::move(c);
  std::vector<totton::vulkan::VulkanStreamingUpsampler> perChannel;
  perChannel.assign(3, u);  // alsa_streamer_main.cpp:248-250
  EXPECT(perChannel.size() == 3 && perChannel[2].ProcessBlock(x, 12).empty());
  u.Reset();
}

void OnGpu(const std::string &dir) {
  std::string err;
  totton::vulkan::VulkanStreamingUpsampler up1;
  if (!up1.LoadFilter(WriteFilter(dir, "k1", 1), &err)) {
    std::fprintf(stderr, "LoadFilter failed: %s\n", err.c_str());
    ++g_failures;
    return;
  }
  const totton::vulkan::FilterConfig &cfg = up1.GetConfig();
  EXPECT(cfg.taps == 5 && cfg.fftSize == 16 && cfg.blockSize == 12 && cfg.upsampleFactor == 1);
  EXPECT(cfg.coefficientsPath == dir + "/k1.bin");

  // (a) impulse response == direct convolution
  std::vector<float> imp(12, 0.0f);
  imp[4] = 1.0f;
  const std::vector<float> impOut = up1.ProcessBlock(imp.data(), imp.size());
  EXPECT(impOut.size() == 12);
  EXPECT(Near(impOut, DirectFir(imp, 1, 12)));

  // (b) two consecutive blocks == first 24 samples of the streamed convolution (overlap carry)
  std::vector<float> a(12), b(12);
  for (int i = 0; i < 12; ++i) {
    a[i] = 1.0f + i;
    b[i] = 101.0f + i;
  }
  up1.Reset();
  std::vector<float> got = up1.ProcessBlock(a.data(), a.size());
  // copy made mid-stream: carries the history of block a (deep copy, reference :455-479)
  totton::vulkan::VulkanStreamingUpsampler fork(up1);
  const std::vector<float> outB = up1.ProcessBlock(b.data(), b.size());
  got.insert(got.end(), outB.begin(), outB.end());
  std::vector<float> ab = a;
  ab.insert(ab.end(), b.begin(), b.end());
  EXPECT(Near(got, DirectFir(ab, 1, 24)));
  const std::vector<float> forkB = fork.ProcessBlock(b.data(), b.size());
  EXPECT(forkB == outB);  // same kernels, same history: bit-identical
  fork.Reset();           // and independent: resetting the copy leaves the original's history alone
  EXPECT(Near(fork.ProcessBlock(a.data(), a.size()), DirectFir(a, 1, 12)));
  std::vector<float> c(12, 0.0f);
  std::vector<float> abc = ab;
  abc.insert(abc.end(), c.begin(), c.end());
  const std::vector<float> tail = DirectFir(abc, 1, 36);
  EXPECT(Near(up1.ProcessBlock(c.data(), c.size()), std::vector<float>(tail.begin() + 24, tail.end())));

  // guards: wrong counts and a null input return an empty vector (reference :502-519)
  EXPECT(up1.ProcessBlock(a.data(), 0).empty());
  EXPECT(up1.ProcessBlock(a.data(), 11).empty());
  EXPECT(up1.ProcessBlock(nullptr, 12).empty());

  // (c) upsample_factor 2: six input samples -> twelve outputs == zero-stuff + convolve
  totton::vulkan::VulkanStreamingUpsampler up2;
  if (!up2.LoadFilter(WriteFilter(dir, "k2", 2), &err)) {
    std::fprintf(stderr, "LoadFilter failed (2x): %s\n", err.c_str());
    ++g_failures;
    return;
  }
  std::vector<float> imp2(6, 0.0f);
  imp2[2] = 1.0f;
  const std::vector<float> out2 = up2.ProcessBlock(imp2.data(), imp2.size());
  EXPECT(out2.size() == 12);
  EXPECT(Near(out2, DirectFir(imp2, 2, 12)));
  EXPECT(up2.ProcessBlock(imp2.data(), 12).empty());  // block, not block / factor

  // move keeps the loaded state; the moved-from object is empty but usable
  totton::vulkan::VulkanStreamingUpsampler moved(std::move(up2));
  EXPECT(moved.GetConfig().upsampleFactor == 2);
  moved.Reset();
  EXPECT(Near(moved.ProcessBlock(imp2.data(), imp2.size()), DirectFir(imp2, 2, 12)));
  // vector::assign of a loaded object = one clone per channel (alsa_streamer_main.cpp:248-250)
  std::vector<totton::vulkan::VulkanStreamingUpsampler> perChannel;
  perChannel.assign(4, moved);
  for (auto &u : perChannel) {
    u.Reset();
    EXPECT(Near(u.ProcessBlock(imp2.data(), imp2.size()), DirectFir(imp2, 2, 12)));
  }
}

}  // namespace

int main(int argc, char **argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: %s <tmpdir> [--no-gpu]\n", argv[0]);
    return 2;
  }
  const std::string dir = argv[1];
  WriteFilter(dir, "k1", 1);
  HostOnly(dir);
  if (argc < 3 || std::strcmp(argv[2], "--no-gpu") != 0) {
    OnGpu(dir);
  }
  if (g_failures) {
    std::fprintf(stderr, "%d check(s) failed\n", g_failures);
    return 1;
  }
  std::puts("OK");
  return 0;
}
