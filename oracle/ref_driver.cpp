// TEST INFRASTRUCTURE ONLY -- never linked into, loaded by, or called from the
// product path. Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load the library built from this file.
//
// Thin extern "C" driver around the REFERENCE's own classes, compiled by
// oracle/Makefile directly from the sources where they lie under
// /root/reference (nothing is copied into this repo):
//   src/vulkan/vulkan_streaming_upsampler.cpp   (CPU fallback: ENABLE_VULKAN off)
//   src/vulkan/fft_utils.h
//   src/audio/eq_parser.cpp, src/audio/eq_to_fir.cpp
// The output (oracle/_ref/libref_oracle.so) is the oracle used to (a) pin the
// CPU restatement in oracle/oracle_upsampler.c, (b) generate tests/golden/*,
// (c) serve as bench.py's cpu_baseline with kind="reference".
#include <complex>
#include <cstddef>
#include <cstring>
#include <string>
#include <vector>

#include "audio/eq_parser.h"
#include "audio/eq_to_fir.h"
#include "fft_utils.h"
#include "vulkan/vulkan_streaming_upsampler.h"

using totton::vulkan::VulkanStreamingUpsampler;

extern "C" {

void *ref_ups_create() { return new VulkanStreamingUpsampler(); }

void ref_ups_destroy(void *h) {
  delete static_cast<VulkanStreamingUpsampler *>(h);
}

void *ref_ups_clone(void *h) {
  return new VulkanStreamingUpsampler(
      *static_cast<VulkanStreamingUpsampler *>(h));
}

// returns 1 on success; on failure copies the reference's message into err.
int ref_ups_load_filter(void *h, const char *json_path, char *err,
                        size_t errcap) {
  std::string message;
  const bool ok =
      static_cast<VulkanStreamingUpsampler *>(h)->LoadFilter(json_path,
                                                             &message);
  if (err && errcap) {
    std::strncpy(err, message.c_str(), errcap - 1);
    err[errcap - 1] = '\0';
  }
  return ok ? 1 : 0;
}

void ref_ups_get_config(void *h, size_t *taps, size_t *fft, size_t *block,
                        size_t *factor) {
  const auto &c = static_cast<VulkanStreamingUpsampler *>(h)->GetConfig();
  *taps = c.taps;
  *fft = c.fftSize;
  *block = c.blockSize;
  *factor = c.upsampleFactor;
}

// returns number of output samples written (0 == the reference's empty vector).
long ref_ups_process_block(void *h, const float *in, size_t count, float *out,
                           size_t outcap) {
  std::vector<float> y =
      static_cast<VulkanStreamingUpsampler *>(h)->ProcessBlock(in, count);
  if (y.size() > outcap) {
    return -1;
  }
  if (!y.empty()) {
    std::memcpy(out, y.data(), y.size() * sizeof(float));
  }
  return static_cast<long>(y.size());
}

void ref_ups_reset(void *h) {
  static_cast<VulkanStreamingUpsampler *>(h)->Reset();
}

// The reference's fp32 recurrence-twiddle radix-2 FFT on interleaved re/im.
void ref_fft(float *reim, size_t n, int inverse) {
  std::vector<std::complex<float>> v(n);
  for (size_t i = 0; i < n; ++i) {
    v[i] = std::complex<float>(reim[2 * i], reim[2 * i + 1]);
  }
  totton::vulkan::fft::Fft(v, inverse != 0);
  for (size_t i = 0; i < n; ++i) {
    reim[2 * i] = v[i].real();
    reim[2 * i + 1] = v[i].imag();
  }
}

// ---- EQ (src/audio) -------------------------------------------------------
// Parses APO text with the reference parser. Returns number of bands, or -1
// when parseEqString returns false. Band fields are written as 8 doubles:
// enabled,type,frequency,gain,q,hasBwHz,bwHz,hasBwOct (bwOct in slot 8 -> 9).
long ref_eq_parse(const char *text, double *preamp_db, double *bands,
                  size_t max_bands) {
  EQ::EqProfile profile;
  if (!EQ::parseEqString(text, profile)) {
    return -1;
  }
  *preamp_db = profile.preampDb;
  const size_t n = profile.bands.size() < max_bands ? profile.bands.size()
                                                    : max_bands;
  for (size_t i = 0; i < n; ++i) {
    const auto &b = profile.bands[i];
    double *o = bands + 9 * i;
    o[0] = b.enabled ? 1.0 : 0.0;
    o[1] = static_cast<double>(static_cast<int>(b.type));
    o[2] = b.frequency;
    o[3] = b.gain;
    o[4] = b.q;
    o[5] = b.hasBandwidthHz ? 1.0 : 0.0;
    o[6] = b.bandwidthHz;
    o[7] = b.hasBandwidthOct ? 1.0 : 0.0;
    o[8] = b.bandwidthOct;
  }
  return static_cast<long>(profile.bands.size());
}

int ref_eq_parse_filter_type(const char *s) {
  return static_cast<int>(EQ::parseFilterType(s));
}

const char *ref_eq_filter_type_name(int t) {
  return EQ::filterTypeName(static_cast<EQ::FilterType>(t));
}

// biquad coefficients b0,b1,b2,a1,a2 for one band.
void ref_eq_biquad(int enabled, int type, double freq, double gain, double q,
                   double fs, double *out5) {
  EQ::EqBand band;
  band.enabled = enabled != 0;
  band.type = static_cast<EQ::FilterType>(type);
  band.frequency = freq;
  band.gain = gain;
  band.q = q;
  const EQ::BiquadCoeffs c = EQ::calculateBiquadCoeffs(band, fs);
  out5[0] = c.b0;
  out5[1] = c.b1;
  out5[2] = c.b2;
  out5[3] = c.a1;
  out5[4] = c.a2;
}

// computeEqResponseForFft on APO text; writes num_bins (re,im) pairs.
int ref_eq_response(const char *text, size_t num_bins, size_t full_fft,
                    double fs_out, double *reim) {
  EQ::EqProfile profile;
  EQ::parseEqString(text, profile);
  auto r = EQ::computeEqResponseForFft(num_bins, full_fft, fs_out, profile);
  for (size_t i = 0; i < r.size(); ++i) {
    reim[2 * i] = r[i].real();
    reim[2 * i + 1] = r[i].imag();
  }
  return static_cast<int>(r.size());
}

int ref_eq_magnitude(const char *text, size_t num_bins, size_t full_fft,
                     double fs_out, double *mag) {
  EQ::EqProfile profile;
  EQ::parseEqString(text, profile);
  auto r = EQ::computeEqMagnitudeForFft(num_bins, full_fft, fs_out, profile);
  for (size_t i = 0; i < r.size(); ++i) {
    mag[i] = r[i];
  }
  return static_cast<int>(r.size());
}

}  // extern "C"
