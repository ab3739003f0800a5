// Equalizer-APO text -> band list -> per-bin response on the upsampler's
// N-point frequency grid, folded into the filter spectrum at load time.
//
// Mirrors the reference's EQ interface (include/audio/eq_parser.h,
// include/audio/eq_to_fir.h): same type list, same grammar, same RBJ biquads
// (PK / LS / HS implemented, every other type parses and evaluates as bypass),
// same fp64 arithmetic. The per-bin cascade itself is evaluated by a HIP kernel
// (eq_response.hip); ComputeEqResponseHost is the host statement of the same
// maths used by the CPU-only tests.
#pragma once

#include <complex>
#include <cstddef>
#include <string>
#include <vector>

namespace miups::eq {

// include/audio/eq_parser.h:13-39 (order matters: tests pin the integer ids)
enum class FilterType {
  PK, MODAL, PEQ, LP, LPQ, HP, HPQ, BP, NO, AP, LS, HS, LSC, HSC, LSQ, HSQ, LS_6DB, LS_12DB, HS_6DB, HS_12DB,
};

struct EqBand {
  bool enabled = true;
  FilterType type = FilterType::PK;
  double frequency = 1000.0;
  double gain = 0.0;
  double q = 1.0;
  bool hasBandwidthHz = false;
  double bandwidthHz = 0.0;
  bool hasBandwidthOct = false;
  double bandwidthOct = 0.0;
};

struct EqProfile {
  std::string name;
  double preampDb = 0.0;
  std::vector<EqBand> bands;
  bool isEmpty() const { return bands.empty() && preampDb == 0.0; }
  std::size_t activeBandCount() const;
};

struct BiquadCoeffs {
  double b0, b1, b2, a1, a2;
};

const char *filterTypeName(FilterType type);
FilterType parseFilterType(const std::string &typeStr);
bool parseEqString(const std::string &content, EqProfile &profile);
bool parseEqFile(const std::string &filePath, EqProfile &profile);

BiquadCoeffs calculateBiquadCoeffs(const EqBand &band, double sampleRate);

// Cascade description handed to the device kernel: linear preamp and the
// normalised coefficients of every ENABLED band (bypass bands included as
// 1,0,0,0,0 so host and device multiply the same factors in the same order).
struct Cascade {
  double preampLinear = 1.0;
  std::vector<BiquadCoeffs> sections;
};
Cascade buildCascade(const EqProfile &profile, double sampleRate);

// computeEqResponseForFft (eq_to_fir.cpp:145-151): bins i = 0..numBins-1 at
// f_i = i * outputSampleRate / fullFftSize.
std::vector<std::complex<double>> ComputeEqResponseHost(std::size_t numBins, std::size_t fullFftSize,
                                                        double outputSampleRate, const EqProfile &profile);
// computeEqMagnitudeForFft (eq_to_fir.cpp:153-177): |H|, divided by max if max > 1.
std::vector<double> ComputeEqMagnitudeHost(std::size_t numBins, std::size_t fullFftSize, double outputSampleRate,
                                           const EqProfile &profile);

}  // namespace miups::eq
