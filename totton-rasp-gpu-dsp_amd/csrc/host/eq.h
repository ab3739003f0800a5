// Equalizer-APO text -> band list -> biquad cascade, folded into the filter at load time as a TRUE FIR: the cascade's
// recursion runs over the filter taps (fp64) and the result is cut back to `taps` samples (FoldCascadeIntoTaps), so the
// overlap-save geometry fft - block == taps - 1 still holds and no block-periodic time aliasing exists. The per-bin
// response on the N-point grid (computeEqResponseForFft) is kept as the reference's interface and as the yardstick the
// folded FIR's response is measured against (ResponseDeviation).
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

// ---- EQ folded into the FIR ("h_total truncated/windowed back to taps, state the residual") ----------
// h_ideal = taps (*) h_cascade is infinitely long; the upsampler convolves with
//   fir[n] = w[n] * h_ideal[n],  0 <= n < taps,
//   w = 1 except a closing half-Hann over the last W = (taps - 1) / 64 samples (a hard cut of a low-frequency tail is a
//   step that lifts the stop band from -140 to -110 dB; the short roll-off keeps it, profiles/r04_a_eq_fold.txt).
// Residual = what the cut drops, relative to the ideal response: tailL1 = ||h_ideal - fir||_1 / ||h_ideal||_1 bounds the
// output error for any input by tailL1 * ||h_ideal||_1 * max|x|; tailL2 the same in the 2-norm. The free decay after the
// last tap is followed until a 4096-sample chunk adds less than 1e-17 of the total, for at most kFoldTailWork
// section-steps; when it has not died by then (tailComplete = false) the rest is ESTIMATED from the largest pole radius.
struct EqFold {
  std::vector<double> fir;  // `taps` samples
  double tailL1 = 0.0, tailL2 = 0.0;
  bool tailComplete = true;
  std::size_t taper = 0;       // W
  double responseDev = 0.0;    // max_k |FFT_N(fir)[k] - H_fir[k] EQ[k]| / max_k |H_fir[k] EQ[k]|  (ResponseDeviation)
};
constexpr double kFoldTailWork = 2.0e8;
// default threshold on tailL1 above which an EQ change is reported (or, in strict mode, refused): -60 dB
constexpr double kFoldDefaultLimit = 1.0e-3;
EqFold FoldCascadeIntoTaps(const std::vector<float> &taps, const Cascade &cascade);
// how far the folded FIR's response is from H_fir * EQ on the N-point grid (eqHalf: bins 0..N/2, from the device kernel
// or ComputeEqResponseHost); both spectra from ONE fp64 transform of taps + j fir
double ResponseDeviation(const std::vector<float> &taps, const std::vector<double> &fir,
                         const std::vector<std::complex<double>> &eqHalf, std::size_t fftSize);
// "EQ cut to 80001 taps drops -22.3 dB of the ideal response (limit -60.0 dB): ..." ("" when within the limit)
std::string FoldWarning(std::size_t firTaps, double tailL1, double tailL2, bool tailComplete, double limit);

}  // namespace miups::eq
