// Rate-family detection, output-rate negotiation and ratio validation: the step between "an input rate appeared" and
// "this filter runs" (reference: src/audio/auto_negotiation.cpp:72-155, include/audio/pcm_format_set.h:48-52,
// src/io/dac_capability.cpp:133-145). The DAC is described by what the reference's ALSA probe would report
// (min / max rate and, when the device lists them, the discrete rates); probing itself stays out of scope.
#pragma once

#include <string>
#include <vector>

namespace miups {

enum class RateFamily { kUnknown = 0, k44k = 1, k48k = 2 };

struct DacRates {
  bool valid = false;
  int minRate = 0, maxRate = 0;
  std::vector<int> rates;  // empty: every rate in [minRate, maxRate]
  std::string errorMessage;
};

struct Negotiated {
  int inputRate = 0;
  RateFamily family = RateFamily::kUnknown;
  int outputRate = 0;
  int ratio = 0;
  bool valid = false;
  bool requiresReconfiguration = false;  // the output device has to be reopened (family / output rate changed)
  std::string errorMessage;
};

RateFamily GetRateFamily(int sampleRate);
bool IsSameFamily(int a, int b);
int TargetRateForFamily(RateFamily f);  // 705600 / 768000
bool IsRateSupported(const DacRates &dac, int rate);
int BestRateForFamily(RateFamily f, const DacRates &dac);
int CalculateUpsampleRatio(int inputRate, int outputRate);
Negotiated Negotiate(int inputRate, const DacRates &dac, int currentOutputRate);

}  // namespace miups
