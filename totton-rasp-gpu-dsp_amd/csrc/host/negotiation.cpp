#include "negotiation.h"

#include <algorithm>

namespace miups {
namespace {
const int kRates44k[5] = {44100, 88200, 176400, 352800, 705600};
const int kRates48k[5] = {48000, 96000, 192000, 384000, 768000};
bool In(const int (&t)[5], int r) { return std::find(t, t + 5, r) != t + 5; }
}  // namespace

RateFamily GetRateFamily(int sampleRate) {
  if (In(kRates44k, sampleRate)) {
    return RateFamily::k44k;
  }
  if (In(kRates48k, sampleRate)) {
    return RateFamily::k48k;
  }
  // other rates: by divisibility, 48k family by default (auto_negotiation.cpp:24-31)
  if (sampleRate % 44100 == 0 || sampleRate % 11025 == 0) {
    return RateFamily::k44k;
  }
  return RateFamily::k48k;
}

bool IsSameFamily(int a, int b) { return GetRateFamily(a) == GetRateFamily(b); }

int TargetRateForFamily(RateFamily f) { return f == RateFamily::k44k ? 705600 : 768000; }

bool IsRateSupported(const DacRates &dac, int rate) {
  if (!dac.valid || rate < dac.minRate || rate > dac.maxRate) {
    return false;
  }
  return dac.rates.empty() || std::find(dac.rates.begin(), dac.rates.end(), rate) != dac.rates.end();
}

int BestRateForFamily(RateFamily f, const DacRates &dac) {
  if (!dac.valid) {
    return 0;
  }
  const int(&t)[5] = f == RateFamily::k44k ? kRates44k : kRates48k;
  for (int i = 4; i >= 0; --i) {  // highest supported multiple of the family's base rate
    if (IsRateSupported(dac, t[i])) {
      return t[i];
    }
  }
  return 0;
}

int CalculateUpsampleRatio(int inputRate, int outputRate) {
  if (inputRate <= 0 || outputRate <= 0 || outputRate % inputRate != 0) {
    return 0;
  }
  return outputRate / inputRate;
}

Negotiated Negotiate(int inputRate, const DacRates &dac, int currentOutputRate) {
  Negotiated n;
  n.inputRate = inputRate;
  if (inputRate <= 0) {
    n.errorMessage = "Invalid input rate: " + std::to_string(inputRate);
    return n;
  }
  if (!dac.valid) {
    n.errorMessage = "Invalid DAC capability: " + dac.errorMessage;
    return n;
  }
  n.family = GetRateFamily(inputRate);
  const int target = BestRateForFamily(n.family, dac);
  if (target == 0) {
    n.errorMessage = "No supported output rate for input family";
    return n;
  }
  if (target < inputRate) {
    n.errorMessage = "Target output rate (" + std::to_string(target) + ") is less than input rate (" +
                     std::to_string(inputRate) + ")";
    return n;
  }
  const int ratio = CalculateUpsampleRatio(inputRate, target);
  if (ratio == 0) {
    n.errorMessage = "Cannot calculate integer upsampling ratio";
    return n;
  }
  if (ratio != 1 && ratio != 2 && ratio != 4 && ratio != 8 && ratio != 16) {
    n.errorMessage = "Unsupported input rate: " + std::to_string(inputRate) + " Hz (ratio " + std::to_string(ratio) +
                     " not in {1, 2, 4, 8, 16})";
    return n;
  }
  n.outputRate = target;
  n.ratio = ratio;
  n.valid = true;
  // first configuration, or the output rate (= the family) changed: the device has to be reopened; an input-rate
  // change inside a family keeps the output rate and only swaps the filter
  n.requiresReconfiguration = currentOutputRate == 0 || currentOutputRate != target;
  return n;
}

}  // namespace miups
