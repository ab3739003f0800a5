#include "eq.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <fstream>
#include <iostream>
#include <iterator>
#include <regex>
#include <sstream>
#include <utility>

namespace miups::eq {
namespace {

constexpr double kPi = 3.14159265358979323846;

struct TypeEntry {
  FilterType type;
  const char *label;                 // filterTypeName()
  std::initializer_list<const char *> spellings;  // upper-case aliases
};

// type ids, display names and accepted spellings: eq_parser.cpp:23-141
const TypeEntry kTypes[] = {
    {FilterType::PK, "PK", {"PK", "PEAK", "PEAKING"}},
    {FilterType::MODAL, "MODAL", {"MODAL"}},
    {FilterType::PEQ, "PEQ", {"PEQ"}},
    {FilterType::LP, "LP", {"LP", "LOWPASS"}},
    {FilterType::LPQ, "LPQ", {"LPQ"}},
    {FilterType::HP, "HP", {"HP", "HIGHPASS"}},
    {FilterType::HPQ, "HPQ", {"HPQ"}},
    {FilterType::BP, "BP", {"BP", "BANDPASS"}},
    {FilterType::NO, "NO", {"NO", "NOTCH"}},
    {FilterType::AP, "AP", {"AP", "ALLPASS"}},
    {FilterType::LS, "LS", {"LS", "LOWSHELF"}},
    {FilterType::HS, "HS", {"HS", "HIGHSHELF"}},
    {FilterType::LSC, "LSC", {"LSC"}},
    {FilterType::HSC, "HSC", {"HSC"}},
    {FilterType::LSQ, "LSQ", {"LSQ"}},
    {FilterType::HSQ, "HSQ", {"HSQ"}},
    {FilterType::LS_6DB, "LS 6dB", {"LS 6DB", "LS6DB"}},
    {FilterType::LS_12DB, "LS 12dB", {"LS 12DB", "LS12DB"}},
    {FilterType::HS_6DB, "HS 6dB", {"HS 6DB", "HS6DB"}},
    {FilterType::HS_12DB, "HS 12dB", {"HS 12DB", "HS12DB"}},
};

std::string Upper(std::string s) {
  std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return static_cast<char>(std::toupper(c)); });
  return s;
}

std::string Strip(const std::string &s) {
  const char *ws = " \t\r\n";
  const auto b = s.find_first_not_of(ws);
  if (b == std::string::npos) {
    return std::string();
  }
  return s.substr(b, s.find_last_not_of(ws) - b + 1);
}

// Q from "BW Oct x" (eq_parser.cpp:151-166) and from "BW x Hz" (:168-175)
double QFromOctaves(double oct) {
  if (oct <= 0.0) {
    return 1.0;
  }
  const double den = 2.0 * std::sinh(0.34657359037935203 * oct);
  return den > 0.0 ? 1.0 / den : 1.0;
}
double QFromHz(double fc, double bw) { return (fc > 0.0 && bw > 0.0) ? fc / bw : 1.0; }

// The APO line grammar, as the reference's regular expressions define it
// (eq_parser.cpp:184-194, 232-246). Each token is searched anywhere in the line.
struct Grammar {
  std::regex preamp{R"(Preamp:\s*([-+]?\d+\.?\d*)\s*[dD][bB]?)", std::regex::icase};
  std::regex filter{R"(Filter\s*(\d+)?\s*:\s*(ON|OFF)\s+(.+?)\s+Fc\s+([\d.]+)\s*(?:Hz)?)", std::regex::icase};
  std::regex gain{R"(Gain\s+([-+]?\d+\.?\d*)\s*dB)", std::regex::icase};
  std::regex q{R"(Q\s+([\d.]+))", std::regex::icase};
  std::regex bwOct{R"(BW\s+Oct\s+([-+]?\d+\.?\d*))", std::regex::icase};
  std::regex bwHz{R"(BW\s+([-+]?\d+\.?\d*)\s*(?:Hz)?)", std::regex::icase};
};

bool Find(const std::string &line, const std::regex &re, int group, double *value) {
  std::smatch m;
  if (!std::regex_search(line, m, re)) {
    return false;
  }
  *value = std::stod(m[group].str());
  return true;
}

}  // namespace

std::size_t EqProfile::activeBandCount() const {
  return static_cast<std::size_t>(std::count_if(bands.begin(), bands.end(), [](const EqBand &b) { return b.enabled; }));
}

const char *filterTypeName(FilterType type) {
  for (const auto &e : kTypes) {
    if (e.type == type) {
      return e.label;
    }
  }
  return "??";
}

FilterType parseFilterType(const std::string &typeStr) {
  const std::string up = Upper(typeStr);
  for (const auto &e : kTypes) {
    for (const char *s : e.spellings) {
      if (up == s) {
        return e.type;
      }
    }
  }
  return FilterType::PK;  // unknown spellings fall back to peaking
}

bool parseEqString(const std::string &content, EqProfile &profile) {
  static const Grammar g;
  profile.bands.clear();
  profile.preampDb = 0.0;

  std::istringstream in(content);
  for (std::string raw; std::getline(in, raw);) {
    const std::string line = Strip(raw);
    if (line.empty() || line[0] == '#' || line[0] == ';') {
      continue;
    }
    double value = 0.0;
    if (Find(line, g.preamp, 1, &value)) {
      profile.preampDb = value;
      continue;
    }
    std::smatch m;
    if (!std::regex_search(line, m, g.filter)) {
      continue;
    }
    EqBand band;
    band.enabled = Upper(m[2].str()) == "ON";
    band.type = parseFilterType(Strip(m[3].str()));
    band.frequency = std::stod(m[4].str());
    band.gain = Find(line, g.gain, 1, &value) ? value : 0.0;
    const bool qGiven = Find(line, g.q, 1, &value);
    band.q = qGiven ? value : 1.0;
    if (Find(line, g.bwOct, 1, &value)) {
      band.hasBandwidthOct = true;
      band.bandwidthOct = value;
      if (!qGiven) {
        band.q = QFromOctaves(value);
      }
    }
    if (Find(line, g.bwHz, 1, &value)) {
      band.hasBandwidthHz = true;
      band.bandwidthHz = value;
      if (!qGiven && !band.hasBandwidthOct) {
        band.q = QFromHz(band.frequency, value);
      }
    }
    profile.bands.push_back(band);
  }
  return !profile.bands.empty() || profile.preampDb != 0.0;
}

bool parseEqFile(const std::string &filePath, EqProfile &profile) {
  std::ifstream file(filePath);
  if (!file.is_open()) {
    std::cerr << "EQ Parser: Cannot open file: " << filePath << '\n';
    return false;
  }
  // profile name = file stem (eq_parser.cpp:268-281)
  std::size_t begin = filePath.find_last_of("/\\");
  begin = (begin == std::string::npos) ? 0 : begin + 1;
  const std::size_t dot = filePath.find_last_of('.');
  profile.name = (dot != std::string::npos && dot > begin) ? filePath.substr(begin, dot - begin) : filePath.substr(begin);

  const std::string text((std::istreambuf_iterator<char>(file)), std::istreambuf_iterator<char>());
  const bool ok = parseEqString(text, profile);
  if (ok) {
    std::cout << "EQ Parser: Loaded '" << profile.name << "' with " << profile.activeBandCount()
              << " active bands, preamp " << profile.preampDb << " dB" << '\n';
  }
  return ok;
}

// RBJ cookbook sections, normalised by a0 (eq_to_fir.cpp:9-75).
BiquadCoeffs calculateBiquadCoeffs(const EqBand &band, double sampleRate) {
  const BiquadCoeffs bypass{1.0, 0.0, 0.0, 0.0, 0.0};
  if (!band.enabled || band.gain == 0.0) {
    return bypass;
  }
  const double A = std::pow(10.0, band.gain / 40.0);
  const double w0 = 2.0 * kPi * band.frequency / sampleRate;
  const double cw = std::cos(w0), sw = std::sin(w0);
  const double alpha = sw / (2.0 * band.q);
  double b0, b1, b2, a0, a1, a2;
  if (band.type == FilterType::PK) {
    b0 = 1.0 + alpha * A;
    b1 = -2.0 * cw;
    b2 = 1.0 - alpha * A;
    a0 = 1.0 + alpha / A;
    a1 = -2.0 * cw;
    a2 = 1.0 - alpha / A;
  } else if (band.type == FilterType::LS || band.type == FilterType::HS) {
    // low and high shelf differ only in the sign of the cos terms
    const double sgn = band.type == FilterType::LS ? 1.0 : -1.0;
    const double beta = 2.0 * std::sqrt(A) * alpha;
    const double ap = A + 1.0, am = A - 1.0;
    b0 = A * (ap - sgn * am * cw + beta);
    b1 = sgn * 2.0 * A * (am - sgn * ap * cw);
    b2 = A * (ap - sgn * am * cw - beta);
    a0 = ap + sgn * am * cw + beta;
    a1 = -sgn * 2.0 * (am + sgn * ap * cw);
    a2 = ap + sgn * am * cw - beta;
  } else {
    std::cerr << "EQ: Filter type " << filterTypeName(band.type) << " not implemented, using bypass" << '\n';
    return bypass;
  }
  return BiquadCoeffs{b0 / a0, b1 / a0, b2 / a0, a1 / a0, a2 / a0};
}

Cascade buildCascade(const EqProfile &profile, double sampleRate) {
  Cascade c;
  if (profile.preampDb != 0.0) {
    c.preampLinear = std::pow(10.0, profile.preampDb / 20.0);
  }
  for (const auto &band : profile.bands) {
    if (band.enabled) {
      c.sections.push_back(calculateBiquadCoeffs(band, sampleRate));
    }
  }
  return c;
}

std::vector<std::complex<double>> ComputeEqResponseHost(std::size_t numBins, std::size_t fullFftSize,
                                                        double outputSampleRate, const EqProfile &profile) {
  const Cascade c = buildCascade(profile, outputSampleRate);
  const double df = outputSampleRate / static_cast<double>(fullFftSize);
  std::vector<std::complex<double>> out(numBins);
  for (std::size_t i = 0; i < numBins; ++i) {
    const double w = 2.0 * kPi * (static_cast<double>(i) * df) / outputSampleRate;
    const std::complex<double> z = std::exp(std::complex<double>(0.0, -w));
    const std::complex<double> z2 = z * z;
    std::complex<double> r(1.0, 0.0);
    if (profile.preampDb != 0.0) {
      r *= c.preampLinear;
    }
    for (const auto &s : c.sections) {
      r *= (s.b0 + s.b1 * z + s.b2 * z2) / (1.0 + s.a1 * z + s.a2 * z2);
    }
    out[i] = r;
  }
  return out;
}

std::vector<double> ComputeEqMagnitudeHost(std::size_t numBins, std::size_t fullFftSize, double outputSampleRate,
                                           const EqProfile &profile) {
  const auto resp = ComputeEqResponseHost(numBins, fullFftSize, outputSampleRate, profile);
  std::vector<double> mag(resp.size());
  double peak = 0.0;
  for (std::size_t i = 0; i < resp.size(); ++i) {
    mag[i] = std::abs(resp[i]);
    peak = std::max(peak, mag[i]);
  }
  if (peak > 1.0) {
    for (double &m : mag) {
      m *= 1.0 / peak;
    }
  }
  return mag;
}

}  // namespace miups::eq
