#include "eq.h"

#include "host_fft.h"

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

EqFold FoldCascadeIntoTaps(const std::vector<float> &taps, const Cascade &cascade) {
  EqFold out;
  const std::size_t n = taps.size();
  out.fir.assign(n, 0.0);
  std::vector<BiquadCoeffs> secs;  // bypass sections are the identity
  for (const auto &s : cascade.sections) {
    if (!(s.b0 == 1.0 && s.b1 == 0.0 && s.b2 == 0.0 && s.a1 == 0.0 && s.a2 == 0.0)) {
      secs.push_back(s);
    }
  }
  const std::size_t ns = secs.size();
  // transposed direct form II per section: y = b0 x + s1; s1 = b1 x - a1 y + s2; s2 = b2 x - a2 y  (fp64: within 2e-11 of
  // an 80-bit recursion at the worst corner the reference's validator accepts, Fc 10 Hz Q 100 at 768 kHz)
  std::vector<double> s1(ns, 0.0), s2(ns, 0.0);
  auto step = [&](double x) {
    for (std::size_t i = 0; i < ns; ++i) {
      const BiquadCoeffs &c = secs[i];
      const double y = c.b0 * x + s1[i];
      s1[i] = c.b1 * x - c.a1 * y + s2[i];
      s2[i] = c.b2 * x - c.a2 * y;
      x = y;
    }
    return x;
  };
  out.taper = n > 1 ? (n - 1) / 64 : 0;
  double l1All = 0.0, l2All = 0.0, l1Cut = 0.0, l2Cut = 0.0;
  for (std::size_t i = 0; i < n; ++i) {
    const double v = step(static_cast<double>(taps[i]) * cascade.preampLinear);
    double w = 1.0;
    if (i + out.taper >= n) {
      const double u = (static_cast<double>(i - (n - out.taper)) + 0.5) / static_cast<double>(out.taper);
      w = 0.5 * (1.0 + std::cos(kPi * u));
    }
    out.fir[i] = v * w;
    const double d = v - out.fir[i];
    l1All += std::fabs(v);
    l2All += v * v;
    l1Cut += std::fabs(d);
    l2Cut += d * d;
  }
  if (ns > 0 && l1All > 0.0) {
    const std::size_t chunk = 4096;
    const std::size_t maxChunks = static_cast<std::size_t>(kFoldTailWork / static_cast<double>(ns * chunk)) + 1;
    double last = 0.0;
    bool dead = false;
    for (std::size_t c = 0; c < maxChunks && !dead; ++c) {
      double a1 = 0.0, a2 = 0.0;
      for (std::size_t i = 0; i < chunk; ++i) {
        const double v = step(0.0);
        a1 += std::fabs(v);
        a2 += v * v;
      }
      l1All += a1;
      l2All += a2;
      l1Cut += a1;
      l2Cut += a2;
      last = a1;
      dead = a1 <= 1e-17 * l1All;
    }
    if (!dead) {
      out.tailComplete = false;
      double rmax = 0.0;  // largest pole radius of the cascade
      for (const auto &c : secs) {
        const double disc = c.a1 * c.a1 - 4.0 * c.a2;
        const double r = disc < 0.0 ? std::sqrt(std::max(c.a2, 0.0))
                                    : std::max(std::fabs(-c.a1 + std::sqrt(disc)), std::fabs(-c.a1 - std::sqrt(disc))) / 2.0;
        rmax = std::max(rmax, r);
      }
      const double rho = std::pow(std::min(rmax, 1.0), static_cast<double>(chunk));
      const double rest = rho < 1.0 - 1e-12 ? last * rho / (1.0 - rho) : 1e300;
      l1All += rest;
      l1Cut += rest;
      const double rest2 = rho < 1.0 - 1e-12 ? (last * last / static_cast<double>(chunk)) * rho * rho / (1.0 - rho * rho) : 1e300;
      l2All += rest2;
      l2Cut += rest2;
    }
  }
  out.tailL1 = l1All > 0.0 ? l1Cut / l1All : 0.0;
  out.tailL2 = l2All > 0.0 ? std::sqrt(l2Cut / l2All) : 0.0;
  return out;
}

double ResponseDeviation(const std::vector<float> &taps, const std::vector<double> &fir,
                         const std::vector<std::complex<double>> &eqHalf, std::size_t fftSize) {
  const std::size_t N = fftSize;
  if (N < 2 || eqHalf.size() != N / 2 + 1 || taps.size() > N || fir.size() > N) {
    return 0.0;
  }
  std::vector<std::complex<double>> z(N, std::complex<double>(0.0, 0.0));
  for (std::size_t i = 0; i < taps.size(); ++i) {
    z[i] = std::complex<double>(static_cast<double>(taps[i]), 0.0);
  }
  for (std::size_t i = 0; i < fir.size(); ++i) {
    z[i] += std::complex<double>(0.0, fir[i]);
  }
  miups::Fft64(z, false);
  double worst = 0.0, peak = 0.0;
  for (std::size_t k = 0; k <= N / 2; ++k) {
    const std::complex<double> a = z[k], b = std::conj(z[(N - k) % N]);
    const std::complex<double> hFir = 0.5 * (a + b);                                  // spectrum of the real part
    const std::complex<double> hUsed = std::complex<double>(0.0, -0.5) * (a - b);     // ... of the imaginary part
    const std::complex<double> ideal = hFir * eqHalf[k];
    worst = std::max(worst, std::abs(hUsed - ideal));
    peak = std::max(peak, std::abs(ideal));
  }
  return peak > 0.0 ? worst / peak : 0.0;
}

std::string FoldWarning(std::size_t firTaps, double tailL1, double tailL2, bool tailComplete, double limit) {
  if (!(tailL1 > limit)) {
    return std::string();
  }
  auto db = [](double v) { return v > 0.0 ? 20.0 * std::log10(v) : -400.0; };
  std::ostringstream m;
  m.setf(std::ios::fixed);
  m.precision(1);
  m << "EQ cut to " << firTaps << " taps drops " << db(tailL1) << " dB of the ideal cascade response (limit "
    << db(limit) << " dB, 2-norm " << db(tailL2) << " dB" << (tailComplete ? "" : ", tail estimated")
    << "): a band this low or this narrow rings longer than the filter is; use a longer filter or a wider band";
  return m.str();
}

}  // namespace miups::eq
