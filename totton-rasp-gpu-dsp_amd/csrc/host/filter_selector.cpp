#include "filter_selector.h"

#include <cctype>
#include <filesystem>

namespace miups {
namespace fs = std::filesystem;
namespace {

// "<digits>" -> value, "2m" -> 640000 (legacy name of the 640k-tap filters),
// anything else -> 0 (entry ignored)
unsigned int TapsFromToken(const std::string &token) {
  if (token == "2m") {
    return 640000;
  }
  if (token.empty()) {
    return 0;
  }
  unsigned long v = 0;
  for (char c : token) {
    if (!std::isdigit(static_cast<unsigned char>(c))) {
      return 0;
    }
    v = v * 10 + static_cast<unsigned long>(c - '0');
    if (v > 0xffffffffUL) {
      return 0;
    }
  }
  return static_cast<unsigned int>(v);
}

std::nullopt_t Miss(std::string *errorMessage, const std::string &message) {
  if (errorMessage) {
    *errorMessage = message;
  }
  return std::nullopt;
}

}  // namespace

std::optional<FilterSelection> ResolveFilterPath(const std::string &filterPath, const std::string &filterDir,
                                                 const std::string &phase, unsigned int ratio, unsigned int inputRate,
                                                 std::string *errorMessage) {
  // an explicit --filter always wins
  if (!filterPath.empty()) {
    if (!fs::exists(filterPath)) {
      return Miss(errorMessage, "Filter file not found: " + filterPath);
    }
    return FilterSelection{filterPath};
  }
  if (filterDir.empty()) {
    return std::nullopt;
  }
  if (!fs::exists(filterDir)) {
    return Miss(errorMessage, "Filter directory not found: " + filterDir);
  }
  const char *family = (inputRate % 44100 == 0) ? "44" : ((inputRate % 48000 == 0) ? "48" : nullptr);
  if (!family) {
    return Miss(errorMessage, "Unsupported input rate family: " + std::to_string(inputRate));
  }
  const std::string phaseName = phase == "min" ? "min_phase" : (phase == "linear" ? "linear_phase" : phase);
  const std::string head = std::string("filter_") + family + "k_" + std::to_string(ratio) + "x_";
  const std::string tail = "_" + phaseName + ".json";

  // filter_{44|48}k_{ratio}x_{taps}_{phase}.json -- the largest tap count wins
  unsigned int best = 0;
  fs::path bestPath;
  for (const auto &entry : fs::directory_iterator(filterDir)) {
    if (!entry.is_regular_file()) {
      continue;
    }
    const std::string name = entry.path().filename().string();
    if (name.size() <= head.size() + tail.size() || name.compare(0, head.size(), head) != 0 ||
        name.compare(name.size() - tail.size(), tail.size(), tail) != 0) {
      continue;
    }
    const unsigned int taps = TapsFromToken(name.substr(head.size(), name.size() - head.size() - tail.size()));
    if (taps > best) {
      best = taps;
      bestPath = entry.path();
    }
  }
  if (best == 0) {
    return Miss(errorMessage, "Filter file not found: " + filterDir + "/" + head + "*" + tail);
  }
  return FilterSelection{bestPath.string()};
}

}  // namespace miups
