#include "filter_bank.h"

#include <sys/stat.h>

#include "host/filter_config.h"
#include "host/filter_selector.h"

namespace miups {

std::unique_ptr<FilterBank> FilterBank::Load(int device, const std::string &dir, std::string *warnings,
                                             std::string *error) {
  struct stat st;
  if (::stat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) {
    if (error) {
      *error = "Filter directory not found: " + dir;
    }
    return nullptr;
  }
  std::unique_ptr<FilterBank> bank(new FilterBank());
  bank->dir_ = dir;
  const unsigned bases[2] = {44100, 48000};
  const unsigned ratios[5] = {1, 2, 4, 8, 16};
  const char *phases[2] = {"min", "linear"};
  std::string lastError;
  for (unsigned base : bases) {
    for (unsigned ratio : ratios) {
      for (const char *phase : phases) {
        std::string err;
        const auto sel = ResolveFilterPath("", dir, phase, ratio, base, &err);
        if (!sel) {
          continue;  // the directory does not serve this key
        }
        const std::string path = sel->path;
        FilterConfig config;
        std::vector<float> taps;
        if (!ReadFilter(path, &config, &taps, &err)) {
          if (warnings) {
            *warnings += path + ": " + err + "\n";
          }
          lastError = err;
          continue;
        }
        auto filter = DeviceFilter::Create(device, config, std::move(taps), kLoadDefault, &err);
        if (!filter) {
          if (warnings) {
            *warnings += path + ": " + err + "\n";
          }
          lastError = err;
          continue;
        }
        Entry e;
        e.familyBaseRate = base;
        e.ratio = ratio;
        e.phase = phase;
        e.path = path;
        e.filter = std::move(filter);
        bank->entries_.push_back(std::move(e));
      }
    }
  }
  if (bank->entries_.empty()) {
    if (error) {
      *error = lastError.empty() ? "No filter found in " + dir : lastError;
    }
    return nullptr;
  }
  return bank;
}

const FilterBank::Entry *FilterBank::Find(unsigned inputRate, unsigned ratio, const std::string &phase,
                                          std::string *error) const {
  unsigned base = 0;
  if (inputRate != 0 && inputRate % 44100 == 0) {
    base = 44100;
  } else if (inputRate != 0 && inputRate % 48000 == 0) {
    base = 48000;
  } else {
    if (error) {
      *error = "Unsupported input rate family: " + std::to_string(inputRate);
    }
    return nullptr;
  }
  if (phase != "min" && phase != "linear") {
    if (error) {
      *error = "Unsupported phase: " + phase;
    }
    return nullptr;
  }
  for (const Entry &e : entries_) {
    if (e.familyBaseRate == base && e.ratio == ratio && e.phase == phase) {
      return &e;
    }
  }
  if (error) {
    *error = "Filter file not found: " + dir_ + "/filter_" + std::string(base == 44100 ? "44k" : "48k") + "_" +
             std::to_string(ratio) + "x_*_" + (phase == "min" ? "min_phase" : "linear_phase") + ".json";
  }
  return nullptr;
}

}  // namespace miups
