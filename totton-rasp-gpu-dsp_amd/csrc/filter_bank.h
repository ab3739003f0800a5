// Resident spectra for every (rate family, ratio, phase) a filter directory offers, so that a change of input rate or
// of the phase setting is a pointer switch between blocks instead of a file load + table build.
//
// Reference: the selector keys a filter by family, ratio and phase (src/alsa/alsa_filter_selector.cpp:33-55: family
// from rate % 44100 / % 48000, "min"/"linear", ratio, highest tap count wins) and the negotiation keeps the output
// rate of a family fixed while the input rate -- hence the ratio -- changes (src/audio/auto_negotiation.cpp:72-155,
// "same-family switching is instant and glitch-free"). The bank resolves each key with the selector's own rule at
// load time and keeps the resulting DeviceFilter resident: 8 shipped geometries x ~1.5 MiB of tables per GPU.
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "engine.h"

namespace miups {

class FilterBank {
 public:
  struct Entry {
    unsigned familyBaseRate = 0;  // 44100 or 48000
    unsigned ratio = 0;
    std::string phase;            // "min" | "linear"
    std::string path;             // the sidecar the selector picked
    std::shared_ptr<DeviceFilter> filter;
  };
  // Loads every key the directory can serve. Fails only when the directory is missing or NOTHING could be loaded;
  // sidecars that fail to load are reported in *warnings (one line each) and skipped.
  static std::unique_ptr<FilterBank> Load(int device, const std::string &dir, std::string *warnings, std::string *error);
  const std::vector<Entry> &entries() const { return entries_; }
  // The resident filter for this input rate / ratio / phase, or null with the selector's message.
  const Entry *Find(unsigned inputRate, unsigned ratio, const std::string &phase, std::string *error) const;

 private:
  std::vector<Entry> entries_;
  std::string dir_;
};

}  // namespace miups
