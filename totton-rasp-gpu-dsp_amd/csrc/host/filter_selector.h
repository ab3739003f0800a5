// --filter / --filter-dir / --ratio / --phase resolution for the streamer CLI.
// Same rule and messages as the reference's ResolveFilterPath
// (include/alsa/alsa_filter_selector.h:8-15, src/alsa/alsa_filter_selector.cpp:8-108).
#pragma once

#include <optional>
#include <string>

namespace miups {

struct FilterSelection {
  std::string path;
};

std::optional<FilterSelection> ResolveFilterPath(const std::string &filterPath, const std::string &filterDir,
                                                 const std::string &phase, unsigned int ratio, unsigned int inputRate,
                                                 std::string *errorMessage);

}  // namespace miups
