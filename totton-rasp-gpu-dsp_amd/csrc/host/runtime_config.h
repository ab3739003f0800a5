// The daemon-side view of config.json (reference: release/config.example.json, web/models.py:9-37): which EQ profile is
// active and which filter the streamer should run. The web UI writes the file and sends RELOAD
// (web/routers/eq.py:220-273); here the streamer re-reads it on SIGHUP or when its mtime changes.
#pragma once

#include <string>

namespace miups {

struct RuntimeConfig {
  bool eqEnabled = false;
  std::string eqProfile;      // name, informational
  std::string eqProfilePath;  // Equalizer-APO text file
  // "filter" section (0 / empty = not given)
  unsigned ratio = 0;
  std::string phaseType;      // "minimum" | "linear"
  std::string filterDirectory;
  // "alsa" section
  unsigned sampleRate = 0, channels = 0, periodFrames = 0, bufferFrames = 0;
  std::string format, inputDevice, outputDevice;
};

// Parses the JSON text. Unknown keys are ignored, missing keys keep their defaults, `null` is "not set".
bool ParseRuntimeConfig(const std::string &jsonText, RuntimeConfig *out, std::string *error);
bool LoadRuntimeConfig(const std::string &path, RuntimeConfig *out, std::string *error);

// "minimum" -> "min", "linear" -> "linear" (the --phase spelling of the streamer); anything else is returned as is
std::string PhaseFlagFromConfig(const std::string &phaseType);
// "S16_LE" / "S24_3LE" / "S32_LE" -> "s16" / "s24" / "s32" (the --format spelling)
std::string FormatFlagFromConfig(const std::string &format);

// Tells when the file changed since the last look (mtime + size); a missing file counts as "unchanged".
class ConfigWatcher {
 public:
  explicit ConfigWatcher(std::string path);
  bool Changed();
  const std::string &path() const { return path_; }

 private:
  std::string path_;
  long long mtimeNs_ = -1, size_ = -1;
};

}  // namespace miups
