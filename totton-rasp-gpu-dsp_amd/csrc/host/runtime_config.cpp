#include "runtime_config.h"

#include <sys/stat.h>

#include <cctype>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>

namespace miups {
namespace {

// A small recursive-descent JSON reader: objects, arrays, strings, numbers, true/false/null. Values are flattened to
// "section<SEP>key" -> text (strings unescaped, null -> empty marker) which is all the config needs. SEP is a control
// character, so a top-level key that merely CONTAINS a dot cannot pose as the nested one; the nesting depth is bounded
// (the file is re-read on SIGHUP while audio is streaming: a hostile or damaged file must produce an error message,
// not a stack overflow).
constexpr char kSep = '\x1f';
constexpr int kMaxDepth = 64;
std::string Key(const char *section, const char *key) { return std::string(section) + kSep + key; }

struct Reader {
  const std::string &s;
  std::size_t i = 0;
  std::string error;
  std::map<std::string, std::string> *out;
  std::map<std::string, bool> *isNull;

  void Ws() {
    while (i < s.size() && std::isspace(static_cast<unsigned char>(s[i]))) {
      ++i;
    }
  }
  bool Fail(const std::string &m) {
    if (error.empty()) {
      error = m + " at offset " + std::to_string(i);
    }
    return false;
  }
  bool String(std::string *v) {
    if (i >= s.size() || s[i] != '"') {
      return Fail("expected string");
    }
    ++i;
    v->clear();
    while (i < s.size() && s[i] != '"') {
      char c = s[i++];
      if (c == '\\') {
        if (i >= s.size()) {
          return Fail("bad escape");
        }
        const char e = s[i++];
        switch (e) {
          case 'n': c = '\n'; break;
          case 't': c = '\t'; break;
          case 'r': c = '\r'; break;
          case 'b': c = '\b'; break;
          case 'f': c = '\f'; break;
          case 'u': {
            if (i + 4 > s.size()) {
              return Fail("bad \\u escape");
            }
            const unsigned code = static_cast<unsigned>(std::strtoul(s.substr(i, 4).c_str(), nullptr, 16));
            i += 4;
            if (code < 0x80) {
              c = static_cast<char>(code);
            } else if (code < 0x800) {
              v->push_back(static_cast<char>(0xC0 | (code >> 6)));
              c = static_cast<char>(0x80 | (code & 0x3F));
            } else {
              v->push_back(static_cast<char>(0xE0 | (code >> 12)));
              v->push_back(static_cast<char>(0x80 | ((code >> 6) & 0x3F)));
              c = static_cast<char>(0x80 | (code & 0x3F));
            }
            break;
          }
          default: c = e; break;  // \" \\ \/
        }
      }
      v->push_back(c);
    }
    if (i >= s.size()) {
      return Fail("unterminated string");
    }
    ++i;
    return true;
  }
  bool Value(const std::string &path, int depth = 0) {
    Ws();
    if (i >= s.size()) {
      return Fail("unexpected end");
    }
    if (depth > kMaxDepth) {
      return Fail("nesting deeper than " + std::to_string(kMaxDepth) + " levels");
    }
    const char c = s[i];
    if (c == '{') {
      ++i;
      Ws();
      if (i < s.size() && s[i] == '}') {
        ++i;
        return true;
      }
      for (;;) {
        Ws();
        std::string key;
        if (!String(&key)) {
          return false;
        }
        Ws();
        if (i >= s.size() || s[i] != ':') {
          return Fail("expected ':'");
        }
        ++i;
        if (!Value(path.empty() ? key : path + kSep + key, depth + 1)) {
          return false;
        }
        Ws();
        if (i < s.size() && s[i] == ',') {
          ++i;
          continue;
        }
        if (i < s.size() && s[i] == '}') {
          ++i;
          return true;
        }
        return Fail("expected ',' or '}'");
      }
    }
    if (c == '[') {
      ++i;
      Ws();
      if (i < s.size() && s[i] == ']') {
        ++i;
        return true;
      }
      for (std::size_t n = 0;; ++n) {
        if (!Value(path + "[" + std::to_string(n) + "]", depth + 1)) {
          return false;
        }
        Ws();
        if (i < s.size() && s[i] == ',') {
          ++i;
          continue;
        }
        if (i < s.size() && s[i] == ']') {
          ++i;
          return true;
        }
        return Fail("expected ',' or ']'");
      }
    }
    if (c == '"') {
      std::string v;
      if (!String(&v)) {
        return false;
      }
      (*out)[path] = v;
      return true;
    }
    const std::size_t start = i;
    while (i < s.size() && s[i] != ',' && s[i] != '}' && s[i] != ']' && !std::isspace(static_cast<unsigned char>(s[i]))) {
      ++i;
    }
    const std::string tok = s.substr(start, i - start);
    if (tok.empty()) {
      return Fail("expected value");
    }
    if (tok == "null") {
      (*isNull)[path] = true;
      return true;
    }
    if (tok != "true" && tok != "false") {
      char *end = nullptr;
      std::strtod(tok.c_str(), &end);
      if (end == tok.c_str() || *end != '\0') {
        return Fail("bad token '" + tok + "'");
      }
    }
    (*out)[path] = tok;
    return true;
  }
};

unsigned ToUnsigned(const std::map<std::string, std::string> &m, const std::string &key) {
  const auto it = m.find(key);
  if (it == m.end()) {
    return 0;
  }
  const double v = std::strtod(it->second.c_str(), nullptr);
  return v > 0.0 && v < 4294967295.0 ? static_cast<unsigned>(v) : 0;
}

std::string ToString(const std::map<std::string, std::string> &m, const std::string &key) {
  const auto it = m.find(key);
  return it == m.end() ? std::string() : it->second;
}

}  // namespace

bool ParseRuntimeConfig(const std::string &jsonText, RuntimeConfig *out, std::string *error) {
  std::map<std::string, std::string> kv;
  std::map<std::string, bool> nulls;
  Reader r{jsonText, 0, std::string(), &kv, &nulls};
  r.Ws();
  if (r.i >= jsonText.size() || jsonText[r.i] != '{') {
    if (error) {
      *error = "config must be a JSON object";
    }
    return false;
  }
  if (!r.Value("")) {
    if (error) {
      *error = "config parse error: " + r.error;
    }
    return false;
  }
  r.Ws();
  if (r.i != jsonText.size()) {
    if (error) {
      *error = "config parse error: trailing characters";
    }
    return false;
  }
  RuntimeConfig c;
  c.eqEnabled = ToString(kv, "eqEnabled") == "true";
  c.eqProfile = ToString(kv, "eqProfile");
  c.eqProfilePath = ToString(kv, "eqProfilePath");
  c.ratio = ToUnsigned(kv, Key("filter", "ratio"));
  c.phaseType = ToString(kv, Key("filter", "phaseType"));
  c.filterDirectory = ToString(kv, Key("filter", "directory"));
  c.sampleRate = ToUnsigned(kv, Key("alsa", "sampleRate"));
  c.channels = ToUnsigned(kv, Key("alsa", "channels"));
  c.periodFrames = ToUnsigned(kv, Key("alsa", "periodFrames"));
  c.bufferFrames = ToUnsigned(kv, Key("alsa", "bufferFrames"));
  c.format = ToString(kv, Key("alsa", "format"));
  c.inputDevice = ToString(kv, Key("alsa", "inputDevice"));
  c.outputDevice = ToString(kv, Key("alsa", "outputDevice"));
  *out = c;
  return true;
}

bool LoadRuntimeConfig(const std::string &path, RuntimeConfig *out, std::string *error) {
  std::ifstream f(path);
  if (!f) {
    if (error) {
      *error = "Cannot open config: " + path;
    }
    return false;
  }
  std::stringstream ss;
  ss << f.rdbuf();
  return ParseRuntimeConfig(ss.str(), out, error);
}

std::string PhaseFlagFromConfig(const std::string &phaseType) {
  if (phaseType == "minimum" || phaseType == "min") {
    return "min";
  }
  return phaseType;
}

std::string FormatFlagFromConfig(const std::string &format) {
  if (format == "S16_LE") return "s16";
  if (format == "S24_3LE") return "s24";
  if (format == "S32_LE") return "s32";
  return format;
}

ConfigWatcher::ConfigWatcher(std::string path) : path_(std::move(path)) { (void)Changed(); }

bool ConfigWatcher::Changed() {
  struct stat st;
  if (path_.empty() || ::stat(path_.c_str(), &st) != 0) {
    return false;
  }
  const long long m = static_cast<long long>(st.st_mtim.tv_sec) * 1000000000ll + st.st_mtim.tv_nsec;
  const long long sz = static_cast<long long>(st.st_size);
  const bool changed = (mtimeNs_ >= 0) && (m != mtimeNs_ || sz != size_);
  mtimeNs_ = m;
  size_ = sz;
  return changed;
}

}  // namespace miups
