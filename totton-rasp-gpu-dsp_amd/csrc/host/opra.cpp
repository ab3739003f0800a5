#include "opra.h"

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <utility>
#include <vector>

namespace miups {
namespace {

// A JSON value tree (objects keep their members in order; duplicate keys: the last one wins, as in Python's json).
struct Value {
  enum Kind { kNull, kBool, kNumber, kString, kArray, kObject } kind = kNull;
  double number = 0.0;
  bool boolean = false;
  std::string text;
  std::vector<Value> items;
  std::vector<std::pair<std::string, Value>> members;

  const Value *Get(const std::string &key) const {
    const Value *hit = nullptr;
    for (const auto &m : members) {
      if (m.first == key) {
        hit = &m.second;
      }
    }
    return hit;
  }
};

class Parser {
 public:
  explicit Parser(const std::string &s) : s_(s) {}
  bool Parse(Value *out, std::string *error) {
    Ws();
    if (!ParseValue(out, 0)) {
      *error = error_;
      return false;
    }
    Ws();
    if (i_ != s_.size()) {
      *error = "trailing characters at offset " + std::to_string(i_);
      return false;
    }
    return true;
  }

 private:
  const std::string &s_;
  std::size_t i_ = 0;
  std::string error_;

  void Ws() {
    while (i_ < s_.size() && std::isspace(static_cast<unsigned char>(s_[i_]))) {
      ++i_;
    }
  }
  bool Fail(const std::string &m) {
    if (error_.empty()) {
      error_ = m + " at offset " + std::to_string(i_);
    }
    return false;
  }
  bool Literal(const char *word) {
    std::size_t n = 0;
    while (word[n]) {
      ++n;
    }
    if (s_.compare(i_, n, word) != 0) {
      return Fail("unexpected token");
    }
    i_ += n;
    return true;
  }
  bool ParseString(std::string *v) {
    ++i_;  // opening quote
    v->clear();
    while (i_ < s_.size() && s_[i_] != '"') {
      char c = s_[i_++];
      if (c == '\\') {
        if (i_ >= s_.size()) {
          return Fail("bad escape");
        }
        const char e = s_[i_++];
        switch (e) {
          case 'n': c = '\n'; break;
          case 't': c = '\t'; break;
          case 'r': c = '\r'; break;
          case 'b': c = '\b'; break;
          case 'f': c = '\f'; break;
          case 'u': {
            if (i_ + 4 > s_.size()) {
              return Fail("bad \\u escape");
            }
            const unsigned code = static_cast<unsigned>(std::strtoul(s_.substr(i_, 4).c_str(), nullptr, 16));
            i_ += 4;
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
    if (i_ >= s_.size()) {
      return Fail("unterminated string");
    }
    ++i_;
    return true;
  }
  bool ParseValue(Value *v, int depth) {
    if (depth > 64) {
      return Fail("nesting too deep");
    }
    if (i_ >= s_.size()) {
      return Fail("unexpected end");
    }
    const char c = s_[i_];
    if (c == '{') {
      v->kind = Value::kObject;
      ++i_;
      Ws();
      if (i_ < s_.size() && s_[i_] == '}') {
        ++i_;
        return true;
      }
      for (;;) {
        Ws();
        if (i_ >= s_.size() || s_[i_] != '"') {
          return Fail("expected member name");
        }
        std::string key;
        if (!ParseString(&key)) {
          return false;
        }
        Ws();
        if (i_ >= s_.size() || s_[i_] != ':') {
          return Fail("expected ':'");
        }
        ++i_;
        Ws();
        Value member;
        if (!ParseValue(&member, depth + 1)) {
          return false;
        }
        v->members.emplace_back(std::move(key), std::move(member));
        Ws();
        if (i_ < s_.size() && s_[i_] == ',') {
          ++i_;
          continue;
        }
        if (i_ < s_.size() && s_[i_] == '}') {
          ++i_;
          return true;
        }
        return Fail("expected ',' or '}'");
      }
    }
    if (c == '[') {
      v->kind = Value::kArray;
      ++i_;
      Ws();
      if (i_ < s_.size() && s_[i_] == ']') {
        ++i_;
        return true;
      }
      for (;;) {
        Ws();
        Value item;
        if (!ParseValue(&item, depth + 1)) {
          return false;
        }
        v->items.push_back(std::move(item));
        Ws();
        if (i_ < s_.size() && s_[i_] == ',') {
          ++i_;
          continue;
        }
        if (i_ < s_.size() && s_[i_] == ']') {
          ++i_;
          return true;
        }
        return Fail("expected ',' or ']'");
      }
    }
    if (c == '"') {
      v->kind = Value::kString;
      return ParseString(&v->text);
    }
    if (c == 't') {
      v->kind = Value::kBool;
      v->boolean = true;
      return Literal("true");
    }
    if (c == 'f') {
      v->kind = Value::kBool;
      return Literal("false");
    }
    if (c == 'n') {
      v->kind = Value::kNull;
      return Literal("null");
    }
    char *end = nullptr;
    const double d = std::strtod(s_.c_str() + i_, &end);
    if (end == s_.c_str() + i_) {
      return Fail("unexpected character");
    }
    v->kind = Value::kNumber;
    v->number = d;
    i_ = static_cast<std::size_t>(end - s_.c_str());
    return true;
  }
};

// number member or the default (absent, null or not a number)
double NumberOr(const Value &obj, const char *key, double fallback, bool *present = nullptr) {
  const Value *v = obj.Get(key);
  const bool ok = v && v->kind == Value::kNumber;
  if (present) {
    *present = ok;
  }
  return ok ? v->number : fallback;
}

// scripts/integration/opra.py:103-125
double SlopeToQ(double slope) {
  if (slope == 6) return 0.5;
  if (slope == 12) return 0.707;
  if (slope == 18) return 0.5;
  if (slope == 24) return 0.541;
  if (slope == 30) return 0.5;
  if (slope == 36) return 0.518;
  return 0.707;
}

struct Band {
  const char *type;
  double frequency, gain, q;
};

std::string Fixed(double x, int digits) {
  char buf[64];
  std::snprintf(buf, sizeof(buf), "%.*f", digits, x);  // correctly rounded, like Python's format(x, ".1f")
  return buf;
}

}  // namespace

bool OpraToApo(const std::string &eqJson, bool modernTarget, std::string *apoText, std::string *error) {
  Value root;
  std::string perr;
  if (!Parser(eqJson).Parse(&root, &perr)) {
    if (error) {
      *error = "OPRA record: " + perr;
    }
    return false;
  }
  if (root.kind != Value::kObject) {
    if (error) {
      *error = "OPRA record: not a JSON object";
    }
    return false;
  }
  const Value empty;
  const Value *params = root.Get("parameters");
  if (!params || params->kind != Value::kObject) {
    params = &empty;
  }
  const Value *bandsV = params->Get("bands");
  std::vector<Band> bands;
  if (bandsV && bandsV->kind == Value::kArray) {
    for (const Value &b : bandsV->items) {
      if (b.kind != Value::kObject) {
        continue;
      }
      const Value *t = b.Get("type");
      const std::string type = (t && t->kind == Value::kString) ? t->text : "";
      const char *apo = type == "peak_dip"     ? "PK"
                        : type == "low_shelf"  ? "LS"
                        : type == "high_shelf" ? "HS"
                        : type == "low_pass"   ? "LP"
                        : type == "high_pass"  ? "HP"
                                               : nullptr;
      if (!apo) {
        continue;  // band_pass, band_stop and anything unknown are dropped (:150-153)
      }
      Band band{apo, NumberOr(b, "frequency", 1000.0), NumberOr(b, "gain_db", 0.0), 1.0};
      bool hasQ = false, hasSlope = false;
      const double q = NumberOr(b, "q", 1.0, &hasQ), slope = NumberOr(b, "slope", 0.0, &hasSlope);
      if ((apo[0] == 'L' || apo[0] == 'H') && apo[1] == 'P') {
        band.q = hasSlope ? SlopeToQ(slope) : 0.707;  // (:156-162)
        band.gain = 0.0;
      } else {
        band.q = hasQ ? q : 1.0;
      }
      bands.push_back(band);
    }
  }
  double preamp = NumberOr(*params, "gain_db", 0.0);
  if (modernTarget) {  // scripts/modern_target.py:43-49, opra.py:224-236
    bands.push_back(Band{"PK", 5366.0, 2.8, 1.5});
    preamp -= 2.8;
  }
  std::string out;
  auto line = [&out](const std::string &l) {
    if (!out.empty()) {
      out += "\n";
    }
    out += l;
  };
  if (preamp != 0.0) {
    line("Preamp: " + Fixed(preamp, 1) + " dB");
  }
  int n = 0;
  for (const Band &b : bands) {
    ++n;
    const bool pass = (b.type[1] == 'P' && (b.type[0] == 'L' || b.type[0] == 'H'));
    std::string l = "Filter " + std::to_string(n) + ": ON " + b.type + " Fc " + Fixed(b.frequency, 1) + " Hz";
    if (!pass) {
      l += " Gain " + Fixed(b.gain, 1) + " dB";
    }
    l += " Q " + Fixed(b.q, 2);
    line(l);
  }
  *apoText = out;
  if (error) {
    error->clear();
  }
  return true;
}

}  // namespace miups
