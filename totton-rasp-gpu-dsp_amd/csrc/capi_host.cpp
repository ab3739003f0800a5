// Host-only part of the C ABI (section (3) of include/mi_upsampler.h).
#include "../../include/mi_upsampler.h"

#include <algorithm>
#include <cctype>
#include <cstring>
#include <string>
#include <vector>

#include "device/pcm.h"
#include "host/eq.h"
#include "host/filter_config.h"
#include "host/filter_selector.h"
#include "host/spectrum.h"

struct mi_tables {
  miups::FilterTables tables;
};

namespace {

void Put(const std::string &m, char *dst, size_t cap) {
  if (dst && cap) {
    std::strncpy(dst, m.c_str(), cap - 1);
    dst[cap - 1] = '\0';
  }
}

const std::vector<miups::cf> *Pick(const mi_tables *t, int which) {
  switch (which) {
    case 0: return &t->tables.Gs;
    case 1: return &t->tables.Gc;
    case 2: return &t->tables.Wm;
    case 3: return &t->tables.tw;
    default: return nullptr;
  }
}

}  // namespace

extern "C" {

int mi_read_filter(const char *json_path, mi_ups_config *out, char *err, size_t errcap) {
  if (!json_path || !out) {
    Put("null argument", err, errcap);
    return MI_ERR_ARG;
  }
  miups::FilterConfig config;
  std::vector<float> taps;
  std::string error;
  if (!miups::ReadFilter(json_path, &config, &taps, &error)) {
    Put(error, err, errcap);
    return MI_ERR_FILTER;
  }
  out->taps = config.taps;
  out->fft_size = config.fftSize;
  out->block_size = config.blockSize;
  out->upsample_factor = config.upsampleFactor;
  Put(config.coefficientsPath, out->coefficients_path, sizeof(out->coefficients_path));
  Put("", err, errcap);
  return MI_OK;
}

int mi_resolve_filter_path(const char *filter_path, const char *filter_dir, const char *phase, unsigned ratio,
                           unsigned input_rate, char *out_path, size_t out_cap, char *err, size_t errcap) {
  std::string error;
  auto sel = miups::ResolveFilterPath(filter_path ? filter_path : "", filter_dir ? filter_dir : "",
                                      phase ? phase : "min", ratio, input_rate, &error);
  Put(error, err, errcap);
  if (!sel) {
    Put("", out_path, out_cap);
    return 0;
  }
  Put(sel->path, out_path, out_cap);
  return 1;
}

int mi_parse_format(const char *name) {
  if (!name) {
    return -1;
  }
  std::string low(name);
  std::transform(low.begin(), low.end(), low.begin(), [](unsigned char c) { return static_cast<char>(std::tolower(c)); });
  if (low == "s16" || low == "s16_le") {
    return MI_PCM_S16;
  }
  if (low == "s24" || low == "s24_3le") {
    return MI_PCM_S24_3LE;
  }
  if (low == "s32" || low == "s32_le") {
    return MI_PCM_S32;
  }
  return -1;
}

size_t mi_bytes_per_sample(int fmt) {
  switch (fmt) {
    case MI_PCM_S16: return 2;
    case MI_PCM_S24_3LE: return 3;
    case MI_PCM_S32: return 4;
    case MI_PCM_F32: return 4;
    default: return 0;
  }
}

int mi_pcm_to_float(const void *src, int fmt, size_t n, float *dst) {
  if (!src || !dst || mi_bytes_per_sample(fmt) == 0) {
    return MI_ERR_ARG;
  }
  for (size_t i = 0; i < n; ++i) {
    dst[i] = miups::pcm_load(src, fmt, static_cast<long long>(i));
  }
  return MI_OK;
}

int mi_float_to_pcm(const float *src, size_t n, int fmt, void *dst) {
  if (!src || !dst || mi_bytes_per_sample(fmt) == 0) {
    return MI_ERR_ARG;
  }
  for (size_t i = 0; i < n; ++i) {
    miups::pcm_store(dst, fmt, static_cast<long long>(i), src[i]);
  }
  return MI_OK;
}

long mi_eq_parse(const char *text, double *preamp_db, double *bands9, size_t max_bands) {
  if (!text) {
    return -1;
  }
  miups::eq::EqProfile profile;
  if (!miups::eq::parseEqString(text, profile)) {
    return -1;
  }
  if (preamp_db) {
    *preamp_db = profile.preampDb;
  }
  for (size_t i = 0; bands9 && i < profile.bands.size() && i < max_bands; ++i) {
    const auto &b = profile.bands[i];
    double *o = bands9 + 9 * i;
    o[0] = b.enabled;
    o[1] = static_cast<int>(b.type);
    o[2] = b.frequency;
    o[3] = b.gain;
    o[4] = b.q;
    o[5] = b.hasBandwidthHz;
    o[6] = b.bandwidthHz;
    o[7] = b.hasBandwidthOct;
    o[8] = b.bandwidthOct;
  }
  return static_cast<long>(profile.bands.size());
}

int mi_eq_parse_filter_type(const char *s) { return s ? static_cast<int>(miups::eq::parseFilterType(s)) : 0; }

const char *mi_eq_filter_type_name(int type) {
  return miups::eq::filterTypeName(static_cast<miups::eq::FilterType>(type));
}

int mi_eq_biquad(int enabled, int type, double freq, double gain, double q, double fs, double *out5) {
  if (!out5) {
    return MI_ERR_ARG;
  }
  miups::eq::EqBand band;
  band.enabled = enabled != 0;
  band.type = static_cast<miups::eq::FilterType>(type);
  band.frequency = freq;
  band.gain = gain;
  band.q = q;
  const auto c = miups::eq::calculateBiquadCoeffs(band, fs);
  out5[0] = c.b0;
  out5[1] = c.b1;
  out5[2] = c.b2;
  out5[3] = c.a1;
  out5[4] = c.a2;
  return MI_OK;
}

int mi_eq_response_host(const char *text, size_t num_bins, size_t full_fft, double fs_out, double *out_reim) {
  if (!text || !out_reim) {
    return MI_ERR_ARG;
  }
  miups::eq::EqProfile profile;
  miups::eq::parseEqString(text, profile);
  const auto r = miups::eq::ComputeEqResponseHost(num_bins, full_fft, fs_out, profile);
  std::memcpy(out_reim, r.data(), r.size() * sizeof(std::complex<double>));
  return MI_OK;
}

int mi_eq_magnitude_host(const char *text, size_t num_bins, size_t full_fft, double fs_out, double *out_mag) {
  if (!text || !out_mag) {
    return MI_ERR_ARG;
  }
  miups::eq::EqProfile profile;
  miups::eq::parseEqString(text, profile);
  const auto m = miups::eq::ComputeEqMagnitudeHost(num_bins, full_fft, fs_out, profile);
  std::memcpy(out_mag, m.data(), m.size() * sizeof(double));
  return MI_OK;
}

int mi_tables_build(const char *json_path, int flags, const char *apo_text, double fs_out, mi_tables **out, char *err,
                    size_t errcap) {
  if (!json_path || !out) {
    Put("null argument", err, errcap);
    return MI_ERR_ARG;
  }
  miups::FilterConfig config;
  std::vector<float> taps;
  std::string error;
  if (!miups::ReadFilter(json_path, &config, &taps, &error)) {
    Put(error, err, errcap);
    return MI_ERR_FILTER;
  }
  std::vector<std::complex<double>> half;
  const bool withEq = apo_text && apo_text[0] != '\0';
  if (withEq) {
    miups::eq::EqProfile profile;
    miups::eq::parseEqString(apo_text, profile);
    half = miups::eq::ComputeEqResponseHost(config.fftSize / 2 + 1, config.fftSize, fs_out, profile);
  }
  auto *t = new mi_tables();
  if (!miups::BuildTables(config, taps, withEq ? &half : nullptr, flags, &t->tables, &error)) {
    delete t;
    Put(error, err, errcap);
    return MI_ERR_FILTER;
  }
  *out = t;
  Put("", err, errcap);
  return MI_OK;
}

int mi_tables_geometry(const mi_tables *t, int *g10) {
  if (!t || !g10) {
    return MI_ERR_ARG;
  }
  const miups::Geometry &g = t->tables.geo;
  const int v[10] = {g.log2k, g.K, g.M, g.P, g.S, g.Oc, g.Bc, g.n_in, g.B, g.hist_frames};
  std::memcpy(g10, v, sizeof(v));
  return MI_OK;
}

// which 0..3: natural-order tables; 4 WmT, 5 GT, 6 G0: the fused kernel's thread-order
// copies (GT / G0 entries are {Gs, Gc} pairs = two complex values each)
size_t mi_tables_size(const mi_tables *t, int which) {
  if (!t) {
    return 0;
  }
  if (which == 4) {
    return t->tables.WmT.size();
  }
  if (which == 5) {
    return 2 * t->tables.GT.size();
  }
  if (which == 6) {
    return 2 * t->tables.G0.size();
  }
  const auto *v = Pick(t, which);
  return v ? v->size() : 0;
}

int mi_tables_copy(const mi_tables *t, int which, float *out_reim, size_t cap_complex) {
  if (!t || !out_reim || cap_complex < mi_tables_size(t, which)) {
    return MI_ERR_ARG;
  }
  const void *src = nullptr;
  if (which == 4) {
    src = t->tables.WmT.data();
  } else if (which == 5) {
    src = t->tables.GT.data();
  } else if (which == 6) {
    src = t->tables.G0.data();
  } else {
    const auto *v = Pick(t, which);
    if (!v) {
      return MI_ERR_ARG;
    }
    src = v->data();
  }
  std::memcpy(out_reim, src, mi_tables_size(t, which) * sizeof(miups::cf));
  return MI_OK;
}

int mi_tables_block_b(const mi_tables *t, int *out, size_t cap) {
  if (!t || !out || cap < t->tables.blockB.size()) {
    return MI_ERR_ARG;
  }
  std::memcpy(out, t->tables.blockB.data(), t->tables.blockB.size() * sizeof(int));
  return MI_OK;
}

int mi_lds_swizzle(int word_index) { return miups::lds_swz(word_index); }

int mi_fused_set_of_block(int block, int log2k) { return miups::FusedSetOfBlock(block, log2k); }

int mi_fused_block_a(int tau, int log2k) { return miups::FusedBlockA(tau, log2k); }

void mi_tables_free(mi_tables *t) { delete t; }

}  // extern "C"
