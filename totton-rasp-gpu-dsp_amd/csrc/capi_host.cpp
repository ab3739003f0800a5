// Host-only part of the C ABI (section (3) of include/mi_upsampler.h).
#include "../../include/mi_upsampler.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "device/pcm.h"
#include "host/eq.h"
#include "host/filter_config.h"
#include "host/filter_selector.h"
#include "host/negotiation.h"
#include "host/opra.h"
#include "host/pcm_ring.h"
#include "host/runtime_config.h"
#include "host/stream_loop.h"
#include "host/spectrum.h"

struct mi_ring {
  miups::PcmRing ring;
};

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

int mi_eq_fold_host(const float *taps, size_t n_taps, size_t fft_size, const char *apo_text, double fs_out, double *out_fir,
                    mi_eq_residual *out) {
  if (!taps || !apo_text || n_taps == 0 || !(fs_out > 0.0)) {
    return MI_ERR_ARG;
  }
  try {
    const std::vector<float> h(taps, taps + n_taps);
    miups::eq::EqProfile profile;
    miups::eq::parseEqString(apo_text, profile);
    miups::eq::EqFold fold = miups::eq::FoldCascadeIntoTaps(h, miups::eq::buildCascade(profile, fs_out));
    if (out_fir) {
      std::memcpy(out_fir, fold.fir.data(), n_taps * sizeof(double));
    }
    if (out) {
      auto db = [](double v) { return v > 0.0 ? 20.0 * std::log10(v) : -400.0; };
      std::memset(out, 0, sizeof(*out));
      if (fft_size >= n_taps && fft_size >= 2 && (fft_size & (fft_size - 1)) == 0) {
        fold.responseDev = miups::eq::ResponseDeviation(
            h, fold.fir, miups::eq::ComputeEqResponseHost(fft_size / 2 + 1, fft_size, fs_out, profile), fft_size);
      }
      out->active = 1;
      out->limit = miups::eq::kFoldDefaultLimit;
      out->over_limit = fold.tailL1 > out->limit;
      out->tail_complete = fold.tailComplete ? 1 : 0;
      out->tail_l1 = fold.tailL1;
      out->tail_l2 = fold.tailL2;
      out->tail_l1_db = db(fold.tailL1);
      out->tail_l2_db = db(fold.tailL2);
      out->response_dev = fold.responseDev;
      out->response_dev_db = db(fold.responseDev);
      out->fir_taps = n_taps;
      out->taper = fold.taper;
    }
  } catch (...) {
    return MI_ERR_ARG;
  }
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
  miups::eq::EqFold fold;
  const bool withEq = apo_text && apo_text[0] != '\0';
  if (withEq) {
    miups::eq::EqProfile profile;
    miups::eq::parseEqString(apo_text, profile);
    fold = miups::eq::FoldCascadeIntoTaps(taps, miups::eq::buildCascade(profile, fs_out));
  }
  auto *t = new mi_tables();
  if (!miups::BuildTables(config, taps, withEq ? &fold.fir : nullptr, flags, &t->tables, &error)) {
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

// layout facts of the product's (classic) pass plan
int mi_fused_set_of_block(int block, int log2k) { return miups::FusedSetOfBlock(block, log2k, false); }

int mi_fused_block_a(int tau, int log2k) { return miups::FusedBlockA(tau, log2k, false); }

int mi_fused_plan_radices(int log2k, int *out, size_t cap) {
  const std::vector<int> r = miups::FusedRadices(log2k, false);
  if (!out || cap < r.size()) {
    return -1;
  }
  for (std::size_t i = 0; i < r.size(); ++i) {
    out[i] = r[i];
  }
  return static_cast<int>(r.size());
}

void mi_tables_free(mi_tables *t) { delete t; }


// ---- negotiation -------------------------------------------------------------------------------------------
int mi_rate_family(int sample_rate) { return static_cast<int>(miups::GetRateFamily(sample_rate)); }
int mi_same_family(int a, int b) { return miups::IsSameFamily(a, b) ? 1 : 0; }
int mi_upsample_ratio(int input_rate, int output_rate) { return miups::CalculateUpsampleRatio(input_rate, output_rate); }

int mi_negotiate(int input_rate, int dac_valid, int dac_min_rate, int dac_max_rate, const int *dac_rates, size_t n_rates,
                 int current_output_rate, mi_negotiated *out) {
  if (!out) {
    return MI_ERR_ARG;
  }
  miups::DacRates dac;
  dac.valid = dac_valid != 0;
  dac.minRate = dac_min_rate;
  dac.maxRate = dac_max_rate;
  if (dac_rates && n_rates) {
    dac.rates.assign(dac_rates, dac_rates + n_rates);
  }
  if (!dac.valid) {
    dac.errorMessage = "Device not found";
  }
  const miups::Negotiated n = miups::Negotiate(input_rate, dac, current_output_rate);
  out->input_rate = n.inputRate;
  out->family = static_cast<int>(n.family);
  out->output_rate = n.outputRate;
  out->ratio = n.ratio;
  out->valid = n.valid ? 1 : 0;
  out->requires_reconfiguration = n.requiresReconfiguration ? 1 : 0;
  Put(n.errorMessage, out->error, sizeof(out->error));
  return MI_OK;
}

// ---- OPRA record -> APO text ---------------------------------------------------------------------------------
int mi_opra_to_apo(const char *eq_json, int modern_target, char *out, size_t cap, size_t *needed, char *err, size_t errcap)
try {
  if (!eq_json || !out) {
    Put("null argument", err, errcap);
    return MI_ERR_ARG;
  }
  std::string text, error;
  if (!miups::OpraToApo(eq_json, modern_target != 0, &text, &error)) {
    Put(error, err, errcap);
    return MI_ERR_FILTER;
  }
  if (needed) {
    *needed = text.size() + 1;
  }
  if (text.size() + 1 > cap) {
    Put("output buffer too small", err, errcap);
    return MI_ERR_ARG;
  }
  std::memcpy(out, text.c_str(), text.size() + 1);
  Put("", err, errcap);
  return MI_OK;
} catch (const std::exception &e) {  // no exception crosses the C boundary (std::bad_alloc on a huge record, ...)
  Put(std::string("mi_opra_to_apo: ") + e.what(), err, errcap);
  return MI_ERR_ARG;
} catch (...) {
  Put("mi_opra_to_apo: unknown exception", err, errcap);
  return MI_ERR_ARG;
}

// ---- config.json -------------------------------------------------------------------------------------------
int mi_parse_runtime_config(const char *json_text, mi_runtime_config *out, char *err, size_t errcap) try {
  if (!json_text || !out) {
    Put("null argument", err, errcap);
    return MI_ERR_ARG;
  }
  miups::RuntimeConfig c;
  std::string error;
  if (!miups::ParseRuntimeConfig(json_text, &c, &error)) {
    Put(error, err, errcap);
    return MI_ERR_FILTER;
  }
  std::memset(out, 0, sizeof(*out));
  out->eq_enabled = c.eqEnabled ? 1 : 0;
  Put(c.eqProfile, out->eq_profile, sizeof(out->eq_profile));
  Put(c.eqProfilePath, out->eq_profile_path, sizeof(out->eq_profile_path));
  out->ratio = c.ratio;
  Put(c.phaseType, out->phase_type, sizeof(out->phase_type));
  Put(c.filterDirectory, out->filter_directory, sizeof(out->filter_directory));
  out->sample_rate = c.sampleRate;
  out->channels = c.channels;
  out->period_frames = c.periodFrames;
  out->buffer_frames = c.bufferFrames;
  Put(c.format, out->format, sizeof(out->format));
  Put(c.inputDevice, out->input_device, sizeof(out->input_device));
  Put(c.outputDevice, out->output_device, sizeof(out->output_device));
  Put("", err, errcap);
  return MI_OK;
} catch (const std::exception &e) {
  Put(std::string("config parse error: ") + e.what(), err, errcap);
  return MI_ERR_FILTER;
} catch (...) {
  Put("config parse error: unknown exception", err, errcap);
  return MI_ERR_FILTER;
}

// ---- ring --------------------------------------------------------------------------------------------------
mi_ring *mi_ring_create(size_t capacity_bytes) {
  if (capacity_bytes == 0) {
    return nullptr;
  }
  mi_ring *r = new (std::nothrow) mi_ring();
  if (r) {
    r->ring.Init(capacity_bytes);
  }
  return r;
}
void mi_ring_destroy(mi_ring *r) { delete r; }
int mi_ring_write(mi_ring *r, const void *data, size_t bytes) { return r && data && r->ring.Write(data, bytes) ? 1 : 0; }
int mi_ring_read(mi_ring *r, void *dst, size_t bytes) { return r && dst && r->ring.Read(dst, bytes) ? 1 : 0; }
size_t mi_ring_available_to_read(const mi_ring *r) { return r ? r->ring.AvailableToRead() : 0; }
size_t mi_ring_available_to_write(const mi_ring *r) { return r ? r->ring.AvailableToWrite() : 0; }
void mi_ring_clear(mi_ring *r) {
  if (r) {
    r->ring.DiscardAll();
  }
}

// ---- streaming loop ----------------------------------------------------------------------------------------
int mi_stream_loop_run(const mi_loop_params *p, mi_read_fn read, mi_write_fn write, mi_process_fn process,
                       mi_between_fn between, mi_log_fn log, void *user, const volatile int *running,
                       mi_loop_stats *stats) try {
  if (!p || !read || !write) {
    return MI_ERR_ARG;
  }
  miups::LoopParams lp;
  lp.channels = p->channels;
  lp.format = p->format;
  lp.periodFrames = p->period_frames;
  lp.blockInFrames = p->block_in_frames;
  lp.blockOutFrames = p->block_out_frames;
  lp.maxBlocksPerCall = p->max_blocks_per_call;
  lp.drainAtEnd = p->drain_at_end != 0;
  if (p->pinned_rings) {
    lp.hostAlloc = mi_host_alloc;
    lp.hostFree = mi_host_free;
  }
  miups::LoopStats st;
  miups::ProcessFn proc;
  if (process) {
    proc = [&](const void *in, void *out, std::size_t blocks) { return process(user, in, out, blocks) != 0; };
  }
  miups::BetweenBlocksFn btw;
  if (between) {
    btw = [&]() { between(user); };
  }
  const bool ok = miups::RunStreamLoop(
      lp, [&](void *dst, std::size_t frames) { return read(user, dst, frames); },
      [&](const void *src, std::size_t frames) { return write(user, src, frames) != 0; }, proc, btw,
      [&]() { return !running || *running != 0; }, &st,
      [&](const std::string &m) {
        if (log) {
          log(user, m.c_str());
        }
      });
  if (stats) {
    stats->periods_read = st.periodsRead;
    stats->blocks_processed = st.blocksProcessed;
    stats->frames_written = st.framesWritten;
    stats->silence_frames_written = st.silenceFramesWritten;
    stats->input_overflows = st.inputOverflows;
    stats->output_overflows = st.outputOverflows;
    stats->process_calls = st.processCalls;
    stats->in_place_calls = st.inPlaceCalls;
  }
  return ok ? MI_OK : MI_ERR_DEVICE;
} catch (...) {  // an exception out of a callback or an allocation: stop the loop, never unwind into C
  return MI_ERR_DEVICE;
}

}  // extern "C"
