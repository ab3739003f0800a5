// extern "C" boundary (include/mi_upsampler.h). No exceptions escape; every
// failure leaves a message in the thread-local last-error slot.
#include "../../include/mi_upsampler.h"

#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <unistd.h>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "engine.h"
#include "filter_bank.h"
#include "multi_engine.h"

using miups::DeviceFilter;
using miups::Engine;
using miups::FilterConfig;

struct mi_filter {
  std::shared_ptr<DeviceFilter> filter;
};

struct mi_engine {
  std::unique_ptr<Engine> engine;
};

struct mi_multi {
  std::unique_ptr<miups::MultiEngine> multi;
};

struct mi_bank {
  std::unique_ptr<miups::FilterBank> bank;
};

struct mi_ups {
  int device = 0;
  bool initialized = false;
  FilterConfig config{};
  std::shared_ptr<DeviceFilter> filter;
  std::unique_ptr<Engine> engine;
};

namespace {

void CopyMessage(const std::string &m, char *dst, size_t cap) {
  if (dst && cap) {
    std::strncpy(dst, m.c_str(), cap - 1);
    dst[cap - 1] = '\0';
  }
}

int Fail(int code, const std::string &m, char *err = nullptr, size_t cap = 0) {
  miups::SetLastError(m);
  CopyMessage(m, err, cap);
  return code;
}

void FillConfig(const FilterConfig &c, mi_ups_config *out) {
  out->taps = c.taps;
  out->fft_size = c.fftSize;
  out->block_size = c.blockSize;
  out->upsample_factor = c.upsampleFactor;
  CopyMessage(c.coefficientsPath, out->coefficients_path, sizeof(out->coefficients_path));
}

void FillResidual(const miups::EqReport &r, mi_eq_residual *out) {
  auto db = [](double v) { return v > 0.0 ? 20.0 * std::log10(v) : -400.0; };
  std::memset(out, 0, sizeof(*out));
  out->active = r.active ? 1 : 0;
  out->over_limit = r.overLimit ? 1 : 0;
  out->tail_complete = r.tailComplete ? 1 : 0;
  out->tail_l1 = r.tailL1;
  out->tail_l2 = r.tailL2;
  out->tail_l1_db = db(r.tailL1);
  out->tail_l2_db = db(r.tailL2);
  out->response_dev = r.responseDev;
  out->response_dev_db = db(r.responseDev);
  out->limit = r.limit;
  out->fir_taps = r.firTaps;
  out->taper = r.taper;
}

// an EQ change that went through but dropped more than the limit: success, with the warning where errors go
int OkWithEqWarning(const miups::EqReport &r) {
  miups::SetLastError(miups::EqReportWarning(r));
  return MI_OK;
}

int EqFailCode(const std::string &error) { return error.rfind("EQ cut to", 0) == 0 ? MI_ERR_FILTER : MI_ERR_DEVICE; }

template <typename F>
auto Guard(F &&f, decltype(f()) onThrow) -> decltype(f()) {
  try {
    return f();
  } catch (const std::exception &e) {
    miups::SetLastError(std::string("internal error: ") + e.what());
  } catch (...) {
    miups::SetLastError("internal error");
  }
  return onThrow;
}

}  // namespace

extern "C" {

int mi_ups_abi_version(void) { return MI_UPS_ABI_VERSION; }

int mi_ups_device_count(void) { return miups::DeviceCount(); }

const char *mi_ups_last_error(void) { return miups::LastError().c_str(); }

// ---------------------------------------------------------------- level 1 --
mi_ups *mi_ups_create(int device) {
  return Guard(
      [&]() -> mi_ups * {
        mi_ups *h = new (std::nothrow) mi_ups();
        if (h) {
          h->device = device;
        }
        return h;
      },
      nullptr);
}

void mi_ups_destroy(mi_ups *h) { delete h; }

mi_ups *mi_ups_clone(const mi_ups *other) {
  return Guard(
      [&]() -> mi_ups * {
        if (!other) {
          Fail(MI_ERR_ARG, "null handle");
          return nullptr;
        }
        std::unique_ptr<mi_ups> h(new mi_ups());
        h->device = other->device;
        h->initialized = other->initialized;
        h->config = other->config;
        h->filter = other->filter;  // tables are immutable while shared
        if (other->engine) {
          std::string error;
          h->engine = other->engine->Clone(&error);
          if (!h->engine) {
            Fail(MI_ERR_DEVICE, error);
            return nullptr;
          }
        }
        return h.release();
      },
      nullptr);
}

int mi_ups_load_filter(mi_ups *h, const char *json_path, int flags, char *err, size_t errcap) {
  return Guard(
      [&]() -> int {
        if (!h || !json_path) {
          return Fail(MI_ERR_ARG, "null argument", err, errcap);
        }
        FilterConfig config;
        std::vector<float> taps;
        std::string error;
        if (!miups::ReadFilter(json_path, &config, &taps, &error)) {
          return Fail(MI_ERR_FILTER, error, err, errcap);
        }
        auto filter = DeviceFilter::Create(h->device, config, std::move(taps), flags, &error);
        if (!filter) {
          return Fail(MI_ERR_DEVICE, error, err, errcap);
        }
        auto engine = Engine::Create(filter, 1, 1, MI_PCM_F32, MI_PCM_F32, &error);
        if (!engine) {
          return Fail(MI_ERR_DEVICE, error, err, errcap);
        }
        h->config = config;
        h->filter = std::move(filter);
        h->engine = std::move(engine);
        h->initialized = true;
        CopyMessage("", err, errcap);
        return MI_OK;
      },
      MI_ERR_DEVICE);
}

int mi_ups_get_config(const mi_ups *h, mi_ups_config *out) {
  if (!h || !out) {
    return Fail(MI_ERR_ARG, "null argument");
  }
  FillConfig(h->config, out);
  return MI_OK;
}

long mi_ups_process_block(mi_ups *h, const float *input, size_t count, float *out, size_t outcap) {
  return Guard(
      [&]() -> long {
        // the reference's guards, in its order (vulkan_streaming_upsampler.cpp:502-526)
        if (!h || !h->initialized || !input) {
          Fail(MI_ERR_ARG, "not initialised or null input");
          return 0;
        }
        if (count == 0) {
          Fail(MI_ERR_SIZE, "count == 0");
          return 0;
        }
        const size_t factor = h->config.upsampleFactor > 1 ? h->config.upsampleFactor : 1;
        if (h->config.blockSize % factor != 0) {
          Fail(MI_ERR_SIZE, "block_size not divisible by upsample_factor");
          return 0;
        }
        const size_t expect = h->config.blockSize / factor;
        if (expect == 0 || count != expect) {
          Fail(MI_ERR_SIZE, "count must equal block_size / upsample_factor");
          return 0;
        }
        if (!out || outcap < h->config.blockSize) {
          Fail(MI_ERR_SIZE, "output buffer too small");
          return 0;
        }
        std::string error;
        if (!h->engine->ProcessHost(input, 0, out, 0, 1, &error)) {
          Fail(MI_ERR_DEVICE, error);
          return 0;
        }
        return static_cast<long>(h->config.blockSize);
      },
      0L);
}

int mi_ups_reset(mi_ups *h) {
  return Guard(
      [&]() -> int {
        if (!h) {
          return Fail(MI_ERR_ARG, "null handle");
        }
        if (!h->engine) {
          return MI_OK;  // Reset() on an unloaded instance is a no-op in the reference too
        }
        std::string error;
        return h->engine->Reset(&error) ? MI_OK : Fail(MI_ERR_DEVICE, error);
      },
      MI_ERR_DEVICE);
}

int mi_ups_set_eq(mi_ups *h, const char *apo_text, double fs_out) {
  return Guard(
      [&]() -> int {
        if (!h || !h->initialized) {
          return Fail(MI_ERR_ARG, "not initialised");
        }
        std::string error;
        // the handle and its engine hold one reference each; more means clones share these tables, and an EQ change
        // must stay private to this handle: build a private filter WITH the EQ (one table build, not two)
        if (h->filter.use_count() > 2) {
          auto own = h->filter->Fork(apo_text ? apo_text : "", fs_out, &error);
          if (!own) {
            return Fail(EqFailCode(error), error);
          }
          if (!h->engine->Rebind(own, false, &error)) {
            return Fail(MI_ERR_DEVICE, error);
          }
          h->filter = own;
          return OkWithEqWarning(h->filter->eqReport());
        }
        if (!h->filter->SetEq(apo_text ? apo_text : "", fs_out, &error)) {
          return Fail(EqFailCode(error), error);
        }
        return OkWithEqWarning(h->filter->eqReport());
      },
      MI_ERR_DEVICE);
}

int mi_ups_eq_residual(const mi_ups *h, mi_eq_residual *out) {
  if (!h || !h->initialized || !out) {
    return Fail(MI_ERR_ARG, "null argument");
  }
  FillResidual(h->filter->eqReport(), out);
  return MI_OK;
}

int mi_ups_set_eq_limit(mi_ups *h, double max_tail_l1, int strict) {
  if (!h || !h->initialized) {
    return Fail(MI_ERR_ARG, "not initialised");
  }
  h->filter->SetEqLimit(max_tail_l1, strict != 0);
  return MI_OK;
}

// ---------------------------------------------------------------- level 2 --
int mi_filter_load(int device, const char *json_path, int flags, mi_filter **out, char *err, size_t errcap) {
  return Guard(
      [&]() -> int {
        if (!json_path || !out) {
          return Fail(MI_ERR_ARG, "null argument", err, errcap);
        }
        FilterConfig config;
        std::vector<float> taps;
        std::string error;
        if (!miups::ReadFilter(json_path, &config, &taps, &error)) {
          return Fail(MI_ERR_FILTER, error, err, errcap);
        }
        auto filter = DeviceFilter::Create(device, config, std::move(taps), flags, &error);
        if (!filter) {
          return Fail(MI_ERR_DEVICE, error, err, errcap);
        }
        *out = new mi_filter{std::move(filter)};
        CopyMessage("", err, errcap);
        return MI_OK;
      },
      MI_ERR_DEVICE);
}

int mi_filter_from_taps(int device, const float *taps, size_t n_taps, size_t fft_size, size_t block_size,
                        size_t upsample_factor, int flags, mi_filter **out, char *err, size_t errcap) {
  return Guard(
      [&]() -> int {
        if (!taps || !out) {
          return Fail(MI_ERR_ARG, "null argument", err, errcap);
        }
        // same rules, same messages as the sidecar loader
        if (n_taps == 0 || fft_size == 0 || block_size == 0) {
          return Fail(MI_ERR_FILTER, "taps/fft_size/block_size must be set and non-zero", err, errcap);
        }
        if ((fft_size & (fft_size - 1)) != 0) {
          return Fail(MI_ERR_FILTER, "fft_size must be power of two", err, errcap);
        }
        if (block_size >= fft_size) {
          return Fail(MI_ERR_FILTER, "block_size must be smaller than fft_size", err, errcap);
        }
        if (fft_size - block_size != n_taps - 1) {
          return Fail(MI_ERR_FILTER, "block_size must satisfy fft_size - block_size == taps - 1", err, errcap);
        }
        FilterConfig config;
        config.taps = n_taps;
        config.fftSize = fft_size;
        config.blockSize = block_size;
        config.upsampleFactor = upsample_factor == 0 ? 1 : upsample_factor;
        if (config.upsampleFactor > 1 && block_size % config.upsampleFactor != 0) {
          return Fail(MI_ERR_FILTER, "block_size must be divisible by upsample_factor", err, errcap);
        }
        std::string error;
        auto filter =
            DeviceFilter::Create(device, config, std::vector<float>(taps, taps + n_taps), flags, &error);
        if (!filter) {
          return Fail(MI_ERR_DEVICE, error, err, errcap);
        }
        *out = new mi_filter{std::move(filter)};
        CopyMessage("", err, errcap);
        return MI_OK;
      },
      MI_ERR_DEVICE);
}

int mi_filter_get_config(const mi_filter *f, mi_ups_config *out) {
  if (!f || !out) {
    return Fail(MI_ERR_ARG, "null argument");
  }
  FillConfig(f->filter->config(), out);
  return MI_OK;
}

int mi_filter_set_eq(mi_filter *f, const char *apo_text, double fs_out) {
  return Guard(
      [&]() -> int {
        if (!f) {
          return Fail(MI_ERR_ARG, "null filter");
        }
        std::string error;
        if (!f->filter->SetEq(apo_text ? apo_text : "", fs_out, &error)) {
          return Fail(EqFailCode(error), error);
        }
        return OkWithEqWarning(f->filter->eqReport());
      },
      MI_ERR_DEVICE);
}

int mi_filter_eq_residual(const mi_filter *f, mi_eq_residual *out) {
  if (!f || !out) {
    return Fail(MI_ERR_ARG, "null argument");
  }
  FillResidual(f->filter->eqReport(), out);
  return MI_OK;
}

int mi_filter_set_eq_limit(mi_filter *f, double max_tail_l1, int strict) {
  if (!f) {
    return Fail(MI_ERR_ARG, "null filter");
  }
  f->filter->SetEqLimit(max_tail_l1, strict != 0);
  return MI_OK;
}

int mi_eq_response_device(int device, const char *apo_text, size_t num_bins, size_t full_fft, double fs_out,
                          double *out_reim) {
  return Guard(
      [&]() -> int {
        if (!apo_text || !out_reim) {
          return Fail(MI_ERR_ARG, "null argument");
        }
        std::vector<std::complex<double>> r;
        std::string error;
        if (!miups::EqResponseDevice(device, apo_text, num_bins, full_fft, fs_out, &r, &error)) {
          return Fail(MI_ERR_DEVICE, error);
        }
        std::memcpy(out_reim, r.data(), r.size() * sizeof(std::complex<double>));
        return MI_OK;
      },
      MI_ERR_DEVICE);
}

void mi_filter_release(mi_filter *f) { delete f; }

int mi_engine_create(mi_filter *f, int streams, int channels, int in_fmt, int out_fmt, mi_engine **out) {
  return Guard(
      [&]() -> int {
        if (!f || !out) {
          return Fail(MI_ERR_ARG, "null argument");
        }
        std::string error;
        auto e = Engine::Create(f->filter, streams, channels, in_fmt, out_fmt, &error);
        if (!e) {
          return Fail(MI_ERR_DEVICE, error);
        }
        *out = new mi_engine{std::move(e)};
        return MI_OK;
      },
      MI_ERR_DEVICE);
}

void mi_engine_destroy(mi_engine *e) { delete e; }

int mi_engine_reset(mi_engine *e) {
  return Guard(
      [&]() -> int {
        if (!e) {
          return Fail(MI_ERR_ARG, "null engine");
        }
        std::string error;
        return e->engine->Reset(&error) ? MI_OK : Fail(MI_ERR_DEVICE, error);
      },
      MI_ERR_DEVICE);
}

size_t mi_engine_in_frames_per_block(const mi_engine *e) {
  return e ? static_cast<size_t>(e->engine->filter()->geometry().n_in) : 0;
}

size_t mi_engine_out_frames_per_block(const mi_engine *e) {
  return e ? static_cast<size_t>(e->engine->filter()->geometry().B) : 0;
}

const char *mi_engine_path(const mi_engine *e) { return (e && e->engine->fused()) ? "fused" : "staged"; }

int mi_engine_process_device(mi_engine *e, const void *d_in, size_t in_stream_stride_bytes, void *d_out,
                             size_t out_stream_stride_bytes, size_t blocks, void *hip_stream) {
  return Guard(
      [&]() -> int {
        if (!e) {
          return Fail(MI_ERR_ARG, "null engine");
        }
        std::string error;
        return e->engine->ProcessDevice(d_in, in_stream_stride_bytes, d_out, out_stream_stride_bytes, blocks,
                                        hip_stream, &error)
                   ? MI_OK
                   : Fail(MI_ERR_DEVICE, error);
      },
      MI_ERR_DEVICE);
}

int mi_engine_process_host(mi_engine *e, const void *h_in, size_t in_stream_stride_bytes, void *h_out,
                           size_t out_stream_stride_bytes, size_t blocks) {
  return Guard(
      [&]() -> int {
        if (!e) {
          return Fail(MI_ERR_ARG, "null engine");
        }
        std::string error;
        return e->engine->ProcessHost(h_in, in_stream_stride_bytes, h_out, out_stream_stride_bytes, blocks, &error)
                   ? MI_OK
                   : Fail(MI_ERR_DEVICE, error);
      },
      MI_ERR_DEVICE);
}

int mi_host_register(void *p, size_t bytes) {
  return Guard(
      [&]() -> int {
        std::string error;
        if (!miups::HostRegister(p, bytes, &error)) {
          return Fail(MI_ERR_DEVICE, error);
        }
        return MI_OK;
      },
      MI_ERR_DEVICE);
}

void mi_host_unregister(void *p) { miups::HostUnregister(p); }

int mi_device_copy_rate(int device, size_t bytes, int iters, double *gbps) {
  return Guard(
      [&]() -> int {
        std::string error;
        if (!miups::DeviceCopyRate(device, bytes, iters, gbps, &error)) {
          return Fail(MI_ERR_DEVICE, error);
        }
        return MI_OK;
      },
      MI_ERR_DEVICE);
}

void *mi_host_alloc(size_t bytes) {
  return Guard(
      [&]() -> void * {
        std::string error;
        void *p = miups::HostAlloc(bytes, &error);
        if (!p) {
          Fail(MI_ERR_DEVICE, error);
        }
        return p;
      },
      nullptr);
}

void mi_host_free(void *p) { miups::HostFree(p); }

int mi_engine_rebind(mi_engine *e, mi_filter *f, int reset_history) {
  return Guard(
      [&]() -> int {
        if (!e || !f) {
          return Fail(MI_ERR_ARG, "null argument");
        }
        std::string error;
        return e->engine->Rebind(f->filter, reset_history != 0, &error) ? MI_OK : Fail(MI_ERR_DEVICE, error);
      },
      MI_ERR_DEVICE);
}

unsigned long long mi_filter_generation(const mi_filter *f) { return f ? f->filter->generation() : 0; }

unsigned long long mi_engine_last_generation(const mi_engine *e) { return e ? e->engine->lastGeneration() : 0; }

// Diagnostic: print the native stack of the thread that raises SIGABRT (the HIP runtime aborts without a message in some
// internal failures; Python's faulthandler only knows Python frames), then let the previous handler (or the default) run.
namespace {
struct sigaction g_prevAbort;
int g_abortFd = 2;
void AbortBacktrace(int sig) {
  void *frames[64];
  const int n = backtrace(frames, 64);
  const char msg[] = "\n== native backtrace of the aborting thread ==\n";
  (void)!write(g_abortFd, msg, sizeof(msg) - 1);
  backtrace_symbols_fd(frames, n, g_abortFd);
  sigaction(SIGABRT, &g_prevAbort, nullptr);
  raise(sig);
}
}  // namespace
unsigned long long mi_debug_unsafe_host_copies(void) { return miups::UnsafeHostCopies(); }

void mi_debug_fail_host_call_at(mi_engine *e, int sub_batch) {
  if (e && e->engine) {
    e->engine->FailHostCallAtForTest(sub_batch);
  }
}

void mi_debug_multi_fail_host_call_at(mi_multi *m, int slot, int sub_batch) {
  if (m && m->multi) {
    m->multi->FailHostCallAtForTest(slot, sub_batch);
  }
}

void mi_debug_install_abort_backtrace(void) {
  // MIUPS_ABORT_BACKTRACE=<path>: where to write (a test runner may have redirected descriptor 2); anything else: stderr
  if (const char *path = std::getenv("MIUPS_ABORT_BACKTRACE")) {
    if (path[0] == '/' || path[0] == '.' ) {
      const int fd = open(path, O_WRONLY | O_CREAT | O_APPEND, 0644);
      if (fd >= 0) {
        g_abortFd = fd;
      }
    }
  }
  struct sigaction sa;
  std::memset(&sa, 0, sizeof(sa));
  sa.sa_handler = AbortBacktrace;
  sigemptyset(&sa.sa_mask);
  sigaction(SIGABRT, &sa, &g_prevAbort);
}

void mi_debug_fail_next_table_upload(mi_filter *f) {
  if (f) {
    f->filter->FailNextUploadForTest();
  }
}

int mi_engine_enable_kernel_timing(mi_engine *e, int slots) {
  return Guard(
      [&]() -> int {
        if (!e || slots < 0 || slots > 65536) {
          return Fail(MI_ERR_ARG, "bad timing slot count");
        }
        std::string error;
        return e->engine->EnableTiming(slots, &error) ? MI_OK : Fail(MI_ERR_DEVICE, error);
      },
      MI_ERR_DEVICE);
}

int mi_engine_kernel_ms_stats(mi_engine *e, double *avg, double *min_ms, double *max_ms, int *count) {
  if (!e || !avg || !min_ms || !max_ms || !count) {
    return Fail(MI_ERR_ARG, "null argument");
  }
  return e->engine->KernelMsStats(avg, min_ms, max_ms, count) ? MI_OK : Fail(MI_ERR_DEVICE, "no timed calls recorded");
}

int mi_engine_last_two_level(const mi_engine *e) { return (e && e->engine->lastTwoLevel()) ? 1 : 0; }

int mi_engine_last_phase_parts(const mi_engine *e) { return e ? e->engine->lastPhaseParts() : 0; }

int mi_engine_set_kernel_timing_stride(mi_engine *e, int every) {
  if (!e || every < 1) {
    return Fail(MI_ERR_ARG, "timing stride must be >= 1");
  }
  e->engine->SetTimingStride(every);
  return MI_OK;
}

int mi_engine_last_coop_frames(const mi_engine *e) { return (e && e->engine && e->engine->lastCoopFrames()) ? 1 : 0; }

double mi_engine_last_kernel_ms(mi_engine *e) { return e ? e->engine->LastKernelMs() : -1.0; }

int mi_engine_enable_class_timing(mi_engine *e, int on) {
  if (!e) {
    return Fail(MI_ERR_ARG, "null engine");
  }
  e->engine->EnableClassTiming(on != 0);
  return MI_OK;
}

int mi_engine_last_class_ms(mi_engine *e, double *out4) {
  if (!e || !out4) {
    return Fail(MI_ERR_ARG, "null argument");
  }
  return e->engine->LastClassMs(out4) ? MI_OK : Fail(MI_ERR_DEVICE, "no class-timed call recorded");
}


// --------------------------------------------------------------- level 2b --
int mi_multi_create(const char *json_path, int flags, const int *devices, size_t n_devices, int streams, int channels,
                    int in_fmt, int out_fmt, mi_multi **out, char *err, size_t errcap) {
  return Guard(
      [&]() -> int {
        if (!json_path || !devices || n_devices == 0 || !out) {
          return Fail(MI_ERR_ARG, "null argument", err, errcap);
        }
        FilterConfig config;
        std::vector<float> taps;
        std::string error;
        if (!miups::ReadFilter(json_path, &config, &taps, &error)) {
          return Fail(MI_ERR_FILTER, error, err, errcap);
        }
        if ((flags & MI_MULTI_SPLIT_CHANNELS) && (flags & MI_MULTI_SPLIT_TIME)) {
          return Fail(MI_ERR_ARG, "MI_MULTI_SPLIT_CHANNELS and MI_MULTI_SPLIT_TIME exclude each other", err, errcap);
        }
        const int split = (flags & MI_MULTI_SPLIT_CHANNELS) ? miups::kSplitChannels
                                                            : ((flags & MI_MULTI_SPLIT_TIME) ? miups::kSplitTime : miups::kSplitStreams);
        auto m = miups::MultiEngine::Create(std::vector<int>(devices, devices + n_devices), config, taps,
                                            flags & ~(MI_MULTI_SPLIT_CHANNELS | MI_MULTI_SPLIT_TIME), streams, channels, in_fmt,
                                            out_fmt, &error, split);
        if (!m) {
          return Fail(MI_ERR_DEVICE, error, err, errcap);
        }
        *out = new mi_multi{std::move(m)};
        CopyMessage("", err, errcap);
        return MI_OK;
      },
      MI_ERR_DEVICE);
}

void mi_multi_destroy(mi_multi *m) { delete m; }

int mi_multi_set_eq(mi_multi *m, const char *apo_text, double fs_out) {
  return Guard(
      [&]() -> int {
        if (!m) {
          return Fail(MI_ERR_ARG, "null handle");
        }
        std::string error;
        if (!m->multi->SetEq(apo_text ? apo_text : "", fs_out, &error)) {
          return Fail(error.find("EQ cut to") != std::string::npos ? MI_ERR_FILTER : MI_ERR_DEVICE, error);
        }
        return OkWithEqWarning(m->multi->eqReport());
      },
      MI_ERR_DEVICE);
}

int mi_multi_eq_residual(const mi_multi *m, mi_eq_residual *out) {
  if (!m || !out) {
    return Fail(MI_ERR_ARG, "null argument");
  }
  FillResidual(m->multi->eqReport(), out);
  return MI_OK;
}

int mi_multi_set_eq_limit(mi_multi *m, double max_tail_l1, int strict) {
  if (!m) {
    return Fail(MI_ERR_ARG, "null handle");
  }
  m->multi->SetEqLimit(max_tail_l1, strict != 0);
  return MI_OK;
}

int mi_multi_reset(mi_multi *m) {
  return Guard(
      [&]() -> int {
        if (!m) {
          return Fail(MI_ERR_ARG, "null handle");
        }
        std::string error;
        return m->multi->Reset(&error) ? MI_OK : Fail(MI_ERR_DEVICE, error);
      },
      MI_ERR_DEVICE);
}

int mi_multi_process_host(mi_multi *m, const void *h_in, size_t in_stream_stride_bytes, void *h_out,
                          size_t out_stream_stride_bytes, size_t blocks) {
  return Guard(
      [&]() -> int {
        if (!m) {
          return Fail(MI_ERR_ARG, "null handle");
        }
        std::string error;
        return m->multi->ProcessHost(h_in, in_stream_stride_bytes, h_out, out_stream_stride_bytes, blocks, &error)
                   ? MI_OK
                   : Fail(MI_ERR_DEVICE, error);
      },
      MI_ERR_DEVICE);
}

size_t mi_multi_in_frames_per_block(const mi_multi *m) { return m ? static_cast<size_t>(m->multi->geometry().n_in) : 0; }
size_t mi_multi_out_frames_per_block(const mi_multi *m) { return m ? static_cast<size_t>(m->multi->geometry().B) : 0; }
int mi_multi_device_of_stream(const mi_multi *m, int stream) {
  return (m && stream >= 0 && stream < m->multi->streams()) ? m->multi->deviceOfStream(stream) : -1;
}

int mi_multi_device_of_channel(const mi_multi *m, int channel) {
  return (m && channel >= 0 && channel < m->multi->channels()) ? m->multi->deviceOfChannel(channel) : -1;
}

int mi_multi_partition_channels(int channels, int slots, int *first_channel_of_slot) {
  if (channels < 0 || slots <= 0 || !first_channel_of_slot) {
    return MI_ERR_ARG;
  }
  const std::vector<int> p = miups::PartitionChannels(channels, slots);
  std::copy(p.begin(), p.end(), first_channel_of_slot);
  return MI_OK;
}

int mi_multi_worker_cpus(const mi_multi *m, int slot, char *out, size_t cap) {
  if (!m || slot < 0 || slot >= m->multi->slots() || !out || cap == 0) {
    return MI_ERR_ARG;
  }
  CopyMessage(m->multi->workerAffinity(slot), out, cap);
  return MI_OK;
}

void mi_debug_multi_fail_next_eq_on_slot(mi_multi *m, int slot) {
  if (m) {
    m->multi->FailNextEqOnSlotForTest(slot);
  }
}

int mi_multi_partition(int streams, int slots, int *slot_of_stream) {
  if (streams < 0 || slots <= 0 || !slot_of_stream) {
    return MI_ERR_ARG;
  }
  const std::vector<int> p = miups::PartitionStreams(streams, slots);
  std::copy(p.begin(), p.end(), slot_of_stream);
  return MI_OK;
}

// --------------------------------------------------------------- level 2c --
int mi_bank_load(int device, const char *filter_dir, mi_bank **out, char *warnings, size_t warncap, char *err,
                 size_t errcap) {
  return Guard(
      [&]() -> int {
        if (!filter_dir || !out) {
          return Fail(MI_ERR_ARG, "null argument", err, errcap);
        }
        std::string warn, error;
        auto b = miups::FilterBank::Load(device, filter_dir, &warn, &error);
        CopyMessage(warn, warnings, warncap);
        if (!b) {
          return Fail(MI_ERR_FILTER, error, err, errcap);
        }
        *out = new mi_bank{std::move(b)};
        CopyMessage("", err, errcap);
        return MI_OK;
      },
      MI_ERR_DEVICE);
}

void mi_bank_release(mi_bank *b) { delete b; }

size_t mi_bank_size(const mi_bank *b) { return b ? b->bank->entries().size() : 0; }

int mi_bank_entry(const mi_bank *b, size_t i, unsigned *family_base_rate, unsigned *ratio, char *phase, size_t phasecap,
                  char *path, size_t pathcap, mi_ups_config *config) {
  if (!b || i >= b->bank->entries().size()) {
    return Fail(MI_ERR_ARG, "bank entry out of range");
  }
  const auto &e = b->bank->entries()[i];
  if (family_base_rate) {
    *family_base_rate = e.familyBaseRate;
  }
  if (ratio) {
    *ratio = e.ratio;
  }
  CopyMessage(e.phase, phase, phasecap);
  CopyMessage(e.path, path, pathcap);
  if (config) {
    FillConfig(e.filter->config(), config);
  }
  return MI_OK;
}

mi_filter *mi_bank_select(const mi_bank *b, unsigned input_rate, unsigned ratio, const char *phase, char *err,
                          size_t errcap) {
  return Guard(
      [&]() -> mi_filter * {
        if (!b) {
          Fail(MI_ERR_ARG, "null bank", err, errcap);
          return nullptr;
        }
        std::string error;
        const auto *e = b->bank->Find(input_rate, ratio, phase ? phase : "min", &error);
        if (!e) {
          Fail(MI_ERR_FILTER, error, err, errcap);
          return nullptr;
        }
        CopyMessage("", err, errcap);
        return new mi_filter{e->filter};
      },
      nullptr);
}

}  // extern "C"
