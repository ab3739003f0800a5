// Host-side engine: owns the device-resident filter tables and the per-stream
// history, selects the kernel family for the geometry and enqueues the launches.
// HIP only -- there is no CPU path behind these classes.
#pragma once

#include <complex>
#include <memory>
#include <string>
#include <vector>

#include "device/common.h"
#include "host/filter_config.h"
#include "host/spectrum.h"

namespace miups {

// thread-local "last error" shared by the C ABI
void SetLastError(const std::string &message);
const std::string &LastError();

int DeviceCount();

// Device-resident, immutable-between-reloads filter state shared by engines.
class DeviceFilter {
 public:
  ~DeviceFilter();
  static std::shared_ptr<DeviceFilter> Create(int device, const FilterConfig &config, std::vector<float> taps,
                                              int flags, std::string *error);
  // Rebuild the tables with (or without, when text is empty) the EQ folded in.
  bool SetEq(const std::string &apoText, double fsOut, std::string *error);
  // A private copy with its own tables (used when one handle changes its EQ).
  std::shared_ptr<DeviceFilter> Fork(std::string *error) const;

  int device() const { return device_; }
  const FilterConfig &config() const { return config_; }
  const Geometry &geometry() const { return geo_; }
  const cf *Gs() const { return dGs_; }
  const cf *Gc() const { return dGc_; }
  const cf *Wm() const { return dWm_; }
  const cf *tw() const { return dtw_; }
  bool hasFused() const { return hasFused_; }
  bool fusedSplit() const { return fusedSplit_; }  // tables are laid out for fused_split_kernel
  FusedTables fused() const { return FusedTables{dtw_, dWmT_, dBlockB_, dGT_, dG0_, wb_, dSelfW_}; }

 private:
  DeviceFilter() = default;
  bool Rebuild(const std::vector<std::complex<double>> *eqHalf, std::string *error);
  void Free();

  int device_ = 0;
  FilterConfig config_{};
  std::vector<float> taps_;
  int flags_ = 0;
  Geometry geo_{};
  cf *dGs_ = nullptr, *dGc_ = nullptr, *dWm_ = nullptr, *dtw_ = nullptr;
  // fused-kernel layout of the same spectra (FusedTables)
  bool hasFused_ = false, fusedSplit_ = false;
  cf *dWmT_ = nullptr, *dSelfW_ = nullptr;
  int *dBlockB_ = nullptr;
  f4 *dGT_ = nullptr, *dG0_ = nullptr;
  cf wb_{1.0f, 0.0f};
};

// Evaluate an APO profile's cascade on the device: bins 0..numBins-1 (fp64).
bool EqResponseDevice(int device, const std::string &apoText, std::size_t numBins, std::size_t fullFft, double fsOut,
                      std::vector<std::complex<double>> *out, std::string *error);

class Engine {
 public:
  ~Engine();
  static std::unique_ptr<Engine> Create(std::shared_ptr<DeviceFilter> filter, int streams, int channels, int inFmt,
                                        int outFmt, std::string *error);
  std::unique_ptr<Engine> Clone(std::string *error) const;  // deep copy of the history
  void Rebind(std::shared_ptr<DeviceFilter> filter) { filter_ = std::move(filter); }

  bool Reset(std::string *error);
  bool ProcessDevice(const void *dIn, std::size_t inStride, void *dOut, std::size_t outStride, std::size_t blocks,
                     void *hipStream, std::string *error);
  bool ProcessHost(const void *hIn, std::size_t inStride, void *hOut, std::size_t outStride, std::size_t blocks,
                   std::string *error);

  const std::shared_ptr<DeviceFilter> &filter() const { return filter_; }
  bool fused() const { return fused_; }
  int streams() const { return streams_; }
  int channels() const { return channels_; }
  int inFmt() const { return inFmt_; }
  int outFmt() const { return outFmt_; }
  // kernel timing: up to `slots` most recent calls keep a hipEvent pair around
  // their main kernel(s), recorded on the stream the kernels are launched on
  bool EnableTiming(int slots, std::string *error);
  double LastKernelMs();
  // average/min/max over the recorded calls since EnableTiming (waits for them)
  bool KernelMsStats(double *avg, double *mn, double *mx, int *count);

 private:
  Engine() = default;
  bool EnsureWork(std::size_t items, std::string *error);
  void PickChannelGroup(std::size_t blocks);

  std::shared_ptr<DeviceFilter> filter_;
  int streams_ = 1, channels_ = 1, inFmt_ = kF32, outFmt_ = kF32;
  bool fused_ = false;
  int cuCount_ = 256;
  int cg_ = 1, groups_ = 1;        // fused path: channels per workgroup, groups per stream
  std::size_t wgCapacity_ = 256;   // fused path: workgroups resident on the whole chip at once
  float *scratch_ = nullptr;       // fused path: fp32 staging planes
  std::size_t scratchBytes_ = 0;
  float *planar_ = nullptr;        // fused path, > 2 channels: per-channel fp32 timelines (planarize_kernel)
  std::size_t planarBytes_ = 0;
  void *hist_[2] = {nullptr, nullptr};
  int cur_ = 0;
  std::size_t histStride_ = 0;  // bytes per stream
  // staged-path work arrays
  cf *work_[4] = {nullptr, nullptr, nullptr, nullptr};
  std::size_t workItems_ = 0;
  // host-buffer staging
  void *stageIn_ = nullptr, *stageOut_ = nullptr;
  std::size_t stageInBytes_ = 0, stageOutBytes_ = 0;
  // timing
  std::vector<void *> evStart_, evStop_;
  long long evCount_ = 0;
};

}  // namespace miups
