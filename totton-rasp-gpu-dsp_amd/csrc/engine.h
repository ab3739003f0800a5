// Host-side engine: owns the device-resident filter tables and the per-stream
// history, selects the kernel family for the geometry and enqueues the launches.
// HIP only -- there is no CPU path behind these classes.
#pragma once

#include <complex>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "device/common.h"
#include "host/filter_config.h"
#include "host/spectrum.h"

namespace miups {

// thread-local "last error" shared by the C ABI
void SetLastError(const std::string &message);
const std::string &LastError();

int DeviceCount();

// One immutable set of device tables = one (filter, EQ) state. Engines take a snapshot per call and keep it alive until
// the work they enqueued with it has finished, so an EQ change never frees or rewrites memory a running kernel reads:
// it fills ANOTHER set (a pooled one when a retired set of the same shape is free, else a new allocation), uploads it on
// a private stream and swaps the filter's pointer. No device-wide synchronisation, no stall of the audio streams; the
// first block enqueued after the swap uses the new spectrum (reference: RELOAD semantics of web/routers/eq.py:220-249,
// src/zmq/zmq_server_main.cpp:168-172 -- the data plane there has no call site).
struct TableSet {
  int device = 0;
  unsigned long long generation = 0;
  cf *Gs = nullptr, *Gc = nullptr, *Wm = nullptr, *tw = nullptr, *WmT = nullptr, *selfW = nullptr;
  int *blockB = nullptr;
  f4 *GT = nullptr, *G0 = nullptr;
  // two-level path (device/kernels_tiled.h, K = 2^15 .. 2^18): Gs / Gc / Wm in its [k1][k2] bin order; else one element
  cf *tGs = nullptr, *tGc = nullptr, *tWm = nullptr;
  cf wb{1.0f, 0.0f}, wself{1.0f, 0.0f};
  std::size_t count[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // elements per array (pool reuse needs the same shape)
  FusedTables fused() const { return FusedTables{tw, WmT, blockB, GT, G0, wb, selfW, wself}; }
  ~TableSet();
};
struct TablePool;  // retired sets of one filter, free for reuse

// What folding an EQ into the FIR dropped (host/eq.h EqFold; include/mi_upsampler.h mi_eq_residual). active = false: no EQ.
struct EqReport {
  bool active = false;
  double tailL1 = 0.0, tailL2 = 0.0, responseDev = 0.0;
  bool tailComplete = true;
  std::size_t taper = 0, firTaps = 0;
  double limit = 0.0;
  bool overLimit = false;
};
std::string EqReportWarning(const EqReport &r);  // "" unless r.overLimit

// Device-resident filter state shared by engines.
class DeviceFilter {
 public:
  ~DeviceFilter();
  static std::shared_ptr<DeviceFilter> Create(int device, const FilterConfig &config, std::vector<float> taps,
                                              int flags, std::string *error, const std::string &apoText = std::string(),
                                              double fsOut = 0.0, double eqLimit = -1.0, bool eqStrict = false);
  // Swap in tables with (or without, when text is empty) the EQ folded in. On failure the current tables stay.
  // The EQ becomes part of the FIR (cascade recursion over the taps, cut back to `taps` samples: host/eq.h). When the cut
  // drops more than the limit (SetEqLimit, default -60 dB of the ideal response in the 1-norm) the change still goes
  // through and EqReportWarning(eqReport()) says by how much -- unless the filter is strict: then it fails and nothing changes.
  bool SetEq(const std::string &apoText, double fsOut, std::string *error);
  void SetEqLimit(double maxTailL1, bool strict);
  EqReport eqReport() const;  // of the published tables
  // The same in two steps, for callers that change several filters together (MultiEngine: all slots or none): Stage
  // builds and uploads the new table set without touching the live one; Publish swaps it in (cannot fail). A staged
  // set that is dropped is freed.
  struct Staged {
    std::unique_ptr<TableSet> set;
    Geometry geo{};
    bool hasFused = false, fusedSplit = false, fusedNarrow = false, fusedR32 = false;
    EqReport report;
  };
  bool Stage(const std::string &apoText, double fsOut, Staged *out, std::string *error);
  void Publish(Staged *staged);
  // A private copy with its own tables and EQ (used when one handle of several changes its EQ; empty text = no EQ).
  std::shared_ptr<DeviceFilter> Fork(const std::string &apoText, double fsOut, std::string *error) const;

  int device() const { return device_; }
  const FilterConfig &config() const { return config_; }
  const Geometry &geometry() const { return geo_; }  // fixed by the filter file; an EQ change never alters it
  bool hasFused() const { return hasFused_; }
  bool fusedSplit() const { return fusedSplit_; }  // tables are laid out for fused_split_kernel
  bool fusedNarrow() const { return fusedNarrow_; }  // ... for the one-butterfly-per-thread form
  bool fusedR32() const { return fusedR32_; }        // ... for the radix-32 pass plan (K = 8192, 16384)
  // current tables; the caller keeps the pointer for as long as enqueued work may read them
  std::shared_ptr<const TableSet> tables() const;
  unsigned long long generation() const;  // bumped by every successful SetEq
  // test hook: make the next table upload fail after the host-side build (a failed swap must leave the filter usable)
  void FailNextUploadForTest() { failNextUpload_ = true; }

 private:
  DeviceFilter() = default;
  bool StageTables(const std::vector<double> *totalFir, Staged *out, std::string *error);

  int device_ = 0;
  FilterConfig config_{};
  std::vector<float> taps_;
  int flags_ = 0;
  Geometry geo_{};
  bool hasFused_ = false, fusedSplit_ = false, fusedNarrow_ = false, fusedR32_ = false;
  mutable std::mutex mu_;
  std::shared_ptr<const TableSet> cur_;
  std::shared_ptr<TablePool> pool_;
  unsigned long long generation_ = 0;
  void *uploadStream_ = nullptr;  // hipStream_t, non-blocking: uploads never order against the audio streams
  bool failNextUpload_ = false;
  double eqLimit_ = 1.0e-3;  // eq::kFoldDefaultLimit
  bool eqStrict_ = false;
  EqReport report_;
};

// Evaluate an APO profile's cascade on the device: bins 0..numBins-1 (fp64).
bool EqResponseDevice(int device, const std::string &apoText, std::size_t numBins, std::size_t fullFft, double fsOut,
                      std::vector<std::complex<double>> *out, std::string *error);

// Experiment switches of profiles/ (MIUPS_EXP_*): read from the environment ONCE, when an engine is created -- never on
// the per-call path. Defaults are the measured best; none changes results.
struct ExpSwitches {
  bool stereoExt = false, park = false, noPhaseParts = false, noTiledInterleave = false, noRowsInterleave = false,
       noSplitPlanar = false, noTwoLevel = false, twoLevelNoPlanar = false, twoLevelStoreForward = false,
       hostThreeStreams = false, noCoopFrames = false, forceCoopFrames = false, pitchedAnyWidth = false;
  int coopCap = 0;
  int tileTi = 0, chunkMb = 0, chunkRounds = 0, twoLevelBudgetMb = 0, hostSubBatches = 0;
  int pipeline = -1;  // 0 = never, 1 = always, -1 = by shape
  static ExpSwitches FromEnvironment();
};

class Engine {
 public:
  ~Engine();
  static std::unique_ptr<Engine> Create(std::shared_ptr<DeviceFilter> filter, int streams, int channels, int inFmt,
                                        int outFmt, std::string *error);
  std::unique_ptr<Engine> Clone(std::string *error);  // deep copy of the history (after everything enqueued so far)
  // Replace the carried history by the caller's: for every stream the last hist_frames() input frames before the next
  // block, interleaved PCM in the engine's input format, stream s at h + s*streamStride (host memory). What a
  // time-sharded caller (MultiEngine, kSplitTime) needs: a shard's left context is input, never output.
  bool LoadHistoryHost(const void *h, std::size_t streamStride, std::string *error);
  std::size_t histFrames() const { return static_cast<std::size_t>(filter_->geometry().hist_frames); }
  // Switch to another filter of the SAME geometry class between blocks (rate-family / phase switching with resident
  // spectra, EQ forks). The history is input-domain, so it carries over when the history length matches; otherwise
  // (or with resetHistory) it is zeroed, as after LoadFilter in the reference.
  bool Rebind(std::shared_ptr<DeviceFilter> filter, bool resetHistory, std::string *error);

  bool Reset(std::string *error);
  // Stream contract: all state of an engine (history, staging planes) is ordered by the engine itself. Calls may use
  // different streams: a call on another stream than the previous one first waits (hipStreamWaitEvent) for the
  // previous call's work. One engine must not be entered from two host threads at once.
  bool ProcessDevice(const void *dIn, std::size_t inStride, void *dOut, std::size_t outStride, std::size_t blocks,
                     void *hipStream, std::string *error);
  // Host buffers: H2D / kernels / D2H of consecutive sub-batches overlap on three streams through double-buffered
  // device staging. Pinned host memory (mi_host_alloc) is copied by DMA directly; pageable memory works, slower.
  // inFramePitch / outFramePitch (bytes, 0 = packed frames): the caller's frames are WIDER than this engine's -- the
  // engine takes a contiguous channel group out of them (hIn / hOut already point at the group's first channel): the
  // host copies become pitched 2-D copies (hipMemcpy2DAsync), nothing is de-interleaved on the host and no byte of
  // another group's channels crosses this device's link (MultiEngine, channel split).
  bool ProcessHost(const void *hIn, std::size_t inStride, void *hOut, std::size_t outStride, std::size_t blocks,
                   std::string *error, std::size_t inFramePitch = 0, std::size_t outFramePitch = 0);

  const std::shared_ptr<DeviceFilter> &filter() const { return filter_; }
  bool fused() const { return fused_; }
  int streams() const { return streams_; }
  int channels() const { return channels_; }
  int inFmt() const { return inFmt_; }
  int outFmt() const { return outFmt_; }
  // kernel timing: up to `slots` most recent calls keep a hipEvent pair around
  // their main kernel(s), recorded on the stream the kernels are launched on
  bool EnableTiming(int slots, std::string *error);
  // every n-th process call carries the event pair (n >= 1): a pair costs ~8 us of stream time at the headline shape
  // (scripts/step_overhead.py, profiles/r03_n_step_overhead.txt), a timed region may not want it on every call
  // ProcessHost of a single sub-batch on the engine's own stream (default) or through the three-stream pipeline; the
  // phase-split kernels for small calls (PickChannelGroup) on or off. Test / experiment hooks: both default to on.
  void SetHostOneStream(bool on) { hostOneStream_ = on; }
  // test hook: the next ProcessHost fails after issuing the host-to-device copies of sub-batch `j` (0-based), as if the
  // runtime had refused a call there -- the error path must drain its streams before the caller's buffers are unpinned
  void FailHostCallAtForTest(int j) { failAtSubBatch_ = j; }
  void SetSmallCallSplit(bool on) { smallCallSplit_ = on; }
  // workgroups per (block, stream, channel) of the latest fused call: 0 = the plain form, >= 2 = phase-split (small calls)
  int lastPhaseParts() const { return parts_; }
  // the latest staged call ran the two-level transforms (device/kernels_tiled.h) rather than one launch per pass
  bool lastTwoLevel() const { return lastTwoLevel_; }
  // the latest fused call let the transform kernel's own workgroups assemble frames (cooperative frames)
  bool lastCoopFrames() const { return lastCoop_; }
  void SetTimingStride(int every) { timingEvery_ = every < 1 ? 1 : every; }
  double LastKernelMs();
  // Per kernel class of the LATEST call: [0] planarize, [1] transform, [2] frame assembly (interleave_*), [3] history
  // carry -- each launch bracketed by its own event pair on the stream it runs on (so: a diagnostic, the extra event
  // records perturb the call; classes that overlap on two streams add up to more than the call). -1 where a class has
  // no launch. Off by default.
  void EnableClassTiming(bool on) { classTiming_ = on; }
  bool LastClassMs(double out[4]);
  // average/min/max over the recorded calls since EnableTiming (waits for them)
  bool KernelMsStats(double *avg, double *mn, double *mx, int *count);
  // generation of the filter tables the most recent ProcessDevice call used
  unsigned long long lastGeneration() const { return lastGeneration_; }

 private:
  Engine() = default;
  // work_[0..3] for `items` channel-blocks; fourth = false (two-level path): no second phase-sized buffer
  bool EnsureWork(std::size_t items, std::string *error, bool fourth = true);
  void PickChannelGroup(std::size_t blocks);
  bool PlanarizeInput(const Geometry &g, const IoDesc &io, std::size_t blocks, bool splitPlanar, void *stream, IoDesc *ioF,
                      std::string *error);
  bool LaunchFrames(const Geometry &g, const IoDesc &ioF, float *planes, std::size_t p0, std::size_t np, bool split,
                    bool quad, void *stream, std::string *error, int forceTi = 0);
  bool EnsureStreams(std::string *error);
  void *TakeEvent();                       // hipEvent_t from the pool
  void Reap(bool all);                     // release table snapshots of finished calls
  bool OrderAfterLast(void *stream, std::string *error);
  bool MarkDone(void *stream, std::shared_ptr<const TableSet> tabs, std::string *error);

  std::shared_ptr<DeviceFilter> filter_;
  ExpSwitches exp_;
  int streams_ = 1, channels_ = 1, inFmt_ = kF32, outFmt_ = kF32;
  bool fused_ = false;
  int cuCount_ = 256;
  int cg_ = 1, groups_ = 1;        // fused path: channels per workgroup, groups per stream
  bool hostOneStream_ = true;
  int failAtSubBatch_ = -1;
  bool smallCallSplit_ = true;
  bool lastTwoLevel_ = false;
  int parts_ = 0;                  // fused path, small calls: workgroups per (block, stream, channel) (phase-split), else 0
  std::size_t wgCapacity_ = 256;   // fused path: workgroups resident on the whole chip at once
  float *scratch_ = nullptr;       // fused path: fp32 staging planes (two halves when launches are pipelined)
  std::size_t scratchBytes_ = 0;
  float *planar_ = nullptr;        // fused path, > 2 channels: per-channel fp32 timelines (planarize_kernel)
  std::size_t planarBytes_ = 0;
  void *fsync_ = nullptr;          // cooperative frames: one FrameSync per (stream, block) pair of a launch (device/frame_tile.h)
  std::size_t fsyncPairs_ = 0;
  bool lastCoop_ = false;
  void *park_ = nullptr;           // split form: parked first-pass inputs of the second half transforms (IoDesc::park)
  std::size_t parkBytes_ = 0;
  void *hist_[2] = {nullptr, nullptr};
  int cur_ = 0;
  std::size_t histStride_ = 0;  // bytes per stream
  // staged-path work arrays
  cf *work_[4] = {nullptr, nullptr, nullptr, nullptr};
  std::size_t workItems_ = 0;
  bool workFourth_ = false;  // work_[3] is allocated for workItems_ items
  // host-buffer staging: two device slots per direction
  void *stageIn_[2] = {nullptr, nullptr}, *stageOut_[2] = {nullptr, nullptr};
  std::size_t stageInBytes_ = 0, stageOutBytes_ = 0;
  // engine-owned streams: own_ (Reset/Clone/ProcessHost kernels), aux_ (interleave kernels of the pipelined
  // launches), h2d_/d2h_ (ProcessHost copies)
  void *own_ = nullptr, *aux_ = nullptr, *h2d_ = nullptr, *d2h_ = nullptr;
  // ordering + table lifetime: one event per call, newest last
  struct InFlight {
    void *done;
    void *stream;
    std::shared_ptr<const TableSet> tabs;
  };
  std::deque<InFlight> inflight_;
  std::vector<void *> eventPool_;
  // events of the pipelines: [0..1] transform kernel of a plane buffer done, [2..3] its frames written,
  // [4..5] host input slot copied in, [6..7] its kernels done, [8..9] its output copied out
  void *pipeEv_[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  unsigned long long lastGeneration_ = 0;
  bool classTiming_ = false;
  std::vector<std::pair<void *, void *>> classEv_[4];  // event pairs of the latest call, per class
  std::vector<void *> classPool_;
  bool ClassMark(int cls, void *stream, bool begin);
  // timing
  std::vector<void *> evStart_, evStop_;
  long long evCount_ = 0;
  long long timingCalls_ = 0;
  int timingEvery_ = 1;
};

// Both ends of [p, p + bytes) are page-locked host memory (the first byte alone proves nothing about the extent).
bool HostRangePageLocked(const void *p, std::size_t bytes);
// How many asynchronous host copies ProcessHost has issued, process-wide, on a range that was not page-locked while
// another such copy that may share a page with it was still in flight. The rule of DESIGN 4 says: never. Tests assert 0.
unsigned long long UnsafeHostCopies();

// Pinned host memory for ProcessHost callers (hipHostMalloc / hipHostFree).
void *HostAlloc(std::size_t bytes, std::string *error);
// Pin / unpin a caller's host buffer for DMA (hipHostRegister): what mi_host_alloc memory is from birth.
bool HostRegister(void *p, std::size_t bytes, std::string *error);
void HostUnregister(void *p);
// Measured device-to-device copy rate of `device` (GB/s, bytes read + bytes written): 16 bytes per lane, grid-stride,
// `bytes` per buffer, best of `iters` launches timed with hipEvents. The roofline's "what a plain copy reaches on this
// box" beside the HBM spec figure.
bool DeviceCopyRate(int device, std::size_t bytes, int iters, double *gbps, std::string *error);
void HostFree(void *p);

}  // namespace miups
