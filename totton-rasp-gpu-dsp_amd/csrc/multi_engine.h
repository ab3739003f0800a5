// Multi-GPU host of the batched engine: the independent units of the path sharded over the GPUs of one node, no
// collective (reference: channels are independent objects, alsa_streamer_main.cpp:247-250,536-553).
//
// Two partitions:
//   kSplitStreams   stream s runs on slot s mod G (many independent streams: BASELINE configs[3]);
//   kSplitChannels  every stream's channels are cut into G contiguous groups, group i = channels [i C / G, (i+1) C / G)
//                   runs on slot i (ONE wide stream over several GPUs: BASELINE configs[4], "32-channel, 1->8 GPU sweep").
//                   A slot reads and writes its column group straight out of / into the caller's interleaved frames with
//                   pitched 2-D copies (Engine::ProcessHost frame pitches): no host de-interleave, and each device's link
//                   carries only its own channels. Measured (profiles/r03_h_multi_split.txt): pitched DMA moves ~180 M rows
//                   per second, i.e. 11 / 6 / 3 GB/s for 64 / 32 / 16-byte rows against 54 GB/s contiguous -- right for
//                   wide groups, wrong for narrow ones; hence:
//   kSplitTime      every stream's BLOCKS are cut into G contiguous ranges, range i on slot i, all channels: the copies are
//                   contiguous (full link rate per device) and nothing is re-laid-out. A block depends on earlier blocks
//                   only through the INPUT history ("time-sharded"), so a slot needs the hist_frames input frames
//                   in front of its range: they are in the caller's buffer (or, at the start of a call, in the tail this
//                   object keeps of the previous call) and are handed to the slot's engine before it runs.
// G = number of (device) slots; a slot owns one device, its own copy of the filter tables, the histories of its
// units, one engine with its HIP streams and staging pipeline, and one host worker thread pinned to the CPUs local to
// its device (sysfs local_cpulist of the device's PCI function). Nothing is shared between slots but the caller's
// buffers, and nothing is exchanged. The same device may be listed more than once (two slots on one GPU) -- used by the
// tests to exercise the partitions on a single-GPU box.
#pragma once

#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "engine.h"

namespace miups {

enum MultiSplit : int { kSplitStreams = 0, kSplitChannels = 1, kSplitTime = 2 };

// slot of every stream under the static block-cyclic partition
std::vector<int> PartitionStreams(int streams, int slots);
// first channel of every slot's contiguous group (slots + 1 entries; group i = [out[i], out[i + 1]))
std::vector<int> PartitionChannels(int channels, int slots);
// CPUs local to a HIP device ("" when the platform does not say): /sys/bus/pci/devices/<bus id>/local_cpulist
std::string DeviceLocalCpuList(int device);

class MultiEngine {
 public:
  ~MultiEngine();
  static std::unique_ptr<MultiEngine> Create(const std::vector<int> &devices, const FilterConfig &config,
                                             const std::vector<float> &taps, int flags, int streams, int channels,
                                             int inFmt, int outFmt, std::string *error, int split = kSplitStreams);
  // all streams, host buffers: stream s at base + s*stride, frames of `channels` interleaved channels. Every slot runs
  // its share concurrently; returns when all have finished. Not re-entrant. Pageable buffers work at the runtime's
  // staging speed: pass memory from mi_host_alloc, or pin the caller's own buffer once with mi_host_register.
  bool ProcessHost(const void *hIn, std::size_t inStride, void *hOut, std::size_t outStride, std::size_t blocks,
                   std::string *error);
  // every slot's filter, all or nothing: the new tables of ALL slots are built first, then published together; if one
  // build fails no slot changes (streams of one run never play with different EQs)
  bool SetEq(const std::string &apoText, double fsOut, std::string *error);
  // what the EQ fold dropped (every slot folds the same taps: slot 0's report) and the limit / strictness of all slots
  EqReport eqReport() const { return slots_[0]->filter->eqReport(); }
  void SetEqLimit(double maxTailL1, bool strict) {
    for (auto &s : slots_) {
      s->filter->SetEqLimit(maxTailL1, strict);
    }
  }
  bool Reset(std::string *error);
  int slots() const { return static_cast<int>(slots_.size()); }
  int streams() const { return streams_; }
  int channels() const { return channels_; }
  int split() const { return split_; }
  int deviceOfStream(int s) const { return slots_[static_cast<std::size_t>(s % slots())]->device; }
  int deviceOfChannel(int c) const;
  const Geometry &geometry() const { return slots_[0]->filter->geometry(); }
  bool fused() const;
  // test hook: the next SetEq fails while building slot `slot`'s tables
  void FailNextEqOnSlotForTest(int slot) { failEqSlot_ = slot; }
  // test hook: slot `slot`'s next ProcessHost fails after the host-to-device copies of its sub-batch j
  void FailHostCallAtForTest(int slot, int j) {
    if (slot >= 0 && slot < slots() && slots_[static_cast<std::size_t>(slot)]->engine) {
      slots_[static_cast<std::size_t>(slot)]->engine->FailHostCallAtForTest(j);
    }
  }
  const std::string &workerAffinity(int slot) const { return slots_[static_cast<std::size_t>(slot)]->cpus; }

 private:
  struct Job {
    const void *hIn = nullptr;
    void *hOut = nullptr;
    std::size_t inStride = 0, outStride = 0, blocks = 0;
    bool serial = false;  // caller buffers could not be page-locked for this call: the slots take turns (see ProcessHost)
  };
  struct Slot {
    int index = 0, device = 0, streams = 0;
    int c0 = 0, nch = 0;  // channel group (the whole frame under kSplitStreams / kSplitTime)
    // kSplitTime: the slot's left context, [stream][hist_frames] frames. Page-locked (HostAlloc): it is uploaded on every
    // call from a worker thread, and a pinned source is a plain DMA -- no on-the-fly pinning inside the runtime
    char *context = nullptr;
    std::shared_ptr<DeviceFilter> filter;
    std::unique_ptr<Engine> engine;  // null when the slot has no unit (more slots than streams / channels)
    std::thread worker;
    std::string cpus;                // the CPU list the worker was pinned to ("" = not pinned)
    bool hasJob = false, ok = true;
    std::string error;
  };
  MultiEngine() = default;
  void WorkerMain(Slot *slot);

  std::vector<std::unique_ptr<Slot>> slots_;
  int streams_ = 0, channels_ = 0, split_ = kSplitStreams, inFmt_ = 0, outFmt_ = 0;
  std::mutex mu_;
  std::mutex serialMu_;  // Job::serial: one slot inside its engine at a time
  std::condition_variable cvJob_, cvDone_;
  Job job_;
  int pending_ = 0;
  bool quit_ = false;
  int failEqSlot_ = -1;
  std::vector<char> tail_;  // kSplitTime: the last hist_frames input frames of every stream seen so far (zeros at start)
  std::size_t histBytes_ = 0;  // ... per stream
};

}  // namespace miups
