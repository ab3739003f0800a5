// Multi-GPU host of the batched engine: independent streams sharded over the GPUs of one node, no collective.
//
// Partition: stream s runs on slot s mod G, G = number of (device) slots; a slot owns one device, its own
// copy of the filter tables, the histories of its streams, one engine with its HIP streams and pinned-buffer pipeline,
// and one host worker thread. Nothing is shared between slots but the read-only caller buffers, and nothing is
// exchanged: the reference already treats channels as independent objects (alsa_streamer_main.cpp:248-250,537-553).
// The same device may be listed more than once (two slots on one GPU) -- used by the tests to exercise the
// partition on a single-GPU box.
#pragma once

#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "engine.h"

namespace miups {

// slot of every stream under the static block-cyclic partition
std::vector<int> PartitionStreams(int streams, int slots);

class MultiEngine {
 public:
  ~MultiEngine();
  static std::unique_ptr<MultiEngine> Create(const std::vector<int> &devices, const FilterConfig &config,
                                             const std::vector<float> &taps, int flags, int streams, int channels,
                                             int inFmt, int outFmt, std::string *error);
  // all streams, host buffers: stream s at base + s*stride. Every slot runs its streams concurrently; returns when
  // all have finished. Not re-entrant.
  bool ProcessHost(const void *hIn, std::size_t inStride, void *hOut, std::size_t outStride, std::size_t blocks,
                   std::string *error);
  bool SetEq(const std::string &apoText, double fsOut, std::string *error);  // every slot's filter
  bool Reset(std::string *error);
  int slots() const { return static_cast<int>(slots_.size()); }
  int streams() const { return streams_; }
  int deviceOfStream(int s) const { return slots_[static_cast<std::size_t>(s % slots())]->device; }
  const Geometry &geometry() const { return slots_[0]->filter->geometry(); }
  bool fused() const { return slots_[0]->engine && slots_[0]->engine->fused(); }

 private:
  struct Job {
    const void *hIn = nullptr;
    void *hOut = nullptr;
    std::size_t inStride = 0, outStride = 0, blocks = 0;
  };
  struct Slot {
    int index = 0, device = 0, streams = 0;
    std::shared_ptr<DeviceFilter> filter;
    std::unique_ptr<Engine> engine;  // null when the slot has no stream (more slots than streams)
    std::thread worker;
    bool hasJob = false, ok = true;
    std::string error;
  };
  MultiEngine() = default;
  void WorkerMain(Slot *slot);

  std::vector<std::unique_ptr<Slot>> slots_;
  int streams_ = 0;
  std::mutex mu_;
  std::condition_variable cvJob_, cvDone_;
  Job job_;
  int pending_ = 0;
  bool quit_ = false;
};

}  // namespace miups
