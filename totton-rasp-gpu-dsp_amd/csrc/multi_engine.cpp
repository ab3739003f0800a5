#include "multi_engine.h"

#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <fstream>

namespace miups {

std::vector<int> PartitionStreams(int streams, int slots) {
  std::vector<int> out(static_cast<std::size_t>(std::max(streams, 0)), 0);
  for (int s = 0; s < streams; ++s) {
    out[static_cast<std::size_t>(s)] = slots > 0 ? s % slots : 0;
  }
  return out;
}

std::vector<int> PartitionChannels(int channels, int slots) {
  std::vector<int> out(static_cast<std::size_t>(std::max(slots, 0)) + 1, 0);
  for (int i = 0; i <= slots; ++i) {
    out[static_cast<std::size_t>(i)] = slots > 0 ? static_cast<int>(static_cast<long long>(i) * channels / slots) : 0;
  }
  return out;
}

std::string DeviceLocalCpuList(int device) {
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, static_cast<int>(sizeof(bus)), device) != hipSuccess) {
    (void)hipGetLastError();
    return std::string();
  }
  std::string id(bus);
  std::transform(id.begin(), id.end(), id.begin(), [](unsigned char c) { return static_cast<char>(std::tolower(c)); });
  std::ifstream f("/sys/bus/pci/devices/" + id + "/local_cpulist");
  std::string list;
  std::getline(f, list);
  return list;
}

namespace {

// "0-15,128-143" -> affinity of the calling thread, intersected with what the process may use. Returns the list
// actually applied ("" when nothing was changed: unknown topology, or no CPU of the list is allowed here).
std::string PinThisThread(const std::string &cpulist) {
  if (cpulist.empty()) {
    return std::string();
  }
  cpu_set_t allowed, want;
  CPU_ZERO(&allowed);
  CPU_ZERO(&want);
  if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) {
    return std::string();
  }
  std::size_t pos = 0;
  int picked = 0;
  while (pos < cpulist.size()) {
    char *end = nullptr;
    const long a = std::strtol(cpulist.c_str() + pos, &end, 10);
    if (end == cpulist.c_str() + pos) {
      break;
    }
    long b = a;
    pos = static_cast<std::size_t>(end - cpulist.c_str());
    if (pos < cpulist.size() && cpulist[pos] == '-') {
      b = std::strtol(cpulist.c_str() + pos + 1, &end, 10);
      pos = static_cast<std::size_t>(end - cpulist.c_str());
    }
    for (long c = a; c <= b && c < CPU_SETSIZE; ++c) {
      if (c >= 0 && CPU_ISSET(static_cast<int>(c), &allowed)) {
        CPU_SET(static_cast<int>(c), &want);
        ++picked;
      }
    }
    if (pos < cpulist.size() && cpulist[pos] == ',') {
      ++pos;
    } else {
      break;
    }
  }
  if (picked == 0 || pthread_setaffinity_np(pthread_self(), sizeof(want), &want) != 0) {
    return std::string();
  }
  return cpulist;
}

}  // namespace

MultiEngine::~MultiEngine() {
  {
    std::lock_guard<std::mutex> lock(mu_);
    quit_ = true;
  }
  cvJob_.notify_all();
  for (auto &s : slots_) {
    if (s->worker.joinable()) {
      s->worker.join();
    }
    if (s->context) {
      (void)hipSetDevice(s->device);
      HostFree(s->context);
      s->context = nullptr;
    }
  }
}

std::unique_ptr<MultiEngine> MultiEngine::Create(const std::vector<int> &devices, const FilterConfig &config,
                                                 const std::vector<float> &taps, int flags, int streams, int channels,
                                                 int inFmt, int outFmt, std::string *error, int split) {
  if (devices.empty() || streams <= 0 || channels <= 0) {
    if (error) {
      *error = "need at least one device, one stream and one channel";
    }
    return nullptr;
  }
  if (split != kSplitStreams && split != kSplitChannels && split != kSplitTime) {
    if (error) {
      *error = "unknown partition";
    }
    return nullptr;
  }
  const int visible = DeviceCount();
  for (int d : devices) {
    if (d < 0 || d >= visible) {
      if (error) {
        *error = "device " + std::to_string(d) + " requested but only " + std::to_string(visible) +
                 " HIP device(s) are visible";
      }
      return nullptr;
    }
  }
  std::unique_ptr<MultiEngine> m(new MultiEngine());
  m->streams_ = streams;
  m->channels_ = channels;
  m->split_ = split;
  m->inFmt_ = inFmt;
  m->outFmt_ = outFmt;
  const int G = static_cast<int>(devices.size());
  const std::vector<int> groups = PartitionChannels(channels, G);
  for (int i = 0; i < G; ++i) {
    std::unique_ptr<Slot> s(new Slot());
    s->index = i;
    s->device = devices[static_cast<std::size_t>(i)];
    if (split == kSplitChannels) {
      s->streams = streams;
      s->c0 = groups[static_cast<std::size_t>(i)];
      s->nch = groups[static_cast<std::size_t>(i) + 1] - s->c0;
    } else if (split == kSplitTime) {
      s->streams = streams;  // every slot sees every stream, a block range of it
      s->c0 = 0;
      s->nch = channels;
    } else {
      s->streams = std::max(0, (streams - i + G - 1) / G);  // streams i, i+G, i+2G, ...
      s->c0 = 0;
      s->nch = channels;
    }
    // one filter per slot: slots on different devices cannot share tables, and slots on the same device are meant to
    // be independent (they model separate GPUs in the single-GPU tests)
    s->filter = DeviceFilter::Create(s->device, config, taps, flags, error);
    if (!s->filter) {
      return nullptr;
    }
    if (s->streams > 0 && s->nch > 0) {
      s->engine = Engine::Create(s->filter, s->streams, s->nch, inFmt, outFmt, error);
      if (!s->engine) {
        return nullptr;
      }
      if (std::getenv("MIUPS_EXP_MULTI_THREE_STREAMS") != nullptr) {  // experiment switch (scripts/multi_stress.py)
        s->engine->SetHostOneStream(false);
        s->engine->SetSmallCallSplit(false);
      }
    }
    m->slots_.push_back(std::move(s));
  }
  if (split == kSplitTime) {
    m->histBytes_ = m->slots_[0]->engine->histFrames() * static_cast<std::size_t>(channels) * pcm_bytes(inFmt);
    m->tail_.assign(m->histBytes_ * static_cast<std::size_t>(streams), 0);
    for (auto &s : m->slots_) {
      (void)hipSetDevice(s->device);
      s->context = static_cast<char *>(HostAlloc(std::max<std::size_t>(m->tail_.size(), 1), error));
      if (!s->context) {
        return nullptr;
      }
      std::memset(s->context, 0, m->tail_.size());
    }
  }
  for (auto &s : m->slots_) {
    s->worker = std::thread(&MultiEngine::WorkerMain, m.get(), s.get());
  }
  return m;
}

bool MultiEngine::fused() const {
  for (const auto &s : slots_) {
    if (s->engine) {
      return s->engine->fused();
    }
  }
  return false;
}

int MultiEngine::deviceOfChannel(int c) const {
  for (const auto &s : slots_) {
    if (c >= s->c0 && c < s->c0 + s->nch) {
      return s->device;
    }
  }
  return -1;
}

void MultiEngine::WorkerMain(Slot *slot) {
  // the worker feeds its device from host memory: keep it on the CPUs (and the memory controller) next to that device
  const std::string pinned = PinThisThread(DeviceLocalCpuList(slot->device));
  {
    std::lock_guard<std::mutex> lock(mu_);
    slot->cpus = pinned;
  }
  for (;;) {
    Job job;
    {
      std::unique_lock<std::mutex> lock(mu_);
      cvJob_.wait(lock, [&] { return quit_ || slot->hasJob; });
      if (quit_) {
        return;
      }
      job = job_;
    }
    bool ok = true;
    std::string err;
    std::unique_lock<std::mutex> turn(serialMu_, std::defer_lock);
    if (job.serial) {
      turn.lock();
    }
    if (slot->engine) {
      const std::size_t G = slots_.size();
      if (split_ == kSplitTime) {
        // blocks [b0, b1) of every stream; the hist_frames input frames in front of b0 come from the caller's buffer and,
        // where the range starts less than that into the call, from the tail kept of the previous call
        const std::size_t b0 = job.blocks * static_cast<std::size_t>(slot->index) / G;
        const std::size_t b1 = job.blocks * (static_cast<std::size_t>(slot->index) + 1) / G;
        if (b1 > b0) {
          const Geometry &g = geometry();
          const std::size_t frameIn = static_cast<std::size_t>(channels_) * pcm_bytes(inFmt_);
          const std::size_t frameOut = static_cast<std::size_t>(channels_) * pcm_bytes(outFmt_);
          const std::size_t startBytes = b0 * static_cast<std::size_t>(g.n_in) * frameIn;  // of the range, inside the call
          for (int st = 0; st < streams_ && histBytes_ > 0; ++st) {
            char *dst = slot->context + static_cast<std::size_t>(st) * histBytes_;
            const char *in = static_cast<const char *>(job.hIn) + static_cast<std::size_t>(st) * job.inStride;
            const std::size_t fromCall = std::min(startBytes, histBytes_);   // newest part: the call's own frames
            const std::size_t fromTail = histBytes_ - fromCall;              // oldest part: end of the previous call's tail
            if (fromTail) {
              std::memcpy(dst, tail_.data() + static_cast<std::size_t>(st) * histBytes_ + fromCall, fromTail);
            }
            std::memcpy(dst + fromTail, in + startBytes - fromCall, fromCall);
          }
          ok = (histBytes_ == 0 || slot->engine->LoadHistoryHost(slot->context, histBytes_, &err)) &&
               slot->engine->ProcessHost(static_cast<const char *>(job.hIn) + startBytes, job.inStride,
                                         static_cast<char *>(job.hOut) + b0 * static_cast<std::size_t>(g.B) * frameOut,
                                         job.outStride, b1 - b0, &err);
        }
      } else if (split_ == kSplitChannels) {
        // this slot's channel group: a column of the caller's frames -- pitched copies, no gather on the host
        const std::size_t ib = static_cast<std::size_t>(pcm_bytes(inFmt_)), ob = static_cast<std::size_t>(pcm_bytes(outFmt_));
        ok = slot->engine->ProcessHost(static_cast<const char *>(job.hIn) + static_cast<std::size_t>(slot->c0) * ib,
                                       job.inStride, static_cast<char *>(job.hOut) + static_cast<std::size_t>(slot->c0) * ob,
                                       job.outStride, job.blocks, &err, static_cast<std::size_t>(channels_) * ib,
                                       static_cast<std::size_t>(channels_) * ob);
      } else {
        // this slot's streams are s = index, index + G, ...: a strided view of the caller's buffers, no gather
        ok = slot->engine->ProcessHost(
            static_cast<const char *>(job.hIn) + static_cast<std::size_t>(slot->index) * job.inStride, job.inStride * G,
            static_cast<char *>(job.hOut) + static_cast<std::size_t>(slot->index) * job.outStride, job.outStride * G,
            job.blocks, &err);
      }
    }
    {
      std::lock_guard<std::mutex> lock(mu_);
      slot->hasJob = false;
      slot->ok = ok;
      slot->error = err;
      --pending_;
    }
    cvDone_.notify_all();
  }
}

bool MultiEngine::ProcessHost(const void *hIn, std::size_t inStride, void *hOut, std::size_t outStride, std::size_t blocks,
                              std::string *error) {
  if (!hIn || !hOut || blocks == 0) {
    if (error) {
      *error = "null buffer or zero blocks";
    }
    return false;
  }
  const Geometry &g = geometry();
  if (streams_ > 1) {
    // the strides are what separates one stream from the next: they must cover a stream's data
    const std::size_t inRow = blocks * static_cast<std::size_t>(g.n_in) * channels_ * pcm_bytes(inFmt_);
    const std::size_t outRow = blocks * static_cast<std::size_t>(g.B) * channels_ * pcm_bytes(outFmt_);
    if (inStride < inRow || outStride < outRow) {
      if (error) {
        *error = "stream stride smaller than one stream's data";
      }
      return false;
    }
  }
  // The workers copy DIFFERENT sub-ranges of the caller's two buffers at the same time. From pageable memory that is not
  // safe on this runtime: an asynchronous copy pins its host range on the fly, and two threads whose ranges share pages (or
  // a later copy that meets a cached pin of another sub-range) ended in "Memory access fault by GPU ... on address <host
  // page>" -- about one run of the GPU suite in four, always in the block-range partition (adjacent ranges of one buffer;
  // gpurun_out/r1.log, native stack in abort_bt.txt: the HSA runtime's fault handler). So buffers that are not page-locked
  // already (mi_host_alloc / mi_host_register) are registered here for the duration of the call; if that is refused
  // (locked-memory limit, a partly registered range) the slots run one after the other instead.
  const std::size_t inBytes = (static_cast<std::size_t>(streams_) - 1) * inStride +
                              blocks * static_cast<std::size_t>(g.n_in) * channels_ * pcm_bytes(inFmt_);
  const std::size_t outBytes = (static_cast<std::size_t>(streams_) - 1) * outStride +
                               blocks * static_cast<std::size_t>(g.B) * channels_ * pcm_bytes(outFmt_);
  (void)hipSetDevice(slots_[0]->device);
  bool regIn = false, regOut = false, serial = false;
  if (slots_.size() > 1) {
    if (!HostRangePageLocked(hIn, inBytes)) {  // both ends of the extent, not the first byte alone
      regIn = hipHostRegister(const_cast<void *>(hIn), inBytes, hipHostRegisterPortable) == hipSuccess;
      serial = !regIn;
    }
    if (!HostRangePageLocked(hOut, outBytes)) {
      regOut = hipHostRegister(hOut, outBytes, hipHostRegisterPortable) == hipSuccess;
      serial = serial || !regOut;
    }
    if (serial) {
      (void)hipGetLastError();
    }
  }
  {
    std::lock_guard<std::mutex> lock(mu_);
    job_ = Job{hIn, hOut, inStride, outStride, blocks, serial};
    pending_ = static_cast<int>(slots_.size());
    for (auto &s : slots_) {
      s->hasJob = true;
    }
  }
  cvJob_.notify_all();
  std::unique_lock<std::mutex> lock(mu_);
  // pending_ == 0: every worker has RETURNED from its engine's ProcessHost, failed ones included -- and that function never
  // returns with a copy in flight (its failure exits drain the streams first, engine.hip DrainOnFailure), so the
  // registrations below are released with no DMA running on them (tests/test_gpu_stream_host.py, injected failure).
  cvDone_.wait(lock, [&] { return pending_ == 0; });
  if (regIn) {
    (void)hipHostUnregister(const_cast<void *>(hIn));
  }
  if (regOut) {
    (void)hipHostUnregister(hOut);
  }
  if (split_ == kSplitTime && histBytes_ > 0) {
    // what the next call's first ranges will need: the last hist_frames input frames of (tail ++ this call) per stream
    const std::size_t callBytes = blocks * static_cast<std::size_t>(g.n_in) * channels_ * pcm_bytes(inFmt_);
    for (int st = 0; st < streams_; ++st) {
      char *t = tail_.data() + static_cast<std::size_t>(st) * histBytes_;
      const char *in = static_cast<const char *>(hIn) + static_cast<std::size_t>(st) * inStride;
      if (callBytes >= histBytes_) {
        std::memcpy(t, in + callBytes - histBytes_, histBytes_);
      } else {
        std::memmove(t, t + callBytes, histBytes_ - callBytes);
        std::memcpy(t + histBytes_ - callBytes, in, callBytes);
      }
    }
  }
  for (auto &s : slots_) {
    if (!s->ok) {
      if (error) {
        *error = "device " + std::to_string(s->device) + ": " + s->error;
      }
      return false;
    }
  }
  return true;
}

bool MultiEngine::SetEq(const std::string &apoText, double fsOut, std::string *error) {
  // phase 1: every slot's new tables, built and uploaded beside the live ones. phase 2: publish them all. A failure
  // in phase 1 drops what was staged: no slot has changed. Called from the caller's thread between
  // ProcessHost calls (ProcessHost is synchronous), so no worker is inside an engine call here.
  std::vector<DeviceFilter::Staged> staged(slots_.size());
  for (std::size_t i = 0; i < slots_.size(); ++i) {
    if (failEqSlot_ == static_cast<int>(i)) {
      failEqSlot_ = -1;
      slots_[i]->filter->FailNextUploadForTest();
    }
    if (!slots_[i]->filter->Stage(apoText, fsOut, &staged[i], error)) {
      if (error) {
        *error = "slot " + std::to_string(i) + " (device " + std::to_string(slots_[i]->device) + "): " + *error +
                 " -- no slot was changed";
      }
      return false;
    }
  }
  for (std::size_t i = 0; i < slots_.size(); ++i) {
    slots_[i]->filter->Publish(&staged[i]);
  }
  return true;
}

bool MultiEngine::Reset(std::string *error) {
  std::fill(tail_.begin(), tail_.end(), 0);
  for (auto &s : slots_) {
    if (s->engine && !s->engine->Reset(error)) {
      return false;
    }
  }
  return true;
}

}  // namespace miups
