#include "multi_engine.h"

#include <algorithm>

namespace miups {

std::vector<int> PartitionStreams(int streams, int slots) {
  std::vector<int> out(static_cast<std::size_t>(std::max(streams, 0)), 0);
  for (int s = 0; s < streams; ++s) {
    out[static_cast<std::size_t>(s)] = slots > 0 ? s % slots : 0;
  }
  return out;
}

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
  }
}

std::unique_ptr<MultiEngine> MultiEngine::Create(const std::vector<int> &devices, const FilterConfig &config,
                                                 const std::vector<float> &taps, int flags, int streams, int channels,
                                                 int inFmt, int outFmt, std::string *error) {
  if (devices.empty() || streams <= 0) {
    if (error) {
      *error = "need at least one device and one stream";
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
  const int G = static_cast<int>(devices.size());
  for (int i = 0; i < G; ++i) {
    std::unique_ptr<Slot> s(new Slot());
    s->index = i;
    s->device = devices[static_cast<std::size_t>(i)];
    s->streams = (streams - i + G - 1) / G;  // streams i, i+G, i+2G, ...
    if (s->streams < 0) {
      s->streams = 0;
    }
    // one filter per slot: slots on different devices cannot share tables, and slots on the same device are meant to
    // be independent (they model separate GPUs in the single-GPU tests)
    s->filter = DeviceFilter::Create(s->device, config, taps, flags, error);
    if (!s->filter) {
      return nullptr;
    }
    if (s->streams > 0) {
      s->engine = Engine::Create(s->filter, s->streams, channels, inFmt, outFmt, error);
      if (!s->engine) {
        return nullptr;
      }
    }
    m->slots_.push_back(std::move(s));
  }
  for (auto &s : m->slots_) {
    s->worker = std::thread(&MultiEngine::WorkerMain, m.get(), s.get());
  }
  return m;
}

void MultiEngine::WorkerMain(Slot *slot) {
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
    if (slot->engine) {
      const int G = static_cast<int>(slots_.size());
      // this slot's streams are s = index, index + G, ...: a strided view of the caller's buffers, no gather
      ok = slot->engine->ProcessHost(static_cast<const char *>(job.hIn) + static_cast<std::size_t>(slot->index) * job.inStride,
                                     job.inStride * static_cast<std::size_t>(G),
                                     static_cast<char *>(job.hOut) + static_cast<std::size_t>(slot->index) * job.outStride,
                                     job.outStride * static_cast<std::size_t>(G), job.blocks, &err);
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
    const Engine *e = slots_[0]->engine.get();
    const std::size_t inRow = blocks * static_cast<std::size_t>(g.n_in) * e->channels() * pcm_bytes(e->inFmt());
    const std::size_t outRow = blocks * static_cast<std::size_t>(g.B) * e->channels() * pcm_bytes(e->outFmt());
    if (inStride < inRow || outStride < outRow) {
      if (error) {
        *error = "stream stride smaller than one stream's data";
      }
      return false;
    }
  }
  {
    std::lock_guard<std::mutex> lock(mu_);
    job_ = Job{hIn, hOut, inStride, outStride, blocks};
    pending_ = static_cast<int>(slots_.size());
    for (auto &s : slots_) {
      s->hasJob = true;
    }
  }
  cvJob_.notify_all();
  std::unique_lock<std::mutex> lock(mu_);
  cvDone_.wait(lock, [&] { return pending_ == 0; });
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
  for (auto &s : slots_) {
    if (!s->filter->SetEq(apoText, fsOut, error)) {
      return false;
    }
  }
  return true;
}

bool MultiEngine::Reset(std::string *error) {
  for (auto &s : slots_) {
    if (s->engine && !s->engine->Reset(error)) {
      return false;
    }
  }
  return true;
}

}  // namespace miups
