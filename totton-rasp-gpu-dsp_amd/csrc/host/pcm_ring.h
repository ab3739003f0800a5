// Single-producer / single-consumer staging ring for raw PCM bytes (whole interleaved frames) between a capture
// endpoint, the batched engine and a playback endpoint.
//
// What it has to do is what the reference's AudioRingBuffer does for its float samples (include/io/audio_ring_buffer.h:
// all-or-nothing writes and reads, one producer, one consumer); how it does it is this repo's own:
//
//   * two MONOTONIC 64-bit byte counters, produced_ (written only by the producer) and consumed_ (written only by the
//     consumer). Fill level = produced_ - consumed_, position = counter mod capacity. No third shared word, no
//     read-modify-write on shared state: each side publishes its own progress with one release store and observes the
//     other's with one acquire load, and the counters never wrap in practice (2^64 bytes).
//   * SPAN access besides the copying Write/Read: the producer asks for the free space, the consumer for the filled
//     space, each as at most two contiguous pieces of the ring's own memory, works in place and commits what it used.
//     The streaming loop hands those pieces straight to the engine (mi_engine_process_host reads its input out of the
//     ring and writes its output into the other ring), so a block crosses the host exactly once on each side.
//   * the storage can be the caller's: Init(capacity, memory) places the ring in memory from mi_host_alloc (page-locked),
//     so that what the engine is handed is DMA-able without the runtime's staging copy.
//   * DiscardAll() drops everything that is readable NOW. It is a consumer-side operation (it only moves consumed_), so
//     unlike a reset-both-ends clear it is safe while the producer keeps writing.
#pragma once

#include <algorithm>
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>

namespace miups {

class PcmRing {
 public:
  struct Span {
    std::uint8_t *data = nullptr;
    std::size_t size = 0;
  };

  // capacity in bytes. memory == nullptr: the ring owns a heap buffer; else `memory` must hold capacityBytes and outlive
  // the ring (pinned memory: see the header comment). Not thread-safe; call before the two sides start.
  void Init(std::size_t capacityBytes, void *memory = nullptr) {
    owned_.reset();
    capacity_ = capacityBytes;
    if (memory) {
      base_ = static_cast<std::uint8_t *>(memory);
    } else {
      owned_.reset(capacityBytes ? new std::uint8_t[capacityBytes]() : nullptr);
      base_ = owned_.get();
    }
    produced_.store(0, std::memory_order_relaxed);
    consumed_.store(0, std::memory_order_relaxed);
  }
  std::size_t capacity() const { return capacity_; }

  // either side may ask; the answer is a lower bound for the asking side's own next operation
  std::size_t AvailableToRead() const {
    return static_cast<std::size_t>(produced_.load(std::memory_order_acquire) - consumed_.load(std::memory_order_acquire));
  }
  std::size_t AvailableToWrite() const { return capacity_ - AvailableToRead(); }

  // ---- producer ---------------------------------------------------------------------------------------------------
  // free space as up to two contiguous pieces (the second one starts at the ring's base); returns the total
  std::size_t WritableSpans(Span out[2]) const {
    const std::uint64_t p = produced_.load(std::memory_order_relaxed);
    const std::size_t free_bytes = capacity_ - static_cast<std::size_t>(p - consumed_.load(std::memory_order_acquire));
    return Pieces(p, free_bytes, out);
  }
  void CommitWrite(std::size_t count) {  // the first `count` bytes of the spans are now valid
    produced_.store(produced_.load(std::memory_order_relaxed) + count, std::memory_order_release);
  }
  bool Write(const void *data, std::size_t count) {  // all or nothing
    Span s[2];
    if (capacity_ == 0 || WritableSpans(s) < count) {
      return false;
    }
    const std::size_t first = std::min(count, s[0].size);
    std::memcpy(s[0].data, data, first);
    if (count > first) {
      std::memcpy(s[1].data, static_cast<const std::uint8_t *>(data) + first, count - first);
    }
    CommitWrite(count);
    return true;
  }

  // ---- consumer ---------------------------------------------------------------------------------------------------
  std::size_t ReadableSpans(Span out[2]) const {
    const std::uint64_t c = consumed_.load(std::memory_order_relaxed);
    const std::size_t filled = static_cast<std::size_t>(produced_.load(std::memory_order_acquire) - c);
    return Pieces(c, filled, out);
  }
  void CommitRead(std::size_t count) {  // the first `count` bytes of the spans may be overwritten
    consumed_.store(consumed_.load(std::memory_order_relaxed) + count, std::memory_order_release);
  }
  bool Read(void *dst, std::size_t count) {  // all or nothing
    Span s[2];
    if (capacity_ == 0 || ReadableSpans(s) < count) {
      return false;
    }
    const std::size_t first = std::min(count, s[0].size);
    std::memcpy(dst, s[0].data, first);
    if (count > first) {
      std::memcpy(static_cast<std::uint8_t *>(dst) + first, s[1].data, count - first);
    }
    CommitRead(count);
    return true;
  }
  void DiscardAll() { consumed_.store(produced_.load(std::memory_order_acquire), std::memory_order_release); }

 private:
  std::size_t Pieces(std::uint64_t from, std::size_t count, Span out[2]) const {
    out[0] = Span();
    out[1] = Span();
    if (capacity_ == 0 || count == 0) {
      return 0;
    }
    const std::size_t at = static_cast<std::size_t>(from % capacity_);
    const std::size_t first = std::min(count, capacity_ - at);
    out[0].data = base_ + at;
    out[0].size = first;
    if (count > first) {
      out[1].data = base_;
      out[1].size = count - first;
    }
    return count;
  }

  std::unique_ptr<std::uint8_t[]> owned_;
  std::uint8_t *base_ = nullptr;
  std::size_t capacity_ = 0;
  // each counter on a cache line of its own: the producer's stores do not bounce the consumer's line and vice versa
  alignas(64) std::atomic<std::uint64_t> produced_{0};
  alignas(64) std::atomic<std::uint64_t> consumed_{0};
};

}  // namespace miups
