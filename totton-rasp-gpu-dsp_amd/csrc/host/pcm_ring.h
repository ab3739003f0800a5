// Single-producer / single-consumer ring of raw PCM bytes (whole interleaved frames): the staging buffer between a
// capture endpoint, the batched engine and a playback endpoint.
//
// Contract of the reference's AudioRingBuffer (include/io/audio_ring_buffer.h:23-100), on bytes instead of floats
// (the engine takes PCM frames as they come from the device, so nothing is converted on the host):
//   * write() fails -- and writes nothing -- unless the whole chunk fits; read() likewise for the whole request;
//   * producer is the sole writer of tail_, consumer of head_ (relaxed); size_ is the synchronisation point:
//     sample writes happen-before size_.fetch_add(release), size_.load(acquire) happens-before sample reads;
//   * clear() is only legal while neither side is inside write()/read() (the streamer calls it from the one thread
//     that does both, exactly as the reference does, alsa_streamer_main.cpp:515-521,557-562).
#pragma once

#include <algorithm>
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

namespace miups {

class PcmRing {
 public:
  void Init(std::size_t capacityBytes) {
    buffer_.assign(capacityBytes, 0);
    head_.store(0, std::memory_order_relaxed);
    tail_.store(0, std::memory_order_relaxed);
    size_.store(0, std::memory_order_relaxed);
  }
  std::size_t capacity() const { return buffer_.size(); }
  std::size_t AvailableToRead() const { return size_.load(std::memory_order_acquire); }
  std::size_t AvailableToWrite() const { return capacity() - size_.load(std::memory_order_acquire); }

  bool Write(const void *data, std::size_t count) {  // producer
    const std::size_t cap = capacity();
    if (cap == 0 || count > AvailableToWrite()) {
      return false;
    }
    const std::size_t tail = tail_.load(std::memory_order_relaxed);
    const std::size_t first = std::min(count, cap - tail);
    std::memcpy(buffer_.data() + tail, data, first);
    if (count > first) {
      std::memcpy(buffer_.data(), static_cast<const std::uint8_t *>(data) + first, count - first);
    }
    tail_.store((tail + count) % cap, std::memory_order_relaxed);
    size_.fetch_add(count, std::memory_order_release);
    return true;
  }
  bool Read(void *dst, std::size_t count) {  // consumer
    const std::size_t cap = capacity();
    if (cap == 0 || count > AvailableToRead()) {
      return false;
    }
    const std::size_t head = head_.load(std::memory_order_relaxed);
    const std::size_t first = std::min(count, cap - head);
    std::memcpy(dst, buffer_.data() + head, first);
    if (count > first) {
      std::memcpy(static_cast<std::uint8_t *>(dst) + first, buffer_.data(), count - first);
    }
    head_.store((head + count) % cap, std::memory_order_relaxed);
    size_.fetch_sub(count, std::memory_order_release);
    return true;
  }
  void Clear() {
    head_.store(0, std::memory_order_relaxed);
    tail_.store(0, std::memory_order_relaxed);
    size_.store(0, std::memory_order_release);
  }

 private:
  std::vector<std::uint8_t> buffer_;
  std::atomic<std::size_t> head_{0}, tail_{0}, size_{0};
};

}  // namespace miups
