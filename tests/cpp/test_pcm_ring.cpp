// Two-thread stress test of the SPSC PCM ring (csrc/host/pcm_ring.h), modelled on the reference's
// tests/cpp/audio/test_audio_ring_buffer.cpp:212-349: a producer and a consumer move a numbered byte stream through a
// small ring in random chunk sizes; every byte must arrive once, in order. Built with -fsanitize=thread by the test.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <random>
#include <thread>
#include <vector>

#include "host/pcm_ring.h"

int main() {
  miups::PcmRing ring;
  ring.Init(4099);  // prime-sized: wrap-around at every offset
  const std::size_t total = 8u << 20;
  std::atomic<bool> failed{false};
  std::thread producer([&] {
    std::mt19937 rng(1);
    std::vector<std::uint8_t> chunk(2048);
    std::size_t sent = 0;
    while (sent < total && !failed.load()) {
      const std::size_t n = std::min<std::size_t>(1 + rng() % chunk.size(), total - sent);
      for (std::size_t i = 0; i < n; ++i) {
        chunk[i] = static_cast<std::uint8_t>((sent + i) * 2654435761u >> 24);
      }
      while (!ring.Write(chunk.data(), n)) {
        std::this_thread::yield();
      }
      sent += n;
    }
  });
  std::thread consumer([&] {
    std::mt19937 rng(2);
    std::vector<std::uint8_t> chunk(3000);
    std::size_t got = 0;
    while (got < total) {
      const std::size_t want = std::min<std::size_t>(1 + rng() % chunk.size(), total - got);
      const std::size_t n = std::min(want, ring.AvailableToRead());
      if (n == 0) {
        std::this_thread::yield();
        continue;
      }
      if (!ring.Read(chunk.data(), n)) {
        failed.store(true);
        break;
      }
      for (std::size_t i = 0; i < n; ++i) {
        if (chunk[i] != static_cast<std::uint8_t>((got + i) * 2654435761u >> 24)) {
          std::fprintf(stderr, "byte %zu corrupted\n", got + i);
          failed.store(true);
          return;
        }
      }
      got += n;
    }
  });
  producer.join();
  consumer.join();
  // single-thread contract: all-or-nothing writes and reads, clear
  miups::PcmRing r;
  r.Init(8);
  const std::uint8_t a[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  std::uint8_t b[8] = {0};
  bool ok = !failed.load() && r.Write(a, 5) && !r.Write(a, 4) && r.AvailableToRead() == 5 && r.AvailableToWrite() == 3 &&
            !r.Read(b, 6) && r.Read(b, 3) && b[0] == 1 && b[2] == 3 && r.Write(a, 6) && r.AvailableToWrite() == 0 &&
            r.Read(b, 8) && b[0] == 4 && b[1] == 5 && b[2] == 1 && b[7] == 6;
  r.Clear();
  ok = ok && r.AvailableToRead() == 0 && r.AvailableToWrite() == 8;
  std::puts(ok ? "OK" : "FAILED");
  return ok ? 0 : 1;
}
