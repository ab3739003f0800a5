// Two-thread stress test of the SPSC PCM ring (csrc/host/pcm_ring.h), built with -fsanitize=thread by the test.
// Scenario as in the reference's tests/cpp/audio/test_audio_ring_buffer.cpp:212-349 (a producer and a consumer move a
// numbered byte stream through a small ring in random chunk sizes; every byte must arrive once, in order), run twice:
// through the copying Write / Read, and through the span interface (work in place in the ring's memory, then commit) with
// caller-provided storage -- the path the streaming loop hands to the engine.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <random>
#include <thread>
#include <vector>

#include "host/pcm_ring.h"

namespace {

std::uint8_t Pattern(std::size_t i) { return static_cast<std::uint8_t>(i * 2654435761u >> 24); }

bool Stress(bool spans) {
  miups::PcmRing ring;
  std::vector<std::uint8_t> storage(4099);  // prime-sized: wrap-around at every offset
  ring.Init(storage.size(), spans ? storage.data() : nullptr);
  const std::size_t total = 8u << 20;
  std::atomic<bool> failed{false};
  std::thread producer([&] {
    std::mt19937 rng(1);
    std::vector<std::uint8_t> chunk(2048);
    std::size_t sent = 0;
    while (sent < total && !failed.load()) {
      const std::size_t n = std::min<std::size_t>(1 + rng() % chunk.size(), total - sent);
      if (spans) {
        miups::PcmRing::Span s[2];
        if (ring.WritableSpans(s) < n) {
          std::this_thread::yield();
          continue;
        }
        for (std::size_t i = 0; i < n; ++i) {  // in place, across the wrap
          (i < s[0].size ? s[0].data[i] : s[1].data[i - s[0].size]) = Pattern(sent + i);
        }
        ring.CommitWrite(n);
      } else {
        for (std::size_t i = 0; i < n; ++i) {
          chunk[i] = Pattern(sent + i);
        }
        while (!ring.Write(chunk.data(), n)) {
          std::this_thread::yield();
        }
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
      if (spans) {
        miups::PcmRing::Span s[2];
        const std::size_t n = std::min(want, ring.ReadableSpans(s));
        if (n == 0) {
          std::this_thread::yield();
          continue;
        }
        for (std::size_t i = 0; i < n; ++i) {
          const std::uint8_t v = i < s[0].size ? s[0].data[i] : s[1].data[i - s[0].size];
          if (v != Pattern(got + i)) {
            std::fprintf(stderr, "span byte %zu corrupted\n", got + i);
            failed.store(true);
            return;
          }
        }
        ring.CommitRead(n);
        got += n;
        continue;
      }
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
        if (chunk[i] != Pattern(got + i)) {
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
  return !failed.load();
}

// DiscardAll is a consumer-side operation: legal while the producer keeps writing. Afterwards the stream continues at a
// byte the producer wrote AFTER some point -- never a torn or repeated one.
bool DiscardWhileProducing() {
  miups::PcmRing ring;
  ring.Init(1 << 12);
  std::atomic<bool> stop{false};
  std::thread producer([&] {
    std::uint8_t chunk[64];
    std::size_t sent = 0;
    while (!stop.load()) {
      for (std::size_t i = 0; i < sizeof(chunk); ++i) {
        chunk[i] = Pattern(sent + i);
      }
      if (ring.Write(chunk, sizeof(chunk))) {
        sent += sizeof(chunk);
      }
    }
  });
  bool ok = true;
  std::size_t cursor = 0;  // stream position of the next byte the consumer will read
  std::uint8_t got[64];
  for (int round = 0; round < 2000 && ok; ++round) {
    if (round % 7 == 0) {
      // positions are multiples of 64 on both sides, so what was discarded is a whole number of chunks: resynchronise on
      // the next chunk by searching its position (bounded: the ring holds 64 chunks)
      ring.DiscardAll();
      while (!ring.Read(got, sizeof(got))) {
        std::this_thread::yield();
      }
      bool found = false;
      for (std::size_t k = 0; k < 100000 && !found; ++k) {
        const std::size_t pos = cursor + 64 * k;
        found = true;
        for (std::size_t i = 0; i < sizeof(got) && found; ++i) {
          found = got[i] == Pattern(pos + i);
        }
        if (found) {
          cursor = pos + 64;
        }
      }
      ok = found;
      continue;
    }
    while (!ring.Read(got, sizeof(got))) {
      std::this_thread::yield();
    }
    for (std::size_t i = 0; i < sizeof(got) && ok; ++i) {
      ok = got[i] == Pattern(cursor + i);
    }
    cursor += sizeof(got);
  }
  stop.store(true);
  producer.join();
  return ok;
}

}  // namespace

int main() {
  bool ok = Stress(false) && Stress(true) && DiscardWhileProducing();
  // single-thread contract: all-or-nothing writes and reads, spans, discard
  miups::PcmRing r;
  r.Init(8);
  const std::uint8_t a[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  std::uint8_t b[8] = {0};
  ok = ok && r.Write(a, 5) && !r.Write(a, 4) && r.AvailableToRead() == 5 && r.AvailableToWrite() == 3 && !r.Read(b, 6) &&
       r.Read(b, 3) && b[0] == 1 && b[2] == 3 && r.Write(a, 6) && r.AvailableToWrite() == 0 && r.Read(b, 8) && b[0] == 4 &&
       b[1] == 5 && b[2] == 1 && b[7] == 6;
  miups::PcmRing::Span s[2];
  ok = ok && r.WritableSpans(s) == 8 && s[0].size == 5 && s[1].size == 3;  // position 11 mod 8 = 3: 5 to the end, 3 from the base
  r.Write(a, 2);
  r.DiscardAll();
  ok = ok && r.AvailableToRead() == 0 && r.AvailableToWrite() == 8 && r.ReadableSpans(s) == 0;
  std::puts(ok ? "OK" : "FAILED");
  return ok ? 0 : 1;
}
