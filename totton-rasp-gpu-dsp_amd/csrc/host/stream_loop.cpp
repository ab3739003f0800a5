#include "stream_loop.h"

#include <algorithm>
#include <cstdint>
#include <vector>

#include "../device/pcm.h"
#include "pcm_ring.h"

namespace miups {
namespace {

// no filter: PCM -> float -> PCM, exactly what the reference does with its float buffers
// (alsa_streamer_main.cpp:499-505,571-582)
void RoundTrip(const std::uint8_t *src, std::uint8_t *dst, int fmt, std::size_t samples, std::vector<float> *scratch) {
  scratch->resize(samples);
  for (std::size_t i = 0; i < samples; ++i) {
    (*scratch)[i] = pcm_load(src, fmt, static_cast<long long>(i));
  }
  for (std::size_t i = 0; i < samples; ++i) {
    pcm_store(dst, fmt, static_cast<long long>(i), (*scratch)[i]);
  }
}

}  // namespace

bool RunStreamLoop(const LoopParams &p, const ReadFn &read, const WriteFn &write, const ProcessFn &process,
                   const BetweenBlocksFn &between, const RunningFn &running, LoopStats *stats, const LogFn &log) {
  LoopStats local;
  LoopStats &st = stats ? *stats : local;
  st = LoopStats();
  const std::size_t frameBytes = static_cast<std::size_t>(pcm_bytes(p.format)) * p.channels;
  if (frameBytes == 0 || p.periodFrames == 0 || !read || !write) {
    return false;
  }
  const bool filtered = p.blockInFrames > 0 && p.blockOutFrames > 0 && process;
  const std::size_t factor = filtered ? std::max<std::size_t>(p.blockOutFrames / p.blockInFrames, 1) : 1;
  const std::size_t outPeriod = p.periodFrames * factor;  // frames per sink write
  const std::size_t maxBlocks = std::max<std::size_t>(p.maxBlocksPerCall, 1);

  std::vector<std::uint8_t> raw(p.periodFrames * frameBytes), outChunk(outPeriod * frameBytes);
  std::vector<float> scratch;
  PcmRing inRing, outRing;
  std::vector<std::uint8_t> inBlocks, outBlocks;  // bounce buffers: only for a batch that wraps around a ring's end
  struct RingMemory {  // ring storage from the caller's allocator (page-locked for the engine), released on every exit path
    void *p = nullptr;
    void (*release)(void *) = nullptr;
    ~RingMemory() {
      if (p && release) {
        release(p);
      }
    }
  } inMem, outMem;
  if (filtered) {
    // staging capacities: three times the larger of a block (batch) and a period (reference :476-481)
    const std::size_t inCap = std::max(p.blockInFrames * maxBlocks, p.periodFrames) * 3 * frameBytes;
    const std::size_t outCap = std::max(p.blockOutFrames * maxBlocks, outPeriod) * 3 * frameBytes;
    if (p.hostAlloc && p.hostFree) {
      inMem.p = p.hostAlloc(inCap);
      inMem.release = p.hostFree;
      outMem.p = p.hostAlloc(outCap);
      outMem.release = p.hostFree;
    }
    inRing.Init(inCap, inMem.p);
    outRing.Init(outCap, outMem.p);
    inBlocks.resize(p.blockInFrames * maxBlocks * frameBytes);
    outBlocks.resize(p.blockOutFrames * maxBlocks * frameBytes);
  }
  bool ok = true;
  std::size_t inputFrames = 0;  // frames accepted from the source (drainAtEnd: bounds the output length)

  // every complete block that is staged AND fits the output staging (:524-563); padTail: also the last partial one
  auto process_available = [&](bool padTail) -> bool {
    while (running()) {
      std::size_t avail = inRing.AvailableToRead() / (p.blockInFrames * frameBytes);
      const std::size_t room = outRing.AvailableToWrite() / (p.blockOutFrames * frameBytes);
      std::size_t tailBytes = 0;
      if (padTail && avail == 0 && inRing.AvailableToRead() > 0) {
        tailBytes = inRing.AvailableToRead();
        avail = 1;
      }
      const std::size_t k = std::min({avail, room, maxBlocks});
      if (k == 0) {
        break;
      }
      const std::size_t want = k * p.blockInFrames * frameBytes, produce = k * p.blockOutFrames * frameBytes;
      // in place where the batch is one contiguous piece of the ring, through the bounce buffer where it wraps
      PcmRing::Span rd[2], wr[2];
      inRing.ReadableSpans(rd);
      outRing.WritableSpans(wr);  // room >= k blocks was checked above
      const bool inDirect = !tailBytes && rd[0].size >= want, outDirect = wr[0].size >= produce;
      const std::uint8_t *src = rd[0].data;
      if (!inDirect) {
        src = inBlocks.data();
        if (tailBytes) {
          std::fill(inBlocks.begin(), inBlocks.begin() + static_cast<std::ptrdiff_t>(want), 0);  // zero-padded (:301-304)
        }
        if (!inRing.Read(inBlocks.data(), tailBytes ? tailBytes : want)) {
          return false;
        }
      }
      std::uint8_t *dst = outDirect ? wr[0].data : outBlocks.data();
      if (between) {
        between();
      }
      if (!process(src, dst, k)) {
        log("Filter output size mismatch");
        return false;
      }
      if (inDirect) {
        inRing.CommitRead(want);  // only now may the producer overwrite the blocks
      }
      ++st.processCalls;
      st.blocksProcessed += k;
      st.inPlaceCalls += (inDirect && outDirect) ? 1 : 0;
      if (outDirect) {
        outRing.CommitWrite(produce);
      } else if (!outRing.Write(outBlocks.data(), produce)) {
        log("Output buffer overflow; dropping accumulated audio");
        outRing.DiscardAll();
        ++st.outputOverflows;
        break;
      }
    }
    return true;
  };

  while (running()) {
    const long got = read(raw.data(), p.periodFrames);
    if (got < static_cast<long>(p.periodFrames)) {  // the reference stops here (ReadFull failed / EOF)
      if (got > 0 && filtered && p.drainAtEnd) {
        inputFrames += static_cast<std::size_t>(got);
        if (!inRing.Write(raw.data(), static_cast<std::size_t>(got) * frameBytes)) {
          log("Input buffer overflow; dropping accumulated audio");
          inRing.DiscardAll();
          ++st.inputOverflows;
        }
      } else if (got > 0 && !filtered && p.drainAtEnd) {
        RoundTrip(raw.data(), outChunk.data(), p.format, static_cast<std::size_t>(got) * p.channels, &scratch);
        if (write(outChunk.data(), static_cast<std::size_t>(got))) {
          st.framesWritten += static_cast<std::size_t>(got);
        }
      }
      break;
    }
    ++st.periodsRead;
    inputFrames += p.periodFrames;
    if (!filtered) {
      RoundTrip(raw.data(), outChunk.data(), p.format, p.periodFrames * p.channels, &scratch);
      if (!write(outChunk.data(), p.periodFrames)) {
        ok = false;
        break;
      }
      st.framesWritten += p.periodFrames;
      continue;
    }
    if (!inRing.Write(raw.data(), raw.size())) {
      log("Input buffer overflow; dropping accumulated audio");
      inRing.DiscardAll();
      ++st.inputOverflows;
    }
    if (!process_available(false)) {
      ok = false;
      break;
    }
    // drain in period*ratio chunks; keep the sink fed with silence when nothing is ready (:583-609)
    bool wrote = false;
    while (outRing.AvailableToRead() >= outChunk.size() && running()) {
      if (!outRing.Read(outChunk.data(), outChunk.size())) {
        log("Output buffer underrun");
        break;
      }
      if (!write(outChunk.data(), outPeriod)) {
        ok = false;
        break;
      }
      st.framesWritten += outPeriod;
      wrote = true;
    }
    if (!ok) {
      break;
    }
    if (!wrote && running()) {
      std::fill(outChunk.begin(), outChunk.end(), 0);  // a zero float buffer converts to zero PCM in every format
      if (!write(outChunk.data(), outPeriod)) {
        ok = false;
        break;
      }
      st.framesWritten += outPeriod;
      st.silenceFramesWritten += outPeriod;
    }
  }
  if (ok && filtered && p.drainAtEnd) {
    // everything that belongs to real input: inputFrames * ratio frames in total, minus what is already out
    const std::size_t realOut = inputFrames * factor;
    std::size_t already = st.framesWritten - st.silenceFramesWritten;
    for (;;) {
      if (!process_available(true)) {
        ok = false;
        break;
      }
      bool moved = false;
      while (outRing.AvailableToRead() > 0 && already < realOut) {
        const std::size_t frames = std::min({outRing.AvailableToRead() / frameBytes, outPeriod, realOut - already});
        if (frames == 0 || !outRing.Read(outChunk.data(), frames * frameBytes)) {
          break;
        }
        if (!write(outChunk.data(), frames)) {
          ok = false;
          break;
        }
        st.framesWritten += frames;
        already += frames;
        moved = true;
      }
      if (!ok || !moved || inRing.AvailableToRead() == 0) {
        break;
      }
    }
  }
  return ok;
}

}  // namespace miups
