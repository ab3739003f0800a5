// The capture -> ring -> batched engine -> ring -> playback loop of the streamer, on raw interleaved PCM.
//
// Reference: the ALSA streaming loop of src/alsa/alsa_streamer_main.cpp:428-611 -- one period is read, appended to
// the input staging, every complete filter block that is available AND fits the output staging is processed, then
// the output staging is drained in period*ratio chunks; when nothing is ready a period of silence is written so that
// the playback side never starves; a staging overflow drops what has accumulated (message + clear, :515-521,557-562).
// The reference keeps one float ring per channel and calls ProcessBlock per channel and block; here the rings hold
// interleaved PCM frames and one engine call takes all channels of up to `maxBlocksPerCall` blocks.
// Endpoints are callbacks (files, memory, ALSA behind HAVE_ALSA in the streamer), so the loop itself has no device
// dependency and is tested on the CPU with a stand-in processor.
#pragma once

#include <atomic>
#include <cstddef>
#include <functional>
#include <string>

namespace miups {

struct LoopParams {
  unsigned channels = 2;
  int format = 3;                   // MI_PCM_*
  std::size_t periodFrames = 1024;  // frames per source read
  std::size_t blockInFrames = 0;    // 0: no filter -> PCM -> float -> PCM pass-through per period, as the reference does
  std::size_t blockOutFrames = 0;
  std::size_t maxBlocksPerCall = 1;
  // additive (the reference stops at the first short read and leaves the staged tail unprocessed):
  // zero-pad the last partial block, process it, and write what remains -- truncated to inputFrames*ratio frames
  bool drainAtEnd = false;
  // optional: where the two staging rings live. With the engine behind `process`, pass mi_host_alloc / mi_host_free:
  // blocks are then handed to the engine IN PLACE in page-locked ring memory whenever a batch does not wrap
  // (LoopStats::inPlaceCalls), so they cross the host once per side and are DMA-able without a staging copy.
  void *(*hostAlloc)(std::size_t bytes) = nullptr;
  void (*hostFree)(void *p) = nullptr;
};

struct LoopStats {
  std::size_t periodsRead = 0, blocksProcessed = 0, framesWritten = 0, silenceFramesWritten = 0;
  std::size_t inputOverflows = 0, outputOverflows = 0, processCalls = 0;
  std::size_t inPlaceCalls = 0;  // process calls whose input AND output were contiguous pieces of the rings (no bounce copy)
};

// read: up to `frames` frames into dst, returns frames read (short = end of stream / stop)
using ReadFn = std::function<long(void *dst, std::size_t frames)>;
// write: exactly `frames` frames; false stops the loop
using WriteFn = std::function<bool(const void *src, std::size_t frames)>;
// process: `blocks` whole blocks, in -> out (interleaved PCM of the loop's format); false stops the loop
using ProcessFn = std::function<bool(const void *in, void *out, std::size_t blocks)>;
// called between engine calls (EQ reload, filter switch): never in the middle of a block
using BetweenBlocksFn = std::function<void()>;
using LogFn = std::function<void(const std::string &)>;
// polled between steps; false stops the loop (a signal handler flips the flag behind it)
using RunningFn = std::function<bool()>;

bool RunStreamLoop(const LoopParams &p, const ReadFn &read, const WriteFn &write, const ProcessFn &process,
                   const BetweenBlocksFn &between, const RunningFn &running, LoopStats *stats, const LogFn &log);

}  // namespace miups
