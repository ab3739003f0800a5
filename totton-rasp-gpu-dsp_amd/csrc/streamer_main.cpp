// alsa_streamer -- command-line front end of the MI355X upsampler.
//
// Keeps the reference streamer's flags, defaults, messages and exit codes
// (src/alsa/alsa_streamer_main.cpp:20-65 options, :67-196 parsing, :198-252
// filter preparation, :254-346 file pipeline, :350-427 main, :428-611 streaming
// loop) so scripts that drive the reference binary drive this one unchanged.
// The per-channel ProcessBlock loops of the reference become one batched engine
// call over all channels and up to --blocks-per-call blocks.
//
// Deviations, all documented in DESIGN.md:
//  * file mode writes framesRead * ratio frames per block (the reference's file
//    pipeline only works for ratio 1, :323-326,340-341);
//  * ALSA capture/playback is compiled only when HAVE_ALSA is defined (this
//    image has no alsa-lib headers); without it --in/--out report an error. The
//    streaming loop itself is always built and runs over file endpoints with --loop;
//  * a non-numeric value of a numeric flag is an error message + exit 1 (the
//    reference lets std::stoul's exception terminate the process);
//  * additive flags: --device, --gpus, --streams, --blocks-per-call, --eq, --eq-rate, --opra, --modern-target,
//    --config, --loop, --drain.
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <csignal>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>
#include <sstream>
#include <string>
#include <thread>
#include <chrono>
#include <vector>

#include "../../include/mi_upsampler.h"

#if defined(HAVE_ALSA)
#include <alsa/asoundlib.h>
#endif

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
  std::vector<int> gpus;        // --gpus a,b,...: stream s runs on gpus[s mod n]
  bool splitChannels = false;   // --split channels: contiguous channel groups of every stream per GPU instead
  bool splitTime = false;       // --split time: contiguous block ranges of every stream per GPU
  unsigned streams = 1;         // file mode: the input file holds this many equal-length streams back to back
  unsigned blocksPerCall = 16;
  std::string eqPath;
  std::string opraPath;       // one OPRA EQ record (JSON) converted to APO text at start-up
  bool modernTarget = false;  // ... with the KB5000_7 correction band
  double eqRate = 0.0;
  std::string configPath;       // config.json (eqEnabled / eqProfilePath), re-read on SIGHUP or when it changes
  bool loop = false;            // file endpoints through the streaming loop (period-sized reads)
  bool drain = false;           // --loop: process the zero-padded tail at end of input and flush
};

volatile int gRunning = 1;
volatile std::sig_atomic_t gReload = 0;
void OnSignal(int) { gRunning = 0; }
void OnHup(int) { gReload = 1; }

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
            << "  --gpus <a,b,...>        HIP devices to shard independent streams over (stream s -> gpus[s mod n])\n"
            << "  --split <streams|channels|time>  with --gpus: whole streams per GPU (default); the channels of every stream in\n"
            << "                          contiguous groups, one per GPU; or the blocks of every call in contiguous ranges, one\n"
            << "                          per GPU (one wide stream over several GPUs, contiguous copies)\n"
            << "  --streams <n>           File mode: the input holds n equal-length streams back to back (default: 1)\n"
            << "  --blocks-per-call <n>   Filter blocks batched per GPU call (default: 16)\n"
            << "  --eq <path>             Equalizer-APO profile folded into the filter\n"
            << "  --eq-rate <hz>          Output rate the EQ is evaluated at (default: rate*ratio)\n"
            << "  --opra <path>           OPRA EQ record (JSON), converted to an Equalizer-APO profile\n"
            << "  --modern-target         with --opra: add the KB5000_7 correction band\n"
            << "  --config <path>         config.json (eqEnabled, eqProfilePath); re-read on SIGHUP or when it changes\n"
            << "  --loop                  File mode: run the files through the streaming loop in --period reads\n"
            << "  --drain                 With --loop: process the zero-padded tail at end of input and flush\n"
            << "  --help                  Show this help\n";
}

bool ParseUnsigned(const std::string &flag, const std::string &s, unsigned *dst) {
  char *end = nullptr;
  const unsigned long v = s.empty() ? 0 : std::strtoul(s.c_str(), &end, 10);
  if (s.empty() || !end || *end != '\0' || s[0] == '-' || v > 0xfffffffful) {
    std::cerr << "Invalid value for " << flag << ": " << s << "\n";
    return false;
  }
  *dst = static_cast<unsigned>(v);
  return true;
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
      return value(&s) && ParseUnsigned(arg, s, dst);
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
    else if (arg == "--streams") ok = number(&o->streams);
    else if (arg == "--blocks-per-call") ok = number(&o->blocksPerCall);
    else if (arg == "--eq") ok = value(&o->eqPath);
    else if (arg == "--opra") ok = value(&o->opraPath);
    else if (arg == "--modern-target") o->modernTarget = true;
    else if (arg == "--config") ok = value(&o->configPath);
    else if (arg == "--loop") o->loop = true;
    else if (arg == "--drain") o->drain = true;
    else if (arg == "--eq-rate") {
      ok = value(&tmp);
      if (ok) {
        char *end = nullptr;
        o->eqRate = std::strtod(tmp.c_str(), &end);
        if (!end || *end != '\0' || !(o->eqRate > 0.0)) {
          std::cerr << "Invalid value for --eq-rate: " << tmp << "\n";
          ok = false;
        }
      }
    } else if (arg == "--split") {
      ok = value(&tmp);
      if (ok && tmp != "streams" && tmp != "channels" && tmp != "time") {
        std::cerr << "Invalid value for --split: " << tmp << "\n";
        ok = false;
      }
      o->splitChannels = ok && tmp == "channels";
      o->splitTime = ok && tmp == "time";
    } else if (arg == "--gpus") {
      ok = value(&tmp);
      std::stringstream ss(tmp);
      std::string item;
      while (ok && std::getline(ss, item, ',')) {
        unsigned d = 0;
        ok = ParseUnsigned("--gpus", item, &d);
        o->gpus.push_back(static_cast<int>(d));
      }
      if (ok && o->gpus.empty()) {
        std::cerr << "Invalid value for --gpus: " << tmp << "\n";
        ok = false;
      }
    } else {
      std::cerr << "Unknown argument: " << arg << "\n";
      return false;
    }
    if (!ok) {
      return false;
    }
  }
  return true;
}

bool ReadTextFile(const std::string &path, std::string *out) {
  std::ifstream f(path);
  if (!f) {
    return false;
  }
  out->assign((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  return true;
}

// One filter on one device (mi_filter + mi_engine), or the same filter on several devices with the streams sharded
// over them (mi_multi).
struct Pipeline {
  mi_filter *filter = nullptr;
  mi_engine *engine = nullptr;
  mi_multi *multi = nullptr;
  size_t inFrames = 0, outFrames = 0, factor = 1;
  unsigned streams = 1;
  double eqRate = 0.0;  // output rate the EQ cascade is evaluated at
  bool active() const { return engine || multi; }
  ~Pipeline() {
    if (multi) mi_multi_destroy(multi);
    if (engine) mi_engine_destroy(engine);
    if (filter) mi_filter_release(filter);
  }
  // An EQ whose ringing does not fit the filter still goes through (the cascade is cut to the filter's tap count); the
  // library says how much of the ideal response the cut dropped, and that goes to stderr (include/mi_upsampler.h).
  bool SetEq(const std::string &text) {
    if ((multi ? mi_multi_set_eq(multi, text.c_str(), eqRate) : mi_filter_set_eq(filter, text.c_str(), eqRate)) != MI_OK) {
      return false;
    }
    const char *warning = mi_ups_last_error();
    if (warning && warning[0]) {
      std::cerr << "EQ warning: " << warning << "\n";
    }
    return true;
  }
  // `blocks` blocks of every stream; streams are `inStride` / `outStride` bytes apart in the host buffers
  bool Process(const void *in, size_t inStride, void *out, size_t outStride, size_t blocks) {
    return (multi ? mi_multi_process_host(multi, in, inStride, out, outStride, blocks)
                  : mi_engine_process_host(engine, in, inStride, out, outStride, blocks)) == MI_OK;
  }
};

// PrepareFilter (alsa_streamer_main.cpp:198-252): false = fatal, true with
// !p->active() = run without filter.
bool PrepareFilter(const CliOptions &o, int fmt, Pipeline *p) {
  const bool required = !o.filterPath.empty();
  p->streams = std::max(1u, o.streams);
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
  mi_ups_config c;
  if (!o.gpus.empty()) {
    if (mi_multi_create(path,
                        MI_LOAD_DEFAULT | (o.splitChannels ? MI_MULTI_SPLIT_CHANNELS : 0) | (o.splitTime ? MI_MULTI_SPLIT_TIME : 0),
                        o.gpus.data(), o.gpus.size(),
                        static_cast<int>(p->streams),
                        static_cast<int>(o.channels), fmt, fmt, &p->multi, err, sizeof(err)) != MI_OK) {
      std::cerr << "Filter load failed: " << err << "\n";
      std::cerr << "Filter path: " << path << "\n";
      return false;
    }
    if (mi_read_filter(path, &c, err, sizeof(err)) != MI_OK) {
      std::cerr << "Filter load failed: " << err << "\n";
      return false;
    }
    p->inFrames = mi_multi_in_frames_per_block(p->multi);
    p->outFrames = mi_multi_out_frames_per_block(p->multi);
    if (o.splitChannels && o.gpus.size() > 1) {
      // A channel group is a COLUMN of the caller's frames: it crosses the host link by pitched DMA, which moves about
      // 180 M rows/s whatever the row width (profiles/r03_h_multi_split.txt: 64 / 32 / 16-byte rows = 11.6 / 5.7 / 2.9 GB/s
      // against 54 GB/s for contiguous copies). Say so where the groups are narrow, and name the partition that is not.
      std::vector<int> first(o.gpus.size() + 1, 0);
      mi_multi_partition_channels(static_cast<int>(o.channels), static_cast<int>(o.gpus.size()), first.data());
      size_t narrow = SIZE_MAX;
      for (size_t i = 0; i < o.gpus.size(); ++i) {
        if (first[i + 1] > first[i]) {
          narrow = std::min(narrow, static_cast<size_t>(first[i + 1] - first[i]) * mi_bytes_per_sample(fmt));
        }
      }
      if (narrow != SIZE_MAX && narrow < 64) {
        std::cerr << "note: --split channels gives a GPU rows of " << narrow << " bytes; pitched DMA moves about 180 M rows/s, i.e. "
                  << (static_cast<double>(narrow) * 0.18) << " GB/s per direction on this link against ~54 GB/s for contiguous copies. "
                  << "For one host-fed stream --split time (contiguous block ranges per GPU) keeps the full link rate; "
                  << "--split channels pays off for groups of 64 bytes or more, or for data that is already on the device.\n";
      }
    }
  } else {
    if (mi_filter_load(o.device, path, MI_LOAD_DEFAULT, &p->filter, err, sizeof(err)) != MI_OK) {
      std::cerr << "Filter load failed: " << err << "\n";
      std::cerr << "Filter path: " << path << "\n";
      return false;
    }
    mi_filter_get_config(p->filter, &c);
  }
  p->factor = std::max<size_t>(c.upsample_factor, 1);
  p->eqRate = o.eqRate > 0.0 ? o.eqRate : static_cast<double>(o.requestedRate) * static_cast<double>(p->factor);
  if (!p->multi) {
    if (mi_engine_create(p->filter, static_cast<int>(p->streams), static_cast<int>(o.channels), fmt, fmt, &p->engine) !=
        MI_OK) {
      std::cerr << "Filter load failed: " << mi_ups_last_error() << "\n";
      return false;
    }
    p->inFrames = mi_engine_in_frames_per_block(p->engine);
    p->outFrames = mi_engine_out_frames_per_block(p->engine);
  }
  if (!o.eqPath.empty()) {
    std::string text;
    if (!ReadTextFile(o.eqPath, &text)) {
      std::cerr << "EQ Parser: Cannot open file: " << o.eqPath << "\n";
      return false;
    }
    if (!p->SetEq(text)) {
      std::cerr << "EQ load failed: " << mi_ups_last_error() << "\n";
      return false;
    }
  }
  if (!o.opraPath.empty()) {  // reference: web/routers/opra.py:140-160 (record -> APO text -> the EQ path)
    std::string json;
    if (!ReadTextFile(o.opraPath, &json)) {
      std::cerr << "OPRA: Cannot open file: " << o.opraPath << "\n";
      return false;
    }
    std::vector<char> apo(16384);
    char err[512];
    size_t need = 0;
    int rc = mi_opra_to_apo(json.c_str(), o.modernTarget ? 1 : 0, apo.data(), apo.size(), &need, err, sizeof err);
    if (rc == MI_ERR_ARG && need > apo.size()) {
      apo.resize(need);
      rc = mi_opra_to_apo(json.c_str(), o.modernTarget ? 1 : 0, apo.data(), apo.size(), &need, err, sizeof err);
    }
    if (rc != MI_OK) {
      std::cerr << "OPRA: " << err << "\n";
      return false;
    }
    if (!p->SetEq(apo.data())) {
      std::cerr << "EQ load failed: " << mi_ups_last_error() << "\n";
      return false;
    }
  }
  return true;
}

// EQ activation (reference control plane: web/routers/eq.py:220-273 writes config.json and sends RELOAD; the reference
// daemon only counts the command, src/zmq/zmq_server_main.cpp:168-172). Here: SIGHUP or a changed config.json makes
// the streamer re-read eqEnabled / eqProfilePath between two blocks and swap the filter tables without a stall.
struct EqActivation {
  std::string configPath;
  long long mtimeNs = -1, size = -1;
  bool Stat(long long *m, long long *s) const {
    struct stat st;
    if (configPath.empty() || ::stat(configPath.c_str(), &st) != 0) {
      return false;
    }
    *m = static_cast<long long>(st.st_mtim.tv_sec) * 1000000000ll + st.st_mtim.tv_nsec;
    *s = static_cast<long long>(st.st_size);
    return true;
  }
  bool Apply(Pipeline *p) {
    if (configPath.empty() || !p->active()) {
      return true;
    }
    (void)Stat(&mtimeNs, &size);
    std::string text;
    if (!ReadTextFile(configPath, &text)) {
      std::cerr << "Config reload failed: cannot open " << configPath << "\n";
      return false;
    }
    mi_runtime_config c;
    char err[512];
    if (mi_parse_runtime_config(text.c_str(), &c, err, sizeof(err)) != MI_OK) {
      std::cerr << "Config reload failed: " << err << "\n";
      return false;
    }
    std::string apo;
    if (c.eq_enabled && c.eq_profile_path[0]) {
      if (!ReadTextFile(c.eq_profile_path, &apo)) {
        std::cerr << "EQ Parser: Cannot open file: " << c.eq_profile_path << "\n";
        return false;
      }
    }
    if (!p->SetEq(apo)) {  // on failure the previous spectrum stays active
      std::cerr << "EQ reload failed: " << mi_ups_last_error() << "\n";
      return false;
    }
    if (apo.empty()) {
      std::cerr << "EQ disabled\n";
    } else {
      std::cerr << "EQ reloaded: " << (c.eq_profile[0] ? c.eq_profile : c.eq_profile_path) << "\n";
    }
    return true;
  }
  // between two blocks
  void Poll(Pipeline *p) {
    if (configPath.empty()) {
      return;
    }
    bool due = gReload != 0;
    gReload = 0;
    long long m = 0, s = 0;
    if (!due && Stat(&m, &s) && (m != mtimeNs || s != size)) {
      due = true;
    }
    if (due) {
      (void)Apply(p);
    }
  }
};

// ProcessFilePipeline (alsa_streamer_main.cpp:254-346), batched. With --streams n the file is n equal-length streams
// laid out one after the other; every call processes the same block range of all of them.
bool ProcessFile(const CliOptions &o, int fmt, Pipeline *p, unsigned periodFrames, EqActivation *eq) {
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
  const size_t streams = p->streams;
  size_t streamBytes = 0;  // bytes of one stream in the input file (streams > 1 only)
  if (streams > 1) {
    input.seekg(0, std::ios::end);
    const size_t total = static_cast<size_t>(input.tellg());
    input.seekg(0, std::ios::beg);
    if (total % (streams * frameBytes) != 0) {
      std::cerr << "Input file does not hold " << streams << " equal-length streams of whole frames\n";
      return false;
    }
    streamBytes = total / streams;
    if (!p->active()) {
      std::cerr << "--streams needs a filter\n";
      return false;
    }
  }
  const size_t blocksPerCall = p->active() ? std::max(1u, o.blocksPerCall) : 1;
  const size_t callFrames = static_cast<size_t>(periodFrames) * blocksPerCall;
  // pinned staging (DMA without a bounce copy) when a device path exists
  const size_t inBytes = callFrames * frameBytes * streams, outBytes = callFrames * p->factor * frameBytes * streams;
  std::vector<uint8_t> rawVec, outVec;
  uint8_t *raw = nullptr, *out = nullptr;
  if (p->active()) {
    raw = static_cast<uint8_t *>(mi_host_alloc(inBytes));
    out = static_cast<uint8_t *>(mi_host_alloc(outBytes));
  }
  const bool pinned = raw && out;
  if (!pinned) {
    mi_host_free(raw);
    mi_host_free(out);
    rawVec.resize(inBytes);
    outVec.resize(outBytes);
    raw = rawVec.data();
    out = outVec.data();
  }
  std::vector<float> scratch;
  bool ok = true;
  size_t done = 0;  // frames of every stream already processed

  std::cerr << "File processing started: input " << o.requestedRate << " Hz, period " << periodFrames << " frames\n";
  while (gRunning) {
    size_t framesRead = 0;
    if (streams == 1) {
      input.read(reinterpret_cast<char *>(raw), static_cast<std::streamsize>(callFrames * frameBytes));
      framesRead = static_cast<size_t>(std::max<std::streamsize>(input.gcount(), 0)) / frameBytes;
    } else {
      const size_t left = streamBytes / frameBytes - done;
      framesRead = std::min(left, callFrames);
      for (size_t s = 0; s < streams && framesRead; ++s) {
        input.clear();
        input.seekg(static_cast<std::streamoff>(s * streamBytes + done * frameBytes));
        input.read(reinterpret_cast<char *>(raw + s * callFrames * frameBytes),
                   static_cast<std::streamsize>(framesRead * frameBytes));
      }
    }
    if (framesRead == 0) {
      break;
    }
    // a short tail is zero-padded up to whole blocks (:301-304)
    const size_t blocks = (framesRead + periodFrames - 1) / periodFrames;
    for (size_t s = 0; s < streams; ++s) {
      uint8_t *row = raw + s * callFrames * frameBytes;
      std::fill(row + framesRead * frameBytes, row + blocks * periodFrames * frameBytes, 0);
    }
    if (p->active()) {
      eq->Poll(p);
      if (!p->Process(raw, callFrames * frameBytes, out, callFrames * p->factor * frameBytes, blocks)) {
        std::cerr << "Filter processing failed: " << mi_ups_last_error() << "\n";
        ok = false;
        break;
      }
    } else {
      // no filter: PCM -> float -> PCM exactly as the reference does
      const size_t n = blocks * periodFrames * o.channels;
      scratch.resize(n);
      if (mi_pcm_to_float(raw, fmt, n, scratch.data()) != MI_OK || mi_float_to_pcm(scratch.data(), n, fmt, out) != MI_OK) {
        std::cerr << "PCM conversion failed\n";
        ok = false;
        break;
      }
    }
    for (size_t s = 0; s < streams; ++s) {
      if (streams > 1) {
        output.seekp(static_cast<std::streamoff>((s * streamBytes + done * frameBytes) * p->factor));
      }
      output.write(reinterpret_cast<const char *>(out + s * callFrames * p->factor * frameBytes),
                   static_cast<std::streamsize>(framesRead * p->factor * frameBytes));
    }
    done += framesRead;
  }
  if (pinned) {
    mi_host_free(raw);
    mi_host_free(out);
  }
  if (ok) {
    std::cerr << "File processing stopped\n";
  }
  return ok;
}

// ---- streaming loop endpoints ---------------------------------------------------------------------------------
struct LoopContext {
  Pipeline *pipe = nullptr;
  EqActivation *eq = nullptr;
  size_t frameBytes = 0;
  std::ifstream *in = nullptr;
  std::ofstream *out = nullptr;
#if defined(HAVE_ALSA)
  snd_pcm_t *capture = nullptr, *playback = nullptr;
#endif
};

long FileRead(void *user, void *dst, size_t frames) {
  auto *c = static_cast<LoopContext *>(user);
  c->in->read(static_cast<char *>(dst), static_cast<std::streamsize>(frames * c->frameBytes));
  return static_cast<long>(static_cast<size_t>(std::max<std::streamsize>(c->in->gcount(), 0)) / c->frameBytes);
}
int FileWrite(void *user, const void *src, size_t frames) {
  auto *c = static_cast<LoopContext *>(user);
  c->out->write(static_cast<const char *>(src), static_cast<std::streamsize>(frames * c->frameBytes));
  return c->out->good() ? 1 : 0;
}
int EngineProcess(void *user, const void *in, void *out, size_t blocks) {
  auto *c = static_cast<LoopContext *>(user);
  return c->pipe->Process(in, 0, out, 0, blocks) ? 1 : 0;
}
void BetweenBlocks(void *user) {
  auto *c = static_cast<LoopContext *>(user);
  c->eq->Poll(c->pipe);
}
void LogLine(void *, const char *m) { std::cerr << m << "\n"; }

// Built-in "null" endpoints (--in null --out null), the scenario of the reference's tests/cpp/test_alsa_streamer_e2e.cpp
// without an ALSA library: capture delivers silence at the pace of the sample rate, playback discards.
struct NullPacer {
  std::chrono::steady_clock::time_point start = std::chrono::steady_clock::now();
  unsigned long long frames = 0;
  unsigned rate = 44100;
};
NullPacer gNull;
long NullRead(void *user, void *dst, size_t frames) {
  auto *c = static_cast<LoopContext *>(user);
  std::memset(dst, 0, frames * c->frameBytes);
  gNull.frames += frames;
  const auto due = gNull.start + std::chrono::nanoseconds(gNull.frames * 1000000000ull / gNull.rate);
  while (gRunning && std::chrono::steady_clock::now() < due) {
    std::this_thread::sleep_for(std::min<std::chrono::nanoseconds>(
        std::chrono::milliseconds(5), std::chrono::duration_cast<std::chrono::nanoseconds>(due - std::chrono::steady_clock::now())));
  }
  return gRunning ? static_cast<long>(frames) : 0;
}
int NullWrite(void *, const void *, size_t) { return 1; }

#if defined(HAVE_ALSA)
// XRUN policy of the reference (alsa_common.cpp:269-336): snd_pcm_recover on -EPIPE / -ESTRPIPE / -EINTR, retry.
long AlsaRead(void *user, void *dst, size_t frames) {
  auto *c = static_cast<LoopContext *>(user);
  size_t done = 0;
  while (done < frames && gRunning) {
    const snd_pcm_sframes_t n =
        snd_pcm_readi(c->capture, static_cast<char *>(dst) + done * c->frameBytes, frames - done);
    if (n == -EAGAIN) {
      snd_pcm_wait(c->capture, 100);
      continue;
    }
    if (n < 0) {
      if (snd_pcm_recover(c->capture, static_cast<int>(n), 1) < 0) {
        std::cerr << "ALSA read failed: " << snd_strerror(static_cast<int>(n)) << "\n";
        return static_cast<long>(done);
      }
      continue;
    }
    done += static_cast<size_t>(n);
  }
  return static_cast<long>(done);
}
int AlsaWrite(void *user, const void *src, size_t frames) {
  auto *c = static_cast<LoopContext *>(user);
  size_t done = 0;
  while (done < frames && gRunning) {
    const snd_pcm_sframes_t n =
        snd_pcm_writei(c->playback, static_cast<const char *>(src) + done * c->frameBytes, frames - done);
    if (n == -EAGAIN) {
      snd_pcm_wait(c->playback, 100);
      continue;
    }
    if (n < 0) {
      if (snd_pcm_recover(c->playback, static_cast<int>(n), 1) < 0) {
        std::cerr << "ALSA write failed: " << snd_strerror(static_cast<int>(n)) << "\n";
        return 0;
      }
      continue;
    }
    done += static_cast<size_t>(n);
  }
  return done == frames ? 1 : 0;
}
snd_pcm_t *OpenAlsa(const std::string &name, snd_pcm_stream_t dir, int fmt, unsigned channels, unsigned rate,
                    unsigned period, unsigned buffer) {
  snd_pcm_t *h = nullptr;
  if (snd_pcm_open(&h, name.c_str(), dir, 0) < 0) {
    std::cerr << "Failed to open ALSA device: " << name << "\n";
    return nullptr;
  }
  const snd_pcm_format_t f = fmt == MI_PCM_S16 ? SND_PCM_FORMAT_S16_LE
                                               : (fmt == MI_PCM_S24_3LE ? SND_PCM_FORMAT_S24_3LE : SND_PCM_FORMAT_S32_LE);
  const unsigned frames = buffer ? buffer : period * 4;
  const unsigned latencyUs = static_cast<unsigned>(1000000.0 * frames / std::max(rate, 1u));
  if (snd_pcm_set_params(h, f, SND_PCM_ACCESS_RW_INTERLEAVED, channels, rate, 0, latencyUs) < 0) {
    std::cerr << "Failed to configure ALSA device: " << name << "\n";
    snd_pcm_close(h);
    return nullptr;
  }
  return h;
}
#endif

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
  if (o.streams == 0) {
    std::cerr << "Unsupported stream count: 0\n";
    return 1;
  }
  std::signal(SIGINT, OnSignal);
  std::signal(SIGTERM, OnSignal);
  std::signal(SIGHUP, OnHup);

  const bool nullMode = !fileMode && o.inputDevice == "null" && o.outputDevice == "null";
  (void)nullMode;
#if !defined(HAVE_ALSA)
  if (!fileMode && !nullMode) {
    std::cerr << "ALSA support is not compiled into this build (no alsa-lib headers); "
                 "use --in-file/--out-file (or --in null --out null)\n";
    return 1;
  }
#endif

  Pipeline p;
  if (!PrepareFilter(o, fmt, &p)) {
    return 1;
  }
  EqActivation eq;
  eq.configPath = o.configPath;
  if (!o.configPath.empty() && p.active() && !eq.Apply(&p)) {
    return 1;
  }
  size_t blockInputFrames = 0;
  if (p.active()) {
    blockInputFrames = p.inFrames;
    if (blockInputFrames == 0) {
      std::cerr << "Invalid filter block size for input buffering.\n";
      return 1;
    }
  }
  // period sizing (alsa_streamer_main.cpp:404-418)
  const bool pipelineMode = fileMode && !o.loop;
  unsigned periodFrames = o.periodFrames;
  if (pipelineMode && blockInputFrames > 0) {
    periodFrames = static_cast<unsigned>(blockInputFrames);
  } else if (periodFrames == 0) {
    periodFrames = 1024;
    if (blockInputFrames > 0) {
      periodFrames = static_cast<unsigned>(std::min<size_t>(periodFrames, blockInputFrames));
    }
  } else if (!pipelineMode && blockInputFrames > 0 && periodFrames > blockInputFrames) {
    std::cerr << "ALSA period is larger than filter input block; clamping to " << blockInputFrames << " frames\n";
    periodFrames = static_cast<unsigned>(blockInputFrames);
  }
  if (pipelineMode) {
    return ProcessFile(o, fmt, &p, periodFrames, &eq) ? 0 : 1;
  }

  // ---- streaming loop (alsa_streamer_main.cpp:428-611) over file or ALSA endpoints ----
  if (p.streams > 1) {
    std::cerr << "The streaming loop runs one stream; use the file pipeline for --streams\n";
    return 1;
  }
  LoopContext ctx;
  ctx.pipe = &p;
  ctx.eq = &eq;
  ctx.frameBytes = mi_bytes_per_sample(fmt) * o.channels;
  std::ifstream fin;
  std::ofstream fout;
  mi_read_fn readFn = nullptr;
  mi_write_fn writeFn = nullptr;
  unsigned inputRate = o.requestedRate;
  if (fileMode) {
    if (o.requestedRate == 0) {
      std::cerr << "--rate is required for file processing\n";
      return 1;
    }
    fin.open(o.inputFile, std::ios::binary);
    if (!fin) {
      std::cerr << "Failed to open input file: " << o.inputFile << "\n";
      return 1;
    }
    fout.open(o.outputFile, std::ios::binary | std::ios::trunc);
    if (!fout) {
      std::cerr << "Failed to open output file: " << o.outputFile << "\n";
      return 1;
    }
    ctx.in = &fin;
    ctx.out = &fout;
    readFn = FileRead;
    writeFn = FileWrite;
  }
#if !defined(HAVE_ALSA)
  else {  // nullMode (checked above)
    if (inputRate == 0) {
      inputRate = 44100;
    }
    gNull = NullPacer();
    gNull.rate = inputRate;
    readFn = NullRead;
    writeFn = NullWrite;
  }
#endif
#if defined(HAVE_ALSA)
  else {
    if (inputRate == 0) {
      inputRate = 44100;
    }
    ctx.capture = OpenAlsa(o.inputDevice, SND_PCM_STREAM_CAPTURE, fmt, o.channels, inputRate, periodFrames, o.bufferFrames);
    if (!ctx.capture) {
      return 1;
    }
    ctx.playback = OpenAlsa(o.outputDevice, SND_PCM_STREAM_PLAYBACK, fmt, o.channels,
                            static_cast<unsigned>(inputRate * p.factor), static_cast<unsigned>(periodFrames * p.factor),
                            static_cast<unsigned>(o.bufferFrames * p.factor));
    if (!ctx.playback) {
      snd_pcm_close(ctx.capture);
      return 1;
    }
    readFn = AlsaRead;
    writeFn = AlsaWrite;
  }
#endif
  mi_loop_params lp;
  std::memset(&lp, 0, sizeof(lp));
  lp.channels = o.channels;
  lp.format = fmt;
  lp.period_frames = periodFrames;
  lp.block_in_frames = p.active() ? p.inFrames : 0;
  lp.block_out_frames = p.active() ? p.outFrames : 0;
  lp.max_blocks_per_call = std::max(1u, o.blocksPerCall);
  lp.drain_at_end = o.drain ? 1 : 0;
  lp.pinned_rings = p.active() ? 1 : 0;  // the engine reads its blocks out of, and writes them into, page-locked ring memory
  mi_loop_stats st;
  std::cerr << (fileMode ? "File streaming started: input " : "ALSA streaming started: input ") << inputRate << " Hz, "
            << "output " << static_cast<unsigned long long>(inputRate) * p.factor << " Hz, "
            << "period " << periodFrames << " frames\n";
  const int rc = mi_stream_loop_run(&lp, readFn, writeFn, p.active() ? EngineProcess : nullptr,
                                    p.active() ? BetweenBlocks : nullptr, LogLine, &ctx, &gRunning, &st);
#if defined(HAVE_ALSA)
  if (ctx.capture) {
    snd_pcm_drop(ctx.capture);
    snd_pcm_close(ctx.capture);
  }
  if (ctx.playback) {
    snd_pcm_drain(ctx.playback);
    snd_pcm_close(ctx.playback);
  }
#endif
  std::cerr << (fileMode ? "File streaming stopped" : "ALSA streaming stopped") << ": " << st.periods_read << " periods, "
            << st.blocks_processed << " blocks, " << st.frames_written << " frames out (" << st.silence_frames_written
            << " of silence), overflows " << st.input_overflows << "/" << st.output_overflows << "\n";
  return rc == MI_OK ? 0 : 1;
}
