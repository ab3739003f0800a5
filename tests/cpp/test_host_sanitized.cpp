// Host side of the library under AddressSanitizer + UndefinedBehaviorSanitizer ("run host tests under
// ASan/UBSan"): csrc/host/*.cpp and csrc/capi_host.cpp compiled with -fsanitize=address,undefined and driven through the
// C ABI with a corpus the Python test writes -- the shipped sidecars, APO profiles, config.json and OPRA records, and
// hundreds of damaged versions of each (truncated at every offset, bytes flipped, brackets nested thousands deep).
// The readers take text from outside the process (files re-read on SIGHUP while audio is streaming): whatever they are
// fed they must return an error code, not read out of bounds, overflow the stack or leak.
//
// usage: test_host_sanitized <corpus dir>     files: config_*.txt opra_*.txt apo_*.txt sidecar_*.json (+ their .bin)
#include <dirent.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "mi_upsampler.h"

// the two HIP-side entry points capi_host.cpp refers to (pinned ring memory): plain heap here, so that the loop's in-place
// ring path runs under the sanitizers too
extern "C" void *mi_host_alloc(size_t bytes) { return std::malloc(bytes ? bytes : 1); }
extern "C" void mi_host_free(void *p) { std::free(p); }

namespace {

std::string Slurp(const std::string &path) {
  std::ifstream f(path, std::ios::binary);
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

struct LoopUser {
  std::vector<unsigned char> in, out;
  size_t pos = 0, frameBytes = 0, blockIn = 0, blockOut = 0;
};
long ReadCb(void *u, void *dst, size_t frames) {
  LoopUser *l = static_cast<LoopUser *>(u);
  const size_t want = frames * l->frameBytes, have = l->in.size() - l->pos, n = want < have ? want : have;
  std::memcpy(dst, l->in.data() + l->pos, n);
  l->pos += n;
  return static_cast<long>(n / l->frameBytes);
}
int WriteCb(void *u, const void *src, size_t frames) {
  LoopUser *l = static_cast<LoopUser *>(u);
  const unsigned char *p = static_cast<const unsigned char *>(src);
  l->out.insert(l->out.end(), p, p + frames * l->frameBytes);
  return 1;
}
int ProcessCb(void *u, const void *in, void *out, size_t blocks) {  // stand-in processor: repeat every frame ratio times
  LoopUser *l = static_cast<LoopUser *>(u);
  const size_t ratio = l->blockOut / l->blockIn;
  const unsigned char *s = static_cast<const unsigned char *>(in);
  unsigned char *d = static_cast<unsigned char *>(out);
  for (size_t f = 0; f < blocks * l->blockIn; ++f) {
    for (size_t r = 0; r < ratio; ++r) {
      std::memcpy(d + (f * ratio + r) * l->frameBytes, s + f * l->frameBytes, l->frameBytes);
    }
  }
  return 1;
}

}  // namespace

int main(int argc, char **argv) {
  if (argc != 2) {
    return 2;
  }
  const std::string dir = argv[1];
  DIR *d = opendir(dir.c_str());
  if (!d) {
    return 2;
  }
  size_t n = 0, accepted = 0;
  char err[1280], out[1 << 16];
  while (dirent *e = readdir(d)) {
    const std::string name = e->d_name, path = dir + "/" + name;
    if (name.rfind("config_", 0) == 0) {
      mi_runtime_config c;
      accepted += mi_parse_runtime_config(Slurp(path).c_str(), &c, err, sizeof(err)) == MI_OK;
    } else if (name.rfind("opra_", 0) == 0) {
      size_t needed = 0;
      for (int modern = 0; modern < 2; ++modern) {
        accepted += mi_opra_to_apo(Slurp(path).c_str(), modern, out, sizeof(out), &needed, err, sizeof(err)) == MI_OK;
      }
    } else if (name.rfind("apo_", 0) == 0) {
      const std::string text = Slurp(path);
      double pre = 0.0;
      std::vector<double> bands(9 * 64);
      accepted += mi_eq_parse(text.c_str(), &pre, bands.data(), 64) >= 0;
      std::vector<double> resp(2 * 257), mag(257);
      mi_eq_response_host(text.c_str(), 257, 512, 705600.0, resp.data());
      mi_eq_magnitude_host(text.c_str(), 257, 512, 705600.0, mag.data());
      // the EQ folded into a FIR (recursion over the taps, free decay followed to the work cap at most): whatever the
      // profile holds -- unstable sections, zero or absurd Q, frequencies beyond Nyquist -- it must come back
      std::vector<float> taps(300);
      for (size_t i = 0; i < taps.size(); ++i) {
        taps[i] = static_cast<float>((static_cast<int>(i * 37u % 101u) - 50)) * 0.01f;
      }
      std::vector<double> fir(taps.size());
      mi_eq_residual res;
      if (mi_eq_fold_host(taps.data(), taps.size(), 512, text.c_str(), 705600.0, fir.data(), &res) != MI_OK) {
        std::printf("mi_eq_fold_host refused %s\n", name.c_str());
        return 1;
      }
    } else if (name.rfind("sidecar_", 0) == 0 && name.size() > 5 && name.compare(name.size() - 5, 5, ".json") == 0) {
      mi_ups_config c;
      accepted += mi_read_filter(path.c_str(), &c, err, sizeof(err)) == MI_OK;
      mi_tables *t = nullptr;
      if (mi_tables_build(path.c_str(), MI_LOAD_DEFAULT, "Preamp: -3 dB\nFilter 1: ON PK Fc 1000 Hz Gain 2 dB Q 1\n", 96000.0, &t,
                          err, sizeof(err)) == MI_OK) {
        mi_tables_free(t);
      }
    } else {
      continue;
    }
    ++n;
  }
  closedir(d);
  // selector / negotiation / format helpers on odd arguments
  char pathOut[64];
  mi_resolve_filter_path("", dir.c_str(), "min", 4, 44100, pathOut, sizeof(pathOut), err, sizeof(err));  // tiny output buffer
  mi_resolve_filter_path("", "/nonexistent", "linear", 0, 0, out, sizeof(out), err, sizeof(err));
  mi_negotiated neg;
  const int rates[3] = {44100, 0, -5};
  mi_negotiate(-1, 1, 0, 0, rates, 3, 0, &neg);
  mi_negotiate(88200, 1, 44100, 768000, nullptr, 0, 352800, &neg);
  mi_parse_format(nullptr);
  mi_parse_format("s24_3le");
  // the streaming loop through pinned (here: heap) ring memory, in place where a batch does not wrap
  int rc = 0;
  for (unsigned period : {64u, 1000u, 4096u}) {
    LoopUser u;
    u.frameBytes = 6;  // stereo s24
    u.blockIn = 768;
    u.blockOut = 768 * 4;
    u.in.resize(u.frameBytes * (period * 37 + 11));
    for (size_t i = 0; i < u.in.size(); ++i) {
      u.in[i] = static_cast<unsigned char>(i * 131u >> 3);
    }
    mi_loop_params lp;
    std::memset(&lp, 0, sizeof(lp));
    lp.channels = 2;
    lp.format = MI_PCM_S24_3LE;
    lp.period_frames = period;
    lp.block_in_frames = u.blockIn;
    lp.block_out_frames = u.blockOut;
    lp.max_blocks_per_call = 3;
    lp.drain_at_end = 1;
    lp.pinned_rings = 1;
    mi_loop_stats st;
    volatile int running = 1;
    rc |= mi_stream_loop_run(&lp, ReadCb, WriteCb, ProcessCb, nullptr, nullptr, &u, &running, &st) != MI_OK;
    rc |= st.process_calls == 0;
  }
  std::printf("%zu inputs, %zu accepted, loop rc %d\n", n, accepted, rc);
  return rc;
}
