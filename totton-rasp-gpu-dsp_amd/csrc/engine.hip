// Engine implementation + kernel launches (hipcc, gfx950).
#include "engine.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <sstream>

#include "device/kernel_fused.h"
#include "device/kernels_generic.h"
#include "host/eq.h"

namespace miups {
namespace {

thread_local std::string g_lastError;

bool HipOk(hipError_t e, const char *what, std::string *error) {
  if (e == hipSuccess) {
    return true;
  }
  std::ostringstream os;
  os << what << ": " << hipGetErrorString(e);
  if (error) {
    *error = os.str();
  }
  return false;
}

#define MI_HIP(call)                       \
  do {                                     \
    if (!HipOk((call), #call, error)) {    \
      return false;                        \
    }                                      \
  } while (0)

bool UseDevice(int device, std::string *error) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    if (error) {
      *error = "no HIP device available (this library has no CPU path)";
    }
    return false;
  }
  if (device < 0 || device >= n) {
    if (error) {
      *error = "device index out of range";
    }
    return false;
  }
  return HipOk(hipSetDevice(device), "hipSetDevice", error);
}

template <typename T>
bool Upload(const std::vector<T> &src, T **dst, std::string *error) {
  MI_HIP(hipMalloc(reinterpret_cast<void **>(dst), std::max<std::size_t>(src.size(), 1) * sizeof(T)));
  if (!src.empty()) {
    MI_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  }
  return true;
}

unsigned Blocks(long long total, int threads) { return static_cast<unsigned>((total + threads - 1) / threads); }

// ---- EQ cascade on the device (fp64) ---------------------------------------
// H(f_i) = preamp * prod_s (b0 + b1 z + b2 z^2) / (1 + a1 z + a2 z^2), z = e^{-j 2 pi f_i / fs}
// (reference: biquadFrequencyResponse + computeEqFrequencyResponse, eq_to_fir.cpp:77-130)
__global__ void eq_response_kernel(double preamp, int apply_preamp, const eq::BiquadCoeffs *sections, int nsec,
                                   double fs, double df, std::size_t nbins, double2 *out) {
  const std::size_t i = static_cast<std::size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= nbins) {
    return;
  }
  const double pi = 3.14159265358979323846;
  const double w = 2.0 * pi * (static_cast<double>(i) * df) / fs;
  double sn, cs;
  sincos(w, &sn, &cs);
  const double zr = cs, zi = -sn;
  const double z2r = zr * zr - zi * zi, z2i = 2.0 * zr * zi;
  double rr = 1.0, ri = 0.0;
  if (apply_preamp) {
    rr *= preamp;
  }
  for (int s = 0; s < nsec; ++s) {
    const eq::BiquadCoeffs c = sections[s];
    const double nr = c.b0 + c.b1 * zr + c.b2 * z2r, ni = c.b1 * zi + c.b2 * z2i;
    const double dr = 1.0 + c.a1 * zr + c.a2 * z2r, di = c.a1 * zi + c.a2 * z2i;
    const double inv = 1.0 / (dr * dr + di * di);
    const double qr = (nr * dr + ni * di) * inv, qi = (ni * dr - nr * di) * inv;
    const double tr = rr * qr - ri * qi;
    ri = rr * qi + ri * qr;
    rr = tr;
  }
  out[i] = make_double2(rr, ri);
}

// ---- staged-path launch helpers ---------------------------------------------
template <int DIR>
void LaunchPass(int R, const cf *in, cf *out, const cf *tw, int K, int Ns, int log2NsR, long long rows,
                hipStream_t st) {
  const int threads = 256;
  const unsigned grid = Blocks(rows * (K / R), threads);
  switch (R) {
    case 2:
      hipLaunchKernelGGL((gen_pass_kernel<DIR, 2>), dim3(grid), dim3(threads), 0, st, in, out, tw, K, Ns, log2NsR, rows);
      break;
    case 4:
      hipLaunchKernelGGL((gen_pass_kernel<DIR, 4>), dim3(grid), dim3(threads), 0, st, in, out, tw, K, Ns, log2NsR, rows);
      break;
    case 8:
      hipLaunchKernelGGL((gen_pass_kernel<DIR, 8>), dim3(grid), dim3(threads), 0, st, in, out, tw, K, Ns, log2NsR, rows);
      break;
    default:
      hipLaunchKernelGGL((gen_pass_kernel<DIR, 16>), dim3(grid), dim3(threads), 0, st, in, out, tw, K, Ns, log2NsR,
                         rows);
      break;
  }
}

// length-K batched transform, ping-ponging between a and b; returns where the result is
template <int DIR>
cf *StagedFft(cf *a, cf *b, const cf *tw, int log2k, long long rows, hipStream_t st) {
  const int K = 1 << log2k;
  int done = 0;  // log2(Ns)
  cf *src = a, *dst = b;
  auto pass = [&](int log2r) {
    LaunchPass<DIR>(1 << log2r, src, dst, tw, K, 1 << done, done + log2r, rows, st);
    done += log2r;
    std::swap(src, dst);
  };
  if (log2k % 4) {
    pass(log2k % 4);
  }
  while (done < log2k) {
    pass(4);
  }
  return src;
}

template <int LOG2K, bool EXT>
bool LaunchFusedVariant(const Geometry &g, const IoDesc &io, const DeviceFilter &f, unsigned items, hipStream_t st,
                        std::string *error) {
  using Cfg = FusedCfg<LOG2K>;
  static bool attr_set[64] = {};
  int dev = 0;
  MI_HIP(hipGetDevice(&dev));
  if (Cfg::LDS_BYTES > 64 * 1024 && dev < 64 && !attr_set[dev]) {
    MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fused_kernel<LOG2K, EXT>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL((fused_kernel<LOG2K, EXT>), dim3(items), dim3(Cfg::T), Cfg::LDS_BYTES, st, g, io, f.fused());
  return HipOk(hipGetLastError(), "fused_kernel launch", error);
}

template <int LOG2K>
bool LaunchFused(const Geometry &g, const IoDesc &io, const DeviceFilter &f, unsigned items, hipStream_t st,
                 std::string *error) {
  return io.ext_epilogue ? LaunchFusedVariant<LOG2K, true>(g, io, f, items, st, error)
                         : LaunchFusedVariant<LOG2K, false>(g, io, f, items, st, error);
}

template <int LOG2K>
bool LaunchFusedSplit(const Geometry &g, const IoDesc &io, const DeviceFilter &f, unsigned items, hipStream_t st,
                      std::string *error) {
  using Cfg = FusedCfg<LOG2K>;
  static bool attr_set[64] = {};
  int dev = 0;
  MI_HIP(hipGetDevice(&dev));
  if (Cfg::LDS_BYTES_SPLIT > 64 * 1024 && dev < 64 && !attr_set[dev]) {
    MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fused_split_kernel<LOG2K>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES_SPLIT));
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL((fused_split_kernel<LOG2K>), dim3(items), dim3(Cfg::T), Cfg::LDS_BYTES_SPLIT, st, g, io,
                     f.fused());
  return HipOk(hipGetLastError(), "fused_split_kernel launch", error);
}

bool DispatchFused(const Geometry &g, const IoDesc &io, const DeviceFilter &f, unsigned items, hipStream_t st,
                   std::string *error) {
  if (f.fusedSplit()) {
    // block transform length K = 32768: two 16384-point transforms through the LDS
    // (the split layout is only built for that size outside the emulation tests)
    if (g.log2k == 15) {
      return LaunchFusedSplit<14>(g, io, f, items, st, error);
    }
    if (g.log2k == 14) {  // experiment switch MIUPS_EXP_FORCE_SPLIT (profiles/): two 64 KiB workgroups per CU
      return LaunchFusedSplit<13>(g, io, f, items, st, error);
    }
    if (error) {
      *error = "split fused kernel does not cover this geometry";
    }
    return false;
  }
  switch (g.log2k) {
    case 5: return LaunchFused<5>(g, io, f, items, st, error);
    case 6: return LaunchFused<6>(g, io, f, items, st, error);
    case 7: return LaunchFused<7>(g, io, f, items, st, error);
    case 8: return LaunchFused<8>(g, io, f, items, st, error);
    case 9: return LaunchFused<9>(g, io, f, items, st, error);
    case 10: return LaunchFused<10>(g, io, f, items, st, error);
    case 11: return LaunchFused<11>(g, io, f, items, st, error);
    case 12: return LaunchFused<12>(g, io, f, items, st, error);
    case 13: return LaunchFused<13>(g, io, f, items, st, error);
    case 14: return LaunchFused<14>(g, io, f, items, st, error);
    default:
      if (error) {
        *error = "fused kernel does not cover this geometry";
      }
      return false;
  }
}

// The fused kernel addresses samples with 32-bit byte offsets from per-block bases.
bool FusedCovers(const Geometry &g, int channels, int inFmt, int outFmt) {
  const long long inSpan = static_cast<long long>(g.M) * channels * pcm_bytes(inFmt);
  const long long outSpan = static_cast<long long>(g.M) * g.P * channels * pcm_bytes(outFmt);
  return g.S == 1 && g.log2k >= 5 && g.log2k <= 15 && inSpan < (1ll << 31) && outSpan < (1ll << 31);
}

// planarize_kernel keeps a [channels][kPlanarTile + 1] fp32 tile in LDS (64 KiB default limit)
constexpr int kMaxPlanarChannels = 240;

}  // namespace

void SetLastError(const std::string &message) { g_lastError = message; }
const std::string &LastError() { return g_lastError; }

int DeviceCount() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    return 0;
  }
  return n;
}

// ------------------------------------------------------------ DeviceFilter --
DeviceFilter::~DeviceFilter() { Free(); }

void DeviceFilter::Free() {
  if (dGs_ || dGc_ || dWm_ || dtw_) {
    (void)hipSetDevice(device_);
  }
  (void)hipFree(dGs_);
  (void)hipFree(dGc_);
  (void)hipFree(dWm_);
  (void)hipFree(dtw_);
  (void)hipFree(dWmT_);
  (void)hipFree(dBlockB_);
  (void)hipFree(dGT_);
  (void)hipFree(dG0_);
  (void)hipFree(dSelfW_);
  dSelfW_ = nullptr;
  dGs_ = dGc_ = dWm_ = dtw_ = dWmT_ = nullptr;
  dBlockB_ = nullptr;
  dGT_ = dG0_ = nullptr;
  hasFused_ = false;
  fusedSplit_ = false;
}

std::shared_ptr<DeviceFilter> DeviceFilter::Create(int device, const FilterConfig &config, std::vector<float> taps,
                                                   int flags, std::string *error) {
  if (!UseDevice(device, error)) {
    return nullptr;
  }
  std::shared_ptr<DeviceFilter> f(new DeviceFilter());
  f->device_ = device;
  f->config_ = config;
  f->taps_ = std::move(taps);
  f->flags_ = flags & kLoadRefCompatSpectrum;  // the only load flag of the public boundary
  if (config.fftSize / (2 * std::max<std::size_t>(config.upsampleFactor, 1)) == 16384 && std::getenv("MIUPS_EXP_FORCE_SPLIT")) {
    f->flags_ |= kLoadInternalForceSplit;  // experiment switch (profiles/): K = 16384 as two 8192-point halves
  }
  if (!f->Rebuild(nullptr, error)) {
    return nullptr;
  }
  return f;
}

std::shared_ptr<DeviceFilter> DeviceFilter::Fork(std::string *error) const {
  return Create(device_, config_, taps_, flags_, error);
}

bool DeviceFilter::Rebuild(const std::vector<std::complex<double>> *eqHalf, std::string *error) {
  FilterTables t;
  if (!BuildTables(config_, taps_, eqHalf, flags_, &t, error)) {
    return false;
  }
  if (!UseDevice(device_, error)) {
    return false;
  }
  // tables may be in use by enqueued work of any engine sharing this filter
  MI_HIP(hipDeviceSynchronize());
  Free();
  geo_ = t.geo;
  if (!(Upload(t.Gs, &dGs_, error) && Upload(t.Gc, &dGc_, error) && Upload(t.Wm, &dWm_, error) &&
        Upload(t.tw, &dtw_, error))) {
    return false;
  }
  if (t.hasFused) {
    if (!(Upload(t.WmT, &dWmT_, error) && Upload(t.blockB, &dBlockB_, error) && Upload(t.GT, &dGT_, error) &&
          Upload(t.G0, &dG0_, error) && Upload(t.selfW, &dSelfW_, error))) {
      return false;
    }
    wb_ = t.Wb;
    hasFused_ = true;
    fusedSplit_ = t.fusedSplit;
  }
  return true;
}

bool DeviceFilter::SetEq(const std::string &apoText, double fsOut, std::string *error) {
  if (apoText.empty()) {
    return Rebuild(nullptr, error);
  }
  std::vector<std::complex<double>> half;
  if (!EqResponseDevice(device_, apoText, config_.fftSize / 2 + 1, config_.fftSize, fsOut, &half, error)) {
    return false;
  }
  return Rebuild(&half, error);
}

bool EqResponseDevice(int device, const std::string &apoText, std::size_t numBins, std::size_t fullFft, double fsOut,
                      std::vector<std::complex<double>> *out, std::string *error) {
  if (!UseDevice(device, error)) {
    return false;
  }
  if (numBins == 0 || fullFft == 0 || !(fsOut > 0.0)) {
    if (error) {
      *error = "invalid EQ grid";
    }
    return false;
  }
  eq::EqProfile profile;
  eq::parseEqString(apoText, profile);  // an empty profile evaluates to unity, as in the reference
  const eq::Cascade cascade = eq::buildCascade(profile, fsOut);
  eq::BiquadCoeffs *dSec = nullptr;
  double2 *dOut = nullptr;
  const std::size_t nsec = cascade.sections.size();
  MI_HIP(hipMalloc(reinterpret_cast<void **>(&dSec), std::max<std::size_t>(nsec, 1) * sizeof(eq::BiquadCoeffs)));
  bool ok = HipOk(hipMalloc(reinterpret_cast<void **>(&dOut), numBins * sizeof(double2)), "hipMalloc", error);
  if (ok && nsec) {
    ok = HipOk(hipMemcpy(dSec, cascade.sections.data(), nsec * sizeof(eq::BiquadCoeffs), hipMemcpyHostToDevice),
               "hipMemcpy", error);
  }
  if (ok) {
    const double df = fsOut / static_cast<double>(fullFft);
    hipLaunchKernelGGL(eq_response_kernel, dim3(Blocks(static_cast<long long>(numBins), 256)), dim3(256), 0, 0,
                       cascade.preampLinear, profile.preampDb != 0.0 ? 1 : 0, dSec, static_cast<int>(nsec), fsOut, df,
                       numBins, dOut);
    ok = HipOk(hipGetLastError(), "eq_response_kernel", error);
  }
  if (ok) {
    out->resize(numBins);
    static_assert(sizeof(std::complex<double>) == sizeof(double2), "layout");
    ok = HipOk(hipMemcpy(out->data(), dOut, numBins * sizeof(double2), hipMemcpyDeviceToHost), "hipMemcpy", error);
  }
  (void)hipFree(dSec);
  (void)hipFree(dOut);
  return ok;
}

// ------------------------------------------------------------------ Engine --
Engine::~Engine() {
  if (filter_) {
    (void)hipSetDevice(filter_->device());
  }
  (void)hipFree(hist_[0]);
  (void)hipFree(hist_[1]);
  for (auto *w : work_) {
    (void)hipFree(w);
  }
  (void)hipFree(stageIn_);
  (void)hipFree(stageOut_);
  (void)hipFree(scratch_);
  (void)hipFree(planar_);
  for (void *e : evStart_) {
    (void)hipEventDestroy(static_cast<hipEvent_t>(e));
  }
  for (void *e : evStop_) {
    (void)hipEventDestroy(static_cast<hipEvent_t>(e));
  }
}

std::unique_ptr<Engine> Engine::Create(std::shared_ptr<DeviceFilter> filter, int streams, int channels, int inFmt,
                                       int outFmt, std::string *error) {
  auto bad = [&](const char *m) -> std::unique_ptr<Engine> {
    if (error) {
      *error = m;
    }
    return nullptr;
  };
  if (!filter) {
    return bad("null filter");
  }
  if (streams <= 0 || channels <= 0 || static_cast<long long>(streams) * channels > (1 << 20)) {
    return bad("streams/channels out of range");
  }
  if (inFmt < kF32 || inFmt > kS32 || outFmt < kF32 || outFmt > kS32) {
    return bad("unknown PCM format");
  }
  if (!UseDevice(filter->device(), error)) {
    return nullptr;
  }
  std::unique_ptr<Engine> e(new Engine());
  e->filter_ = std::move(filter);
  e->streams_ = streams;
  e->channels_ = channels;
  e->inFmt_ = inFmt;
  e->outFmt_ = outFmt;
  const Geometry &g = e->filter_->geometry();
  e->fused_ = e->filter_->hasFused() && FusedCovers(g, channels, inFmt, outFmt);
  // channels per workgroup are chosen per call (PickChannelGroup)
  e->cg_ = 1;
  e->groups_ = channels;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->filter_->device()) != hipSuccess || cus <= 0) {
    cus = 256;
  }
  e->cuCount_ = cus;
  e->histStride_ = static_cast<std::size_t>(g.hist_frames) * channels * pcm_bytes(inFmt);
  const std::size_t bytes = std::max<std::size_t>(e->histStride_ * streams, 16);
  for (int i = 0; i < 2; ++i) {
    if (!HipOk(hipMalloc(&e->hist_[i], bytes), "hipMalloc(history)", error) ||
        !HipOk(hipMemset(e->hist_[i], 0, bytes), "hipMemset(history)", error)) {
      return nullptr;
    }
  }
  return e;
}

std::unique_ptr<Engine> Engine::Clone(std::string *error) const {
  auto e = Create(filter_, streams_, channels_, inFmt_, outFmt_, error);
  if (!e) {
    return nullptr;
  }
  const std::size_t bytes = histStride_ * streams_;
  if (bytes) {
    // default-stream copy: ordered after every earlier enqueue on that stream
    if (!HipOk(hipMemcpy(e->hist_[e->cur_], hist_[cur_], bytes, hipMemcpyDeviceToDevice), "hipMemcpy(history)", error)) {
      return nullptr;
    }
  }
  return e;
}

bool Engine::Reset(std::string *error) {
  if (!UseDevice(filter_->device(), error)) {
    return false;
  }
  const std::size_t bytes = std::max<std::size_t>(histStride_ * streams_, 16);
  MI_HIP(hipDeviceSynchronize());
  MI_HIP(hipMemset(hist_[0], 0, bytes));
  MI_HIP(hipMemset(hist_[1], 0, bytes));
  return true;
}

bool Engine::EnsureWork(std::size_t items, std::string *error) {
  if (items <= workItems_) {
    return true;
  }
  const Geometry &g = filter_->geometry();
  for (auto *&w : work_) {
    (void)hipFree(w);
    w = nullptr;
  }
  workItems_ = 0;
  const std::size_t row = static_cast<std::size_t>(g.K) * sizeof(cf);
  MI_HIP(hipMalloc(reinterpret_cast<void **>(&work_[0]), items * row));
  MI_HIP(hipMalloc(reinterpret_cast<void **>(&work_[1]), items * row));
  MI_HIP(hipMalloc(reinterpret_cast<void **>(&work_[2]), items * row * g.P));
  MI_HIP(hipMalloc(reinterpret_cast<void **>(&work_[3]), items * row * g.P));
  workItems_ = items;
  return true;
}

// Channels per workgroup for this call. A workgroup that owns whole frames (mono,
// stereo) writes them itself from its staging planes, which are still in cache. Wider
// frames, and stereo calls too small to give every CU a workgroup, run one channel per
// workgroup and leave the frames to interleave_*_kernel (ProcessDevice).
void Engine::PickChannelGroup(std::size_t blocks) {
  const Geometry &g = filter_->geometry();
  const int ldsK = filter_->fusedSplit() ? g.K / 2 : g.K;  // transform length held in LDS
  const int threads = std::max(ldsK / 32, 1);
  const int byLds = std::max(1, (160 * 1024) / std::max(ldsK * 8, 1));
  const int byWaves = std::max(1, 8 / std::max(threads / 64, 1));
  const std::size_t capacity = static_cast<std::size_t>(cuCount_) * std::min(byLds, byWaves);
  wgCapacity_ = std::max<std::size_t>(capacity, 1);
  bool whole = channels_ == 1 || (channels_ == 2 && blocks * streams_ >= capacity);
  if (channels_ == 2 && std::getenv("MIUPS_EXP_STEREO_EXT")) {  // experiment switch (profiles/)
    whole = false;
  }
  cg_ = whole ? channels_ : 1;
  groups_ = channels_ / cg_;
}

bool Engine::ProcessDevice(const void *dIn, std::size_t inStride, void *dOut, std::size_t outStride,
                           std::size_t blocks, void *hipStream, std::string *error) {
  if (!dIn || !dOut || blocks == 0) {
    if (error) {
      *error = "null buffer or zero blocks";
    }
    return false;
  }
  const Geometry &g = filter_->geometry();
  const std::size_t inFrameBytes = static_cast<std::size_t>(channels_) * pcm_bytes(inFmt_);
  const std::size_t outFrameBytes = static_cast<std::size_t>(channels_) * pcm_bytes(outFmt_);
  const std::size_t inBytes = blocks * g.n_in * inFrameBytes;
  const std::size_t outBytes = blocks * static_cast<std::size_t>(g.B) * outFrameBytes;
  // shape checks on the host before anything is launched: a stream must not
  // overlap the next one, and the item count must fit the grid.
  if (streams_ > 1 && (inStride < inBytes || outStride < outBytes)) {
    if (error) {
      *error = "stream stride smaller than one stream's data";
    }
    return false;
  }
  const unsigned long long items = static_cast<unsigned long long>(blocks) * streams_ * channels_;
  if (items > 0x7fffffffull || blocks > (1u << 24)) {
    if (error) {
      *error = "too many channel-blocks in one call";
    }
    return false;
  }
  if (!UseDevice(filter_->device(), error)) {
    return false;
  }
  hipStream_t st = static_cast<hipStream_t>(hipStream);

  IoDesc io{};
  io.in = dIn;
  io.hist = hist_[cur_];
  io.out = dOut;
  io.in_stream_stride = static_cast<long long>(inStride);
  io.hist_stream_stride = static_cast<long long>(histStride_);
  io.out_stream_stride = static_cast<long long>(outStride);
  io.channels = channels_;
  io.streams = streams_;
  io.in_fmt = inFmt_;
  io.out_fmt = outFmt_;
  io.blocks = static_cast<int>(blocks);

  const bool timing = !evStart_.empty();
  const std::size_t slot = timing ? static_cast<std::size_t>(evCount_ % static_cast<long long>(evStart_.size())) : 0;
  if (timing) {
    MI_HIP(hipEventRecord(static_cast<hipEvent_t>(evStart_[slot]), st));
  }
  if (fused_) {
    // one workgroup per (stream, block, channel group); launches are chunked by whole
    // (stream, block) pairs so that the fp32 staging planes (channels * B floats per
    // pair) stay bounded. (Measured: keeping a chunk inside the 256 MiB Infinity Cache
    // gains less than launching fewer, fuller rounds of workgroups -- profiles/r01_summary.md.)
    PickChannelGroup(blocks);
    const bool split = filter_->fusedSplit();
    const bool ext = cg_ < channels_ || split;  // the split kernel has no epilogue of its own
    const std::size_t pairs = static_cast<std::size_t>(blocks) * streams_;
    const std::size_t perPair = static_cast<std::size_t>(channels_) * g.B * sizeof(float);
    std::size_t budget = static_cast<std::size_t>(1024) << 20;
    if (const char *mb = std::getenv("MIUPS_EXP_CHUNK_MB")) {  // experiment switch (profiles/)
      budget = static_cast<std::size_t>(std::max(1, std::atoi(mb))) << 20;
    }
    std::size_t chunk = std::max<std::size_t>(1, std::min<std::size_t>(pairs, budget / perPair));
    if (chunk < pairs && chunk * groups_ > wgCapacity_) {
      // several launches: make each a whole number of full-chip rounds of workgroups
      const std::size_t wgs = chunk * groups_ - (chunk * groups_) % wgCapacity_;
      chunk = std::max<std::size_t>(1, wgs / groups_);
    }
    if (chunk * perPair > scratchBytes_) {
      (void)hipFree(scratch_);
      scratch_ = nullptr;
      scratchBytes_ = 0;
      MI_HIP(hipMalloc(reinterpret_cast<void **>(&scratch_), chunk * perPair));
      scratchBytes_ = chunk * perPair;
    }
    io.scratch = scratch_;
    io.cg = cg_;
    io.groups = groups_;
    io.out_vec_ok = (reinterpret_cast<std::uintptr_t>(dOut) % 16 == 0 && outStride % 16 == 0 &&
                     (static_cast<std::size_t>(g.B) * channels_ * 4) % 16 == 0)
                        ? 1
                        : 0;
    IoDesc ioF = io;  // what the fused kernel reads (io keeps the caller's buffers for the history carry)
    if (channels_ > 2 && channels_ <= kMaxPlanarChannels) {
      // wide frames: de-interleave (history ++ new frames) once, coalesced, instead of
      // gathering one sample per cache line in every channel's first pass
      const long long total = static_cast<long long>(g.hist_frames) + static_cast<long long>(blocks) * g.n_in;
      const long long planeFloats = (total + 3) / 4 * 4;
      const std::size_t need = static_cast<std::size_t>(planeFloats) * channels_ * streams_ * sizeof(float);
      if (need > planarBytes_) {
        (void)hipFree(planar_);
        planar_ = nullptr;
        planarBytes_ = 0;
        MI_HIP(hipMalloc(reinterpret_cast<void **>(&planar_), need));
        planarBytes_ = need;
      }
      const int tiles = static_cast<int>((total + kPlanarTile - 1) / kPlanarTile);
      const std::size_t lds = static_cast<std::size_t>(channels_) * (kPlanarTile + 1) * sizeof(float);
      hipLaunchKernelGGL(planarize_kernel, dim3(static_cast<unsigned>(tiles) * streams_), dim3(256), lds, st, g, io,
                         planar_, planeFloats, total, tiles);
      if (!HipOk(hipGetLastError(), "planarize_kernel", error)) {
        return false;
      }
      ioF.in = planar_;
      ioF.in_fmt = kF32;
      ioF.in_planar = 1;
      ioF.in_plane_stride = planeFloats * static_cast<long long>(sizeof(float));
      ioF.in_stream_stride = ioF.in_plane_stride * channels_;
    }
    ioF.ext_epilogue = ext ? 1 : 0;
    ioF.split_planes = split ? 1 : 0;
    const bool quad = ioF.out_vec_ok && (outFmt_ == kF32 || outFmt_ == kS32) && (g.P * channels_) % 4 == 0 &&
                      g.Bc % 4 == 0;
    for (std::size_t p0 = 0; p0 < pairs; p0 += chunk) {
      const std::size_t np = std::min<std::size_t>(chunk, pairs - p0);
      ioF.item0 = static_cast<int>(p0 * groups_);
      if (!DispatchFused(g, ioF, *filter_, static_cast<unsigned>(np * groups_), st, error)) {
        return false;
      }
      if (!ext) {
        continue;
      }
      // staging planes of this chunk -> interleaved PCM frames
      if (quad) {
        const int threads = 256, perWg = threads * 4;  // interleave_quad_kernel: kUnits = 4
        const long long units = static_cast<long long>(g.Bc / 4) * (g.P * channels_ / 4);
        const int wgsPerPair = static_cast<int>((units + perWg - 1) / perWg);
        const dim3 grid(static_cast<unsigned>(np) * wgsPerPair);
        if (outFmt_ == kF32) {
          hipLaunchKernelGGL(interleave_quad_kernel<kF32>, grid, dim3(threads), 0, st, g, ioF, scratch_,
                             static_cast<int>(p0), static_cast<int>(np), wgsPerPair);
        } else {
          hipLaunchKernelGGL(interleave_quad_kernel<kS32>, grid, dim3(threads), 0, st, g, ioF, scratch_,
                             static_cast<int>(p0), static_cast<int>(np), wgsPerPair);
        }
      } else {
        const long long total = static_cast<long long>(np) * g.B * channels_;
        hipLaunchKernelGGL(interleave_scalar_kernel, dim3(Blocks(total, 256)), dim3(256), 0, st, g, ioF, scratch_,
                           static_cast<int>(p0), static_cast<int>(np));
      }
      if (!HipOk(hipGetLastError(), "interleave kernel", error)) {
        return false;
      }
    }
  } else {
    const std::size_t perItem = static_cast<std::size_t>(2 + 2 * g.P) * g.K * sizeof(cf);
    const std::size_t budget = static_cast<std::size_t>(768) << 20;
    const std::size_t chunk = std::max<std::size_t>(1, std::min<std::size_t>(items, budget / std::max<std::size_t>(perItem, 1)));
    if (!EnsureWork(chunk, error)) {
      return false;
    }
    const int threads = 256;
    for (std::size_t item0 = 0; item0 < items; item0 += chunk) {
      const int n = static_cast<int>(std::min<std::size_t>(chunk, items - item0));
      const long long elems = static_cast<long long>(n) * g.K;
      hipLaunchKernelGGL(gen_load_kernel, dim3(Blocks(elems, threads)), dim3(threads), 0, st, g, io, work_[0],
                         static_cast<int>(item0), n);
      cf *Z = StagedFft<-1>(work_[0], work_[1], filter_->tw(), g.log2k, n, st);
      hipLaunchKernelGGL(gen_multiply_kernel, dim3(Blocks(elems, threads)), dim3(threads), 0, st, g, Z, work_[2],
                         filter_->Gs(), filter_->Gc(), filter_->Wm(), n);
      cf *y = StagedFft<+1>(work_[2], work_[3], filter_->tw(), g.log2k, static_cast<long long>(n) * g.P, st);
      hipLaunchKernelGGL(gen_store_kernel, dim3(Blocks(elems * g.P, threads)), dim3(threads), 0, st, g, io, y,
                         static_cast<int>(item0), n);
    }
    if (!HipOk(hipGetLastError(), "staged kernels", error)) {
      return false;
    }
  }
  if (timing) {
    MI_HIP(hipEventRecord(static_cast<hipEvent_t>(evStop_[slot]), st));
    ++evCount_;
  }

  // carry the last hist_frames input frames of every stream to the next call
  const long long histBytes = static_cast<long long>(histStride_) * streams_;
  if (histBytes > 0) {
    hipLaunchKernelGGL(update_history_kernel, dim3(Blocks(histBytes, 256)), dim3(256), 0, st, g, io, hist_[1 - cur_],
                       static_cast<long long>(blocks) * g.n_in);
    if (!HipOk(hipGetLastError(), "update_history_kernel", error)) {
      return false;
    }
    cur_ = 1 - cur_;
  }
  return true;
}

bool Engine::ProcessHost(const void *hIn, std::size_t inStride, void *hOut, std::size_t outStride,
                         std::size_t blocks, std::string *error) {
  if (!hIn || !hOut || blocks == 0) {
    if (error) {
      *error = "null buffer or zero blocks";
    }
    return false;
  }
  if (!UseDevice(filter_->device(), error)) {
    return false;
  }
  const Geometry &g = filter_->geometry();
  const std::size_t inRow = blocks * g.n_in * channels_ * pcm_bytes(inFmt_);
  const std::size_t outRow = blocks * static_cast<std::size_t>(g.B) * channels_ * pcm_bytes(outFmt_);
  if (streams_ > 1 && (inStride < inRow || outStride < outRow)) {
    if (error) {
      *error = "stream stride smaller than one stream's data";
    }
    return false;
  }
  const std::size_t inBytes = inRow * streams_, outBytes = outRow * streams_;
  if (inBytes > stageInBytes_) {
    (void)hipFree(stageIn_);
    stageIn_ = nullptr;
    stageInBytes_ = 0;
    MI_HIP(hipMalloc(&stageIn_, inBytes));
    stageInBytes_ = inBytes;
  }
  if (outBytes > stageOutBytes_) {
    (void)hipFree(stageOut_);
    stageOut_ = nullptr;
    stageOutBytes_ = 0;
    MI_HIP(hipMalloc(&stageOut_, outBytes));
    stageOutBytes_ = outBytes;
  }
  for (int s = 0; s < streams_; ++s) {
    MI_HIP(hipMemcpyAsync(static_cast<char *>(stageIn_) + s * inRow, static_cast<const char *>(hIn) + s * inStride,
                          inRow, hipMemcpyHostToDevice, nullptr));
  }
  if (!ProcessDevice(stageIn_, inRow, stageOut_, outRow, blocks, nullptr, error)) {
    return false;
  }
  for (int s = 0; s < streams_; ++s) {
    MI_HIP(hipMemcpyAsync(static_cast<char *>(hOut) + s * outStride, static_cast<const char *>(stageOut_) + s * outRow,
                          outRow, hipMemcpyDeviceToHost, nullptr));
  }
  MI_HIP(hipStreamSynchronize(nullptr));
  return true;
}

bool Engine::EnableTiming(int slots, std::string *error) {
  if (!UseDevice(filter_->device(), error)) {
    return false;
  }
  MI_HIP(hipDeviceSynchronize());
  for (void *e : evStart_) {
    (void)hipEventDestroy(static_cast<hipEvent_t>(e));
  }
  for (void *e : evStop_) {
    (void)hipEventDestroy(static_cast<hipEvent_t>(e));
  }
  evStart_.clear();
  evStop_.clear();
  evCount_ = 0;
  for (int i = 0; i < slots; ++i) {
    hipEvent_t a, b;
    MI_HIP(hipEventCreate(&a));
    evStart_.push_back(a);
    MI_HIP(hipEventCreate(&b));
    evStop_.push_back(b);
  }
  return true;
}

bool Engine::KernelMsStats(double *avg, double *mn, double *mx, int *count) {
  const long long n = std::min<long long>(evCount_, static_cast<long long>(evStart_.size()));
  if (n <= 0) {
    return false;
  }
  double sum = 0.0, lo = 1e300, hi = 0.0;
  for (long long i = 0; i < n; ++i) {
    if (hipEventSynchronize(static_cast<hipEvent_t>(evStop_[i])) != hipSuccess) {
      return false;
    }
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, static_cast<hipEvent_t>(evStart_[i]), static_cast<hipEvent_t>(evStop_[i])) != hipSuccess) {
      return false;
    }
    sum += ms;
    lo = std::min<double>(lo, ms);
    hi = std::max<double>(hi, ms);
  }
  *avg = sum / static_cast<double>(n);
  *mn = lo;
  *mx = hi;
  *count = static_cast<int>(n);
  return true;
}

double Engine::LastKernelMs() {
  if (evStart_.empty() || evCount_ == 0) {
    return -1.0;
  }
  const std::size_t slot = static_cast<std::size_t>((evCount_ - 1) % static_cast<long long>(evStart_.size()));
  if (hipEventSynchronize(static_cast<hipEvent_t>(evStop_[slot])) != hipSuccess) {
    return -1.0;
  }
  float ms = 0.0f;
  if (hipEventElapsedTime(&ms, static_cast<hipEvent_t>(evStart_[slot]), static_cast<hipEvent_t>(evStop_[slot])) !=
      hipSuccess) {
    return -1.0;
  }
  return static_cast<double>(ms);
}

}  // namespace miups

#if defined(MIUPS_STAMPS)
// Diagnostic library variant only (see MI_STAMP in device/kernel_fused.h).
extern "C" int mi_debug_read_stamps(unsigned long long *out, size_t count) {
  const size_t total = sizeof(mi_stamps) / sizeof(unsigned long long);
  if (!out || count < total) {
    return static_cast<int>(total);
  }
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(mi_stamps), sizeof(mi_stamps)) == hipSuccess ? 0 : -1;
}
#endif
