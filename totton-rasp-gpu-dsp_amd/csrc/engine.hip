// Engine implementation + kernel launches (hipcc, gfx950).
#include "engine.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <initializer_list>
#include <cstring>
#include <sstream>

#include "device/kernels_generic.h"
#include "device/kernels_tiled.h"
#include "fused_launch.h"
#include "hip_check.h"
#include "host/eq.h"
#if defined(MIUPS_SINGLE_TU)
// experiment / diagnostic builds (scripts/build_variant.sh): everything in this one translation unit, so that
// -DMIUPS_ONLY_LOG2K=n, -DMIUPS_STAMPS (one stamp buffer) and the other switches need a single compile
#include "fused_launch_impl.h"
#endif

namespace miups {
namespace {

thread_local std::string g_lastError;

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

// The two-level path (device/kernels_tiled.h) for `n` work items starting at item0 (pair-major: whole pairs):
// PCM -> A (work0) -> X (work1) -> per phase B (work2) -> staging planes.
template <int LOG2M, int K1>
bool LaunchTiledSized(int newestFirst, const Geometry &g, const IoDesc &io, const TableSet &tabs, cf *A, cf *X, cf *B, float *planes,
                      int item0, int n, hipStream_t st, std::string *error) {
  using Cfg = TiledRowCfg<LOG2M>;
  static_assert(Cfg::M * K1 <= (1 << 18), "two-level sizes");
  static bool attr_set[64] = {};
  int dev = 0;
  MI_HIP(hipGetDevice(&dev));
  if (Cfg::LDS_BYTES > 64 * 1024 && dev < 64 && !attr_set[dev]) {
    MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&tiled_row_forward_kernel<LOG2M, K1>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
    MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&tiled_row_inverse_kernel<LOG2M, K1>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
    attr_set[dev] = true;
  }
  const int M2 = Cfg::M;
  const int threads = 256;
  hipLaunchKernelGGL((tiled_load_kernel<K1>), dim3(Blocks(static_cast<long long>(n) * M2, threads)), dim3(threads), 0, st, g, io,
                     tabs.tw, A, item0, n);
  TiledRowSrc plain{A, nullptr, nullptr};
  hipLaunchKernelGGL((tiled_row_forward_kernel<LOG2M, K1>), dim3(static_cast<unsigned>(n) * K1), dim3(Cfg::T), Cfg::LDS_BYTES, st,
                     g, plain, tabs.tw, X);
  TiledRowSrc spectral{X, tabs.tWm, reinterpret_cast<const f4 *>(tabs.tGs)};
  hipLaunchKernelGGL((tiled_row_inverse_kernel<LOG2M, K1>), dim3(static_cast<unsigned>(n) * g.P * K1), dim3(Cfg::T),
                     Cfg::LDS_BYTES, st, g, spectral, tabs.tw, B);
  const long long rows = static_cast<long long>(n) * g.P;
  hipLaunchKernelGGL((tiled_store_kernel<K1>), dim3(Blocks(rows * M2, threads)), dim3(threads), 0, st, g, tabs.tw, B, planes, rows,
                     newestFirst);
  return HipOk(hipGetLastError(), "two-level transform kernels", error);
}

bool LaunchTiled(int newestFirst, const Geometry &g, const IoDesc &io, const TableSet &tabs, cf *A, cf *X, cf *B, float *planes,
                 int item0, int n, hipStream_t st, std::string *error) {
  switch (g.log2k) {
    case 15: return LaunchTiledSized<11, 16>(newestFirst, g, io, tabs, A, X, B, planes, item0, n, st, error);
    case 16: return LaunchTiledSized<12, 16>(newestFirst, g, io, tabs, A, X, B, planes, item0, n, st, error);
    case 17: return LaunchTiledSized<13, 16>(newestFirst, g, io, tabs, A, X, B, planes, item0, n, st, error);
    case 18: return LaunchTiledSized<13, 32>(newestFirst, g, io, tabs, A, X, B, planes, item0, n, st, error);
    default: break;
  }
  if (error) {
    *error = "the two-level path does not cover this transform length";
  }
  return false;
}

// interleave_tiled_kernel<FMT, TI, EPT>: TI in {64, 32, 16}, EPT = rows * TI / 1024 in 1..8
template <int FMT, int TI>
bool LaunchInterleaveTiledEpt(const Geometry &g, const IoDesc &io, const float *planes, int sb0, int nb, int tiles, int ept,
                              hipStream_t st) {
  const int rows = g.P * io.channels;
  const dim3 grid(static_cast<unsigned>(nb) * static_cast<unsigned>(tiles)), block(256);
  const std::size_t lds = static_cast<std::size_t>(rows) * (TI + 1) * sizeof(float);
  switch (ept) {
#define MI_TILED_CASE(n)                                                                                             \
  case n:                                                                                                            \
    hipLaunchKernelGGL((interleave_tiled_kernel<FMT, TI, n>), grid, block, lds, st, g, io, planes, sb0, nb, tiles); \
    return true;
    MI_TILED_CASE(1)
    MI_TILED_CASE(2)
    MI_TILED_CASE(3)
    MI_TILED_CASE(4)
    MI_TILED_CASE(5)
    MI_TILED_CASE(6)
    MI_TILED_CASE(7)
    MI_TILED_CASE(8)
#undef MI_TILED_CASE
    default: return false;
  }
}
bool LaunchInterleaveTiled(const Geometry &g, const IoDesc &io, const float *planes, int sb0, int nb, int tiles, int ti,
                           int ept, bool f32, hipStream_t st) {
  if (f32) {
    switch (ti) {
      case 64: return LaunchInterleaveTiledEpt<kF32, 64>(g, io, planes, sb0, nb, tiles, ept, st);
      case 32: return LaunchInterleaveTiledEpt<kF32, 32>(g, io, planes, sb0, nb, tiles, ept, st);
      case 16: return LaunchInterleaveTiledEpt<kF32, 16>(g, io, planes, sb0, nb, tiles, ept, st);
      default: return false;
    }
  }
  switch (ti) {
    case 64: return LaunchInterleaveTiledEpt<kS32, 64>(g, io, planes, sb0, nb, tiles, ept, st);
    case 32: return LaunchInterleaveTiledEpt<kS32, 32>(g, io, planes, sb0, nb, tiles, ept, st);
    case 16: return LaunchInterleaveTiledEpt<kS32, 16>(g, io, planes, sb0, nb, tiles, ept, st);
    default: return false;
  }
}

}  // namespace

#if defined(MIUPS_SINGLE_TU)
// the per-size entry points of fused_launch.h, defined here for the sizes this build holds
#if !defined(MIUPS_ONLY_LOG2K)
#define MI_FUSED_HERE(n) 1
#else
#define MI_FUSED_HERE(n) ((n) == MIUPS_ONLY_LOG2K)
#endif
#define MI_DEFINE_FUSED(n)                                                                                              \
  bool LaunchFusedK##n(const Geometry &g, const IoDesc &io, const FusedTables &ft, bool narrow, bool r32, unsigned items, \
                       hipStream_t st, std::string *error) {                                                          \
    if constexpr (MI_FUSED_HERE(n)) {                                                                                 \
      return LaunchFused<n>(g, io, ft, narrow, r32, items, st, error);                                                \
    } else {                                                                                                          \
      if (error) {                                                                                                    \
        *error = "this experiment build holds no fused kernel for this transform length";                             \
      }                                                                                                               \
      return false;                                                                                                   \
    }                                                                                                                 \
  }
MI_DEFINE_FUSED(5)
MI_DEFINE_FUSED(6)
MI_DEFINE_FUSED(7)
MI_DEFINE_FUSED(8)
MI_DEFINE_FUSED(9)
MI_DEFINE_FUSED(10)
MI_DEFINE_FUSED(11)
MI_DEFINE_FUSED(12)
MI_DEFINE_FUSED(13)
MI_DEFINE_FUSED(14)
#undef MI_DEFINE_FUSED
bool LaunchFusedSplitK14(const Geometry &g, const IoDesc &io, const FusedTables &ft, unsigned items, hipStream_t st,
                         std::string *error) {
  if constexpr (MI_FUSED_HERE(15)) {
    return LaunchFusedSplit<14>(g, io, ft, items, st, error);
  } else {
    if (error) {
      *error = "this experiment build holds no split fused kernel";
    }
    return false;
  }
}
#endif  // MIUPS_SINGLE_TU

namespace {

bool DispatchFused(const Geometry &g, const IoDesc &io, const FusedTables &f, bool split, bool narrow, bool r32,
                   unsigned items, hipStream_t st, std::string *error) {
  if (split) {
    // block transform length K = 32768: two 16384-point transforms through the LDS
    // (the split layout is only built for that size outside the emulation tests)
    if (g.log2k == 15) {
      return LaunchFusedSplitK14(g, io, f, items, st, error);
    }
    if (error) {
      *error = "split fused kernel does not cover this geometry";
    }
    return false;
  }
  switch (g.log2k) {
#define MI_FUSED_CASE(n) \
  case n: return LaunchFusedK##n(g, io, f, narrow, r32, items, st, error);
    MI_FUSED_CASE(5)
    MI_FUSED_CASE(6)
    MI_FUSED_CASE(7)
    MI_FUSED_CASE(8)
    MI_FUSED_CASE(9)
    MI_FUSED_CASE(10)
    MI_FUSED_CASE(11)
    MI_FUSED_CASE(12)
    MI_FUSED_CASE(13)
    MI_FUSED_CASE(14)
#undef MI_FUSED_CASE
    default:
      break;
  }
  if (error) {
    *error = "fused kernel does not cover this geometry";
  }
  return false;
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

// ------------------------------------------------------------ TableSet / pool --
TableSet::~TableSet() {
  if (Gs || Gc || Wm || tw || WmT || selfW || blockB || GT || G0) {
    (void)hipSetDevice(device);
  }
  (void)hipFree(Gs);
  (void)hipFree(Gc);
  (void)hipFree(Wm);
  (void)hipFree(tw);
  (void)hipFree(WmT);
  (void)hipFree(selfW);
  (void)hipFree(blockB);
  (void)hipFree(GT);
  (void)hipFree(G0);
  (void)hipFree(tGs);
  (void)hipFree(tGc);
  (void)hipFree(tWm);
}

struct TablePool {
  std::mutex mu;
  std::vector<TableSet *> free;
  ~TablePool() {
    for (TableSet *t : free) {
      delete t;
    }
  }
};

namespace {

// shared_ptr deleter of a published table set: the last reference (filter or engine snapshot) has gone, so no
// enqueued work reads the arrays any more -- park the set for the next EQ change instead of freeing it (hipFree
// synchronises the device).
struct PoolReturn {
  std::weak_ptr<TablePool> pool;
  void operator()(const TableSet *t) const {
    TableSet *m = const_cast<TableSet *>(t);
    if (std::shared_ptr<TablePool> p = pool.lock()) {
      std::lock_guard<std::mutex> lock(p->mu);
      if (p->free.size() < 4) {
        p->free.push_back(m);
        return;
      }
    }
    delete m;
  }
};

template <typename T>
bool FillArray(T **dst, std::size_t *count, const std::vector<T> &src, hipStream_t st, std::string *error) {
  const std::size_t n = std::max<std::size_t>(src.size(), 1);
  if (!*dst || *count != n) {
    (void)hipFree(*dst);
    *dst = nullptr;
    *count = 0;
    MI_HIP(hipMalloc(reinterpret_cast<void **>(dst), n * sizeof(T)));
    *count = n;
  }
  if (!src.empty()) {
    // ONE copy from pageable memory in flight at a time: several in flight whose sources share a page (neighbouring heap
    // vectors) ended in "Memory access fault by GPU ... on address <host page>" -- as if the runtime pinned and mapped each
    // source for its own copy and the shared page went with the first to complete (profiles/r03_r_multi_fault.txt)
    MI_HIP(hipMemcpyAsync(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, st));
    MI_HIP(hipStreamSynchronize(st));
  }
  return true;
}

}  // namespace

// ------------------------------------------------------------ DeviceFilter --
DeviceFilter::~DeviceFilter() {
  {
    std::lock_guard<std::mutex> lock(mu_);
    cur_.reset();
  }
  if (uploadStream_) {
    (void)hipSetDevice(device_);
    (void)hipStreamDestroy(static_cast<hipStream_t>(uploadStream_));
  }
}

std::shared_ptr<DeviceFilter> DeviceFilter::Create(int device, const FilterConfig &config, std::vector<float> taps,
                                                   int flags, std::string *error, const std::string &apoText,
                                                   double fsOut, double eqLimit, bool eqStrict) {
  if (!UseDevice(device, error)) {
    return nullptr;
  }
  std::shared_ptr<DeviceFilter> f(new DeviceFilter());
  f->device_ = device;
  f->config_ = config;
  f->taps_ = std::move(taps);
  f->flags_ = flags & kLoadRefCompatSpectrum;  // the only load flag of the public boundary
  if (std::getenv("MIUPS_EXP_NARROW")) {  // experiment switch (profiles/): one butterfly per thread, 4 waves per SIMD
    f->flags_ |= kLoadInternalNarrow;
  }
  if (std::getenv("MIUPS_EXP_R32")) {  // experiment switch (profiles/r03_b_*): radix-32 pass plan at K = 8192 / 16384
    f->flags_ |= kLoadInternalR32;
  }
  f->pool_ = std::make_shared<TablePool>();
  f->SetEqLimit(eqLimit, eqStrict);
  if (!f->SetEq(apoText, fsOut, error)) {  // empty text: the plain filter
    return nullptr;
  }
  return f;
}

std::shared_ptr<DeviceFilter> DeviceFilter::Fork(const std::string &apoText, double fsOut, std::string *error) const {
  double limit;
  bool strict;
  {
    std::lock_guard<std::mutex> lock(mu_);
    limit = eqLimit_;
    strict = eqStrict_;
  }
  return Create(device_, config_, taps_, flags_, error, apoText, fsOut, limit, strict);
}

std::shared_ptr<const TableSet> DeviceFilter::tables() const {
  std::lock_guard<std::mutex> lock(mu_);
  return cur_;
}

unsigned long long DeviceFilter::generation() const {
  std::lock_guard<std::mutex> lock(mu_);
  return generation_;
}

// Build the new tables on the host and upload them into a set that nothing reads. Whatever fails on the way, the
// published set is untouched: engines keep running on the old spectrum and the error is returned.
bool DeviceFilter::StageTables(const std::vector<double> *totalFir, Staged *out, std::string *error) {
  FilterTables t;
  if (!BuildTables(config_, taps_, totalFir, flags_, &t, error)) {
    return false;
  }
  if (!UseDevice(device_, error)) {
    return false;
  }
  if (!uploadStream_) {
    hipStream_t st = nullptr;
    MI_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    uploadStream_ = st;
  }
  hipStream_t up = static_cast<hipStream_t>(uploadStream_);
  // the two-level path's copies of the per-bin tables: bin k = k1 + K1 k2 at [k1][k2]
  std::vector<cf> tGs, tGc, tWm;
  if (tiled_covers(t.geo.log2k) && t.Gs.size() == static_cast<std::size_t>(t.geo.P) * t.geo.K) {
    const int K = t.geo.K, K1 = 1 << tiled_log2k1(t.geo.log2k), M2 = K / K1;
    tGs.resize(2 * t.Gs.size());  // {Gs, Gc} interleaved: one 16-byte word per bin and phase (tGc stays empty)
    tWm.resize(t.Wm.size());
    for (int k1 = 0; k1 < K1; ++k1) {
      for (int k2 = 0; k2 < M2; ++k2) {
        const std::size_t to = static_cast<std::size_t>(k1) * M2 + k2, from = static_cast<std::size_t>(k1) + static_cast<std::size_t>(K1) * k2;
        tWm[to] = t.Wm[from];
        for (int p = 0; p < t.geo.P; ++p) {
          tGs[2 * (static_cast<std::size_t>(p) * K + to)] = t.Gs[static_cast<std::size_t>(p) * K + from];
          tGs[2 * (static_cast<std::size_t>(p) * K + to) + 1] = t.Gc[static_cast<std::size_t>(p) * K + from];
        }
      }
    }
  }
  const std::size_t want[12] = {t.Gs.size(), t.Gc.size(), t.Wm.size(), t.tw.size(), t.WmT.size(), t.selfW.size(),
                                t.blockB.size(), t.GT.size(), t.G0.size(), tGs.size(), tGc.size(), tWm.size()};
  std::unique_ptr<TableSet> set;
  {
    std::lock_guard<std::mutex> lock(pool_->mu);
    for (std::size_t i = 0; i < pool_->free.size(); ++i) {
      bool same = true;
      for (int k = 0; k < 12; ++k) {
        same = same && pool_->free[i]->count[k] == std::max<std::size_t>(want[k], 1);
      }
      if (same) {
        set.reset(pool_->free[i]);
        pool_->free.erase(pool_->free.begin() + static_cast<std::ptrdiff_t>(i));
        break;
      }
    }
  }
  if (!set) {
    set.reset(new TableSet());
  }
  set->device = device_;
  bool ok = FillArray(&set->Gs, &set->count[0], t.Gs, up, error) && FillArray(&set->Gc, &set->count[1], t.Gc, up, error) &&
            FillArray(&set->Wm, &set->count[2], t.Wm, up, error) && FillArray(&set->tw, &set->count[3], t.tw, up, error) &&
            FillArray(&set->WmT, &set->count[4], t.WmT, up, error) &&
            FillArray(&set->selfW, &set->count[5], t.selfW, up, error) &&
            FillArray(&set->blockB, &set->count[6], t.blockB, up, error) &&
            FillArray(&set->GT, &set->count[7], t.GT, up, error) && FillArray(&set->G0, &set->count[8], t.G0, up, error) &&
            FillArray(&set->tGs, &set->count[9], tGs, up, error) && FillArray(&set->tGc, &set->count[10], tGc, up, error) &&
            FillArray(&set->tWm, &set->count[11], tWm, up, error);
  if (ok && failNextUpload_) {
    failNextUpload_ = false;
    ok = false;
    if (error) {
      *error = "table upload failed (injected by test hook)";
    }
  }
  // the host vectors must outlive the copies, and the set must be complete before anyone can see it
  if (!HipOk(hipStreamSynchronize(up), "hipStreamSynchronize(upload)", ok ? error : nullptr) || !ok) {
    return false;  // `set` is freed here; cur_ is unchanged
  }
  set->wb = t.Wb;
  set->wself = t.Wself;
  out->set = std::move(set);
  out->geo = t.geo;
  out->hasFused = t.hasFused;
  out->fusedSplit = t.fusedSplit;
  out->fusedNarrow = t.fusedNarrow;
  out->fusedR32 = t.fusedR32;
  return true;
}

void DeviceFilter::Publish(Staged *staged) {
  if (!staged || !staged->set) {
    return;
  }
  std::lock_guard<std::mutex> lock(mu_);
  if (!cur_) {
    geo_ = staged->geo;
    hasFused_ = staged->hasFused;
    fusedSplit_ = staged->fusedSplit;
    fusedNarrow_ = staged->fusedNarrow;
    fusedR32_ = staged->fusedR32;
  }
  staged->set->generation = ++generation_;
  report_ = staged->report;
  cur_ = std::shared_ptr<const TableSet>(staged->set.release(), PoolReturn{pool_});
}

bool DeviceFilter::Stage(const std::string &apoText, double fsOut, Staged *out, std::string *error) {
  out->report = EqReport{};
  double limit;
  bool strict;
  {
    std::lock_guard<std::mutex> lock(mu_);
    limit = eqLimit_;
    strict = eqStrict_;
  }
  out->report.limit = limit;
  if (apoText.empty()) {
    return StageTables(nullptr, out, error);
  }
  if (!(fsOut > 0.0)) {
    if (error) {
      *error = "invalid EQ grid";
    }
    return false;
  }
  // The cascade becomes part of the FIR: recursion over the taps in fp64, cut back to `taps` samples (host/eq.h). The
  // product stays a linear convolution -- nothing wraps inside a block -- and what the cut dropped is measured.
  eq::EqProfile profile;
  eq::parseEqString(apoText, profile);  // an empty profile evaluates to unity, as in the reference
  eq::EqFold fold = eq::FoldCascadeIntoTaps(taps_, eq::buildCascade(profile, fsOut));
  // ... and its response is held against the cascade's own, evaluated per bin on the device (computeEqResponseForFft)
  std::vector<std::complex<double>> half;
  if (!EqResponseDevice(device_, apoText, config_.fftSize / 2 + 1, config_.fftSize, fsOut, &half, error)) {
    return false;
  }
  fold.responseDev = eq::ResponseDeviation(taps_, fold.fir, half, config_.fftSize);
  EqReport &r = out->report;
  r.active = true;
  r.tailL1 = fold.tailL1;
  r.tailL2 = fold.tailL2;
  r.responseDev = fold.responseDev;
  r.tailComplete = fold.tailComplete;
  r.taper = fold.taper;
  r.firTaps = fold.fir.size();
  r.overLimit = fold.tailL1 > limit;
  if (r.overLimit && strict) {
    if (error) {
      *error = EqReportWarning(r);
    }
    return false;
  }
  return StageTables(&fold.fir, out, error);
}

std::string EqReportWarning(const EqReport &r) {
  return r.overLimit ? eq::FoldWarning(r.firTaps, r.tailL1, r.tailL2, r.tailComplete, r.limit) : std::string();
}

bool DeviceFilter::SetEq(const std::string &apoText, double fsOut, std::string *error) {
  Staged staged;
  if (!Stage(apoText, fsOut, &staged, error)) {
    return false;
  }
  Publish(&staged);
  return true;
}

void DeviceFilter::SetEqLimit(double maxTailL1, bool strict) {
  std::lock_guard<std::mutex> lock(mu_);
  eqLimit_ = maxTailL1 >= 0.0 ? maxTailL1 : eq::kFoldDefaultLimit;
  eqStrict_ = strict;
}

EqReport DeviceFilter::eqReport() const {
  std::lock_guard<std::mutex> lock(mu_);
  return report_;
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
  // Everything on a private non-blocking stream with stream-ordered allocations: an EQ change while audio is
  // streaming must neither join the NULL stream (which orders against every blocking stream) nor hipFree (which
  // synchronises the device).
  hipStream_t st = nullptr;
  MI_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  eq::BiquadCoeffs *dSec = nullptr;
  double2 *dOut = nullptr;
  const std::size_t nsec = cascade.sections.size();
  bool pooled = true;
  bool ok = true;
  if (hipMallocAsync(reinterpret_cast<void **>(&dSec), std::max<std::size_t>(nsec, 1) * sizeof(eq::BiquadCoeffs), st) !=
          hipSuccess ||
      hipMallocAsync(reinterpret_cast<void **>(&dOut), numBins * sizeof(double2), st) != hipSuccess) {
    (void)hipGetLastError();
    if (dSec) {
      (void)hipFreeAsync(dSec, st);
      dSec = nullptr;
    }
    pooled = false;  // no stream-ordered allocator on this runtime: plain allocations
    ok = HipOk(hipMalloc(reinterpret_cast<void **>(&dSec), std::max<std::size_t>(nsec, 1) * sizeof(eq::BiquadCoeffs)),
               "hipMalloc", error) &&
         HipOk(hipMalloc(reinterpret_cast<void **>(&dOut), numBins * sizeof(double2)), "hipMalloc", error);
  }
  if (ok && nsec) {
    // pageable source, and a pageable destination follows on the same stream (out->data() below): one such copy in flight
    // at a time (see FillArray) -- the upload is waited for before anything else is enqueued
    ok = HipOk(hipMemcpyAsync(dSec, cascade.sections.data(), nsec * sizeof(eq::BiquadCoeffs), hipMemcpyHostToDevice, st),
               "hipMemcpyAsync", error) &&
         HipOk(hipStreamSynchronize(st), "hipStreamSynchronize(eq sections)", error);
  }
  if (ok) {
    const double df = fsOut / static_cast<double>(fullFft);
    hipLaunchKernelGGL(eq_response_kernel, dim3(Blocks(static_cast<long long>(numBins), 256)), dim3(256), 0, st,
                       cascade.preampLinear, profile.preampDb != 0.0 ? 1 : 0, dSec, static_cast<int>(nsec), fsOut, df,
                       numBins, dOut);
    ok = HipOk(hipGetLastError(), "eq_response_kernel", error);
  }
  if (ok) {
    out->resize(numBins);
    static_assert(sizeof(std::complex<double>) == sizeof(double2), "layout");
    ok = HipOk(hipMemcpyAsync(out->data(), dOut, numBins * sizeof(double2), hipMemcpyDeviceToHost, st), "hipMemcpyAsync",
               error);
  }
  if (pooled) {
    (void)hipFreeAsync(dSec, st);
    (void)hipFreeAsync(dOut, st);
  }
  ok = HipOk(hipStreamSynchronize(st), "hipStreamSynchronize(eq)", ok ? error : nullptr) && ok;
  if (!pooled) {
    (void)hipFree(dSec);
    (void)hipFree(dOut);
  }
  (void)hipStreamDestroy(st);
  return ok;
}

// ------------------------------------------------------------------ Engine --
Engine::~Engine() {
  if (filter_) {
    (void)hipSetDevice(filter_->device());
  }
  Reap(true);  // every enqueued call has finished before its buffers go
  for (void *st : {own_, aux_, h2d_, d2h_}) {
    if (st) {
      (void)hipStreamSynchronize(static_cast<hipStream_t>(st));
      (void)hipStreamDestroy(static_cast<hipStream_t>(st));
    }
  }
  (void)hipFree(hist_[0]);
  (void)hipFree(hist_[1]);
  for (auto *w : work_) {
    (void)hipFree(w);
  }
  for (int i = 0; i < 2; ++i) {
    (void)hipFree(stageIn_[i]);
    (void)hipFree(stageOut_[i]);
  }
  (void)hipFree(scratch_);
  (void)hipFree(planar_);
  (void)hipFree(park_);
  (void)hipFree(fsync_);
  for (void *e : eventPool_) {
    (void)hipEventDestroy(static_cast<hipEvent_t>(e));
  }
  for (void *e : pipeEv_) {
    if (e) {
      (void)hipEventDestroy(static_cast<hipEvent_t>(e));
    }
  }
  for (auto &v : classEv_) {
    for (auto &pr : v) {
      (void)hipEventDestroy(static_cast<hipEvent_t>(pr.first));
      (void)hipEventDestroy(static_cast<hipEvent_t>(pr.second));
    }
  }
  for (void *e : classPool_) {
    (void)hipEventDestroy(static_cast<hipEvent_t>(e));
  }
  for (void *e : evStart_) {
    (void)hipEventDestroy(static_cast<hipEvent_t>(e));
  }
  for (void *e : evStop_) {
    (void)hipEventDestroy(static_cast<hipEvent_t>(e));
  }
}

ExpSwitches ExpSwitches::FromEnvironment() {
  ExpSwitches x;
  auto on = [](const char *name) { return std::getenv(name) != nullptr; };
  auto num = [](const char *name) {
    const char *v = std::getenv(name);
    return v ? std::max(1, std::atoi(v)) : 0;
  };
  x.stereoExt = on("MIUPS_EXP_STEREO_EXT");
  x.park = on("MIUPS_EXP_PARK");
  x.noPhaseParts = on("MIUPS_EXP_NO_PHASE_PARTS");
  x.noTiledInterleave = on("MIUPS_EXP_NO_TILED_INTERLEAVE");
  x.noRowsInterleave = on("MIUPS_EXP_NO_ROWS_INTERLEAVE");
  x.noSplitPlanar = on("MIUPS_EXP_NO_SPLIT_PLANAR");
  x.noTwoLevel = on("MIUPS_EXP_NO_TWO_LEVEL");
  x.twoLevelNoPlanar = on("MIUPS_EXP_TWO_LEVEL_NO_PLANAR");
  x.twoLevelStoreForward = on("MIUPS_EXP_TWO_LEVEL_STORE_FORWARD");
  x.hostThreeStreams = on("MIUPS_EXP_HOST_THREE_STREAMS");
  x.noCoopFrames = on("MIUPS_EXP_NO_COOP_FRAMES");
  x.forceCoopFrames = on("MIUPS_EXP_COOP_FRAMES");  // wherever a tile width fits, whatever the launch shape (tests, A/B)
  x.pitchedAnyWidth = on("MIUPS_EXP_PITCHED_ANY_WIDTH");  // diagnostic (scripts/pitched_abort_repro.py): no host packing
  x.coopCap = num("MIUPS_EXP_COOP_CAP");
  x.tileTi = num("MIUPS_EXP_TILE_TI");
  x.chunkMb = num("MIUPS_EXP_CHUNK_MB");
  x.chunkRounds = num("MIUPS_EXP_CHUNK_ROUNDS");
  x.twoLevelBudgetMb = num("MIUPS_EXP_TWO_LEVEL_BUDGET_MB");
  x.hostSubBatches = num("MIUPS_EXP_HOST_SUBBATCHES");
  if (const char *v = std::getenv("MIUPS_EXP_PIPELINE")) {
    x.pipeline = v[0] == '1' ? 1 : 0;
  }
  return x;
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
  e->exp_ = ExpSwitches::FromEnvironment();
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
  if (!e->EnsureStreams(error)) {
    return nullptr;
  }
  hipStream_t own = static_cast<hipStream_t>(e->own_);
  for (int i = 0; i < 2; ++i) {
    if (!HipOk(hipMalloc(&e->hist_[i], bytes), "hipMalloc(history)", error) ||
        !HipOk(hipMemsetAsync(e->hist_[i], 0, bytes, own), "hipMemset(history)", error)) {
      return nullptr;
    }
  }
  if (!e->MarkDone(own, nullptr, error)) {
    return nullptr;
  }
  return e;
}

bool Engine::EnsureStreams(std::string *error) {
  for (void **slot : {&own_, &aux_, &h2d_, &d2h_}) {
    if (!*slot) {
      hipStream_t st = nullptr;
      MI_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
      *slot = st;
    }
  }
  for (void *&e : pipeEv_) {
    if (!e) {
      hipEvent_t ev = nullptr;
      MI_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      e = ev;
    }
  }
  return true;
}

void *Engine::TakeEvent() {
  if (!eventPool_.empty()) {
    void *e = eventPool_.back();
    eventPool_.pop_back();
    return e;
  }
  hipEvent_t ev = nullptr;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
    return nullptr;
  }
  return ev;
}

// Drop the table snapshots (and recycle the events) of calls that have finished.
void Engine::Reap(bool all) {
  while (!inflight_.empty()) {
    InFlight &f = inflight_.front();
    if (all) {
      (void)hipEventSynchronize(static_cast<hipEvent_t>(f.done));
    } else if (hipEventQuery(static_cast<hipEvent_t>(f.done)) != hipSuccess) {
      (void)hipGetLastError();  // hipErrorNotReady is not an error
      break;
    }
    eventPool_.push_back(f.done);
    inflight_.pop_front();
  }
}

// All engine state is ordered by the chain of per-call events: work on `stream` starts after the previous call's
// work, whichever stream that ran on.
bool Engine::OrderAfterLast(void *stream, std::string *error) {
  Reap(false);
  if (!inflight_.empty() && inflight_.back().stream != stream) {
    MI_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(inflight_.back().done), 0));
  }
  return true;
}

bool Engine::MarkDone(void *stream, std::shared_ptr<const TableSet> tabs, std::string *error) {
  void *ev = TakeEvent();
  if (!ev) {
    if (error) {
      *error = "hipEventCreate failed";
    }
    return false;
  }
  if (!HipOk(hipEventRecord(static_cast<hipEvent_t>(ev), static_cast<hipStream_t>(stream)), "hipEventRecord", error)) {
    eventPool_.push_back(ev);
    return false;
  }
  inflight_.push_back(InFlight{ev, stream, std::move(tabs)});
  while (inflight_.size() > 64) {  // a caller that never synchronises: bound the queue
    (void)hipEventSynchronize(static_cast<hipEvent_t>(inflight_.front().done));
    eventPool_.push_back(inflight_.front().done);
    inflight_.pop_front();
  }
  return true;
}

std::unique_ptr<Engine> Engine::Clone(std::string *error) {
  auto e = Create(filter_, streams_, channels_, inFmt_, outFmt_, error);
  if (!e) {
    return nullptr;
  }
  const std::size_t bytes = histStride_ * streams_;
  if (bytes) {
    // on this engine's own stream, after everything enqueued so far on whatever stream; the clone is usable on return
    hipStream_t own = static_cast<hipStream_t>(own_);
    if (!OrderAfterLast(own_, error) ||
        !HipOk(hipStreamSynchronize(static_cast<hipStream_t>(e->own_)), "hipStreamSynchronize", error) ||
        !HipOk(hipMemcpyAsync(e->hist_[e->cur_], hist_[cur_], bytes, hipMemcpyDeviceToDevice, own), "hipMemcpy(history)",
               error) ||
        !MarkDone(own_, nullptr, error) || !HipOk(hipStreamSynchronize(own), "hipStreamSynchronize", error)) {
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
  hipStream_t own = static_cast<hipStream_t>(own_);
  if (!OrderAfterLast(own_, error)) {
    return false;
  }
  MI_HIP(hipMemsetAsync(hist_[0], 0, bytes, own));
  MI_HIP(hipMemsetAsync(hist_[1], 0, bytes, own));
  return MarkDone(own_, nullptr, error);
}

bool Engine::LoadHistoryHost(const void *h, std::size_t streamStride, std::string *error) {
  if (!h || streamStride < histStride_) {
    if (error) {
      *error = "history buffer missing or stream stride smaller than one stream's history";
    }
    return false;
  }
  if (!UseDevice(filter_->device(), error)) {
    return false;
  }
  hipStream_t own = static_cast<hipStream_t>(own_);
  if (!OrderAfterLast(own_, error)) {
    return false;
  }
  bool ok = true;
  for (int s = 0; ok && s < streams_ && histStride_ > 0; ++s) {
    ok = HipOk(hipMemcpyAsync(static_cast<char *>(hist_[cur_]) + static_cast<std::size_t>(s) * histStride_,
                              static_cast<const char *>(h) + static_cast<std::size_t>(s) * streamStride, histStride_,
                              hipMemcpyHostToDevice, own),
               "hipMemcpyAsync(history)", error) &&
         // neighbouring rows of a pageable buffer: one copy in flight (see FillArray)
         (streams_ <= 1 || HipOk(hipStreamSynchronize(own), "hipStreamSynchronize(history)", error));
  }
  if (!ok || !MarkDone(own_, nullptr, error)) {
    (void)hipStreamSynchronize(own);  // no copy from the caller's buffer is left in flight behind a failure
    return false;
  }
  // the caller may reuse (or change) its buffer when this returns
  return HipOk(hipStreamSynchronize(own), "hipStreamSynchronize(history)", error);
}

bool Engine::Rebind(std::shared_ptr<DeviceFilter> filter, bool resetHistory, std::string *error) {
  if (!filter || filter->device() != filter_->device()) {
    if (error) {
      *error = "Rebind: the new filter must live on the engine's device";
    }
    return false;
  }
  if (!UseDevice(filter->device(), error)) {
    return false;
  }
  const Geometry &g = filter->geometry();
  const Geometry &old = filter_->geometry();
  const std::size_t stride = static_cast<std::size_t>(g.hist_frames) * channels_ * pcm_bytes(inFmt_);
  const bool sameHistory = stride == histStride_;
  // the history is raw input frames: it stays valid under another filter of the same history length
  if (!sameHistory) {
    // new buffers first: a failed allocation leaves the engine exactly as it was (old filter, old history)
    const std::size_t bytes = std::max<std::size_t>(stride * streams_, 16);
    void *fresh[2] = {nullptr, nullptr};
    for (int i = 0; i < 2; ++i) {
      if (!HipOk(hipMalloc(&fresh[i], bytes), "hipMalloc(history)", error)) {
        (void)hipFree(fresh[0]);
        return false;
      }
    }
    Reap(true);  // buffers of the old size are about to be freed
    for (int i = 0; i < 2; ++i) {
      (void)hipFree(hist_[i]);
      hist_[i] = fresh[i];
    }
    histStride_ = stride;
    cur_ = 0;
  }
  if (g.K != old.K || g.P != old.P) {
    // the staged path's work buffers are sized items * K * (1 | P) complex words: a filter with another transform
    // length or phase count must not find buffers counted in the old filter's items (EnsureWork compares item counts)
    Reap(true);
    for (auto *&w : work_) {
      (void)hipFree(w);
      w = nullptr;
    }
    workItems_ = 0;
    workFourth_ = false;
  }
  filter_ = std::move(filter);
  fused_ = filter_->hasFused() && FusedCovers(g, channels_, inFmt_, outFmt_);
  if (resetHistory || !sameHistory) {
    return Reset(error);
  }
  return true;
}

bool Engine::EnsureWork(std::size_t items, std::string *error, bool fourth) {
  if (items <= workItems_ && (!fourth || workFourth_)) {
    return true;
  }
  items = std::max(items, workItems_);
  const Geometry &g = filter_->geometry();
  Reap(true);
  for (auto *&w : work_) {
    (void)hipFree(w);
    w = nullptr;
  }
  workItems_ = 0;
  workFourth_ = false;
  const std::size_t row = static_cast<std::size_t>(g.K) * sizeof(cf);
  MI_HIP(hipMalloc(reinterpret_cast<void **>(&work_[0]), items * row));
  MI_HIP(hipMalloc(reinterpret_cast<void **>(&work_[1]), items * row));
  MI_HIP(hipMalloc(reinterpret_cast<void **>(&work_[2]), items * row * g.P));
  if (fourth) {
    MI_HIP(hipMalloc(reinterpret_cast<void **>(&work_[3]), items * row * g.P));
    workFourth_ = true;
  }
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
  const bool narrow = filter_->fusedNarrow();               // one butterfly per thread, 128 registers: 16 waves per CU
  const int threads = std::max(ldsK / (narrow ? 16 : 32), 1);
  const int byLds = std::max(1, (160 * 1024) / std::max(ldsK * 8, 1));
  const int byWaves = std::max(1, (narrow ? 16 : 8) / std::max(threads / 64, 1));
  const std::size_t capacity = static_cast<std::size_t>(cuCount_) * std::min(byLds, byWaves);
  wgCapacity_ = std::max<std::size_t>(capacity, 1);
  bool whole = channels_ == 1 || (channels_ == 2 && blocks * streams_ >= capacity);
  if (channels_ == 2 && exp_.stereoExt) {  // experiment switch (profiles/)
    whole = false;
  }
  cg_ = whole ? channels_ : 1;
  groups_ = channels_ / cg_;
  // A call with far fewer work items than CUs (the reference's own call shape is ONE channel-block) leaves the chip idle
  // while one workgroup walks through all P phases: give every (block, stream, channel) `parts_` workgroups that take
  // P / parts_ phases each (each repeats the forward transform) and let the interleave kernel write the frames.
  // mi_ups_process_block at the headline filter: kernel 60 -> 2x us (profiles/r03_n_step_overhead.txt).
  parts_ = 0;
  const std::size_t units = blocks * static_cast<std::size_t>(streams_) * channels_;
  // the split form (K = 32768 as two halves) divides by half transforms: 2P pieces per channel-block; the parked-input
  // experiment shares state between the two halves of a phase and keeps the plain form
  const bool splitOk = filter_->fusedSplit() && g.log2k == 15 && !exp_.park;
  const int pieces = filter_->fusedSplit() ? 2 * g.P : g.P;
  if ((splitOk || (!filter_->fusedSplit() && g.log2k >= kPartsMinLog2K && g.log2k <= 14)) && !narrow && !filter_->fusedR32() &&
      smallCallSplit_ && !exp_.noPhaseParts) {  // experiment switch (profiles/)
    for (int d = pieces; d >= 2; --d) {
      if (pieces % d == 0 && units * static_cast<std::size_t>(d) <= static_cast<std::size_t>(cuCount_)) {
        parts_ = d;
        break;
      }
    }
  }
  if (parts_) {
    cg_ = 1;
    groups_ = channels_;
  }
}

// Wide frames: de-interleave (history ++ new frames) once, coalesced, into one fp32 timeline per channel (planar_) and
// point `ioF` at it, instead of gathering one sample per cache line in every channel's first pass.
bool Engine::PlanarizeInput(const Geometry &g, const IoDesc &io, std::size_t blocks, bool splitPlanar, void *stream, IoDesc *ioF,
                            std::string *error) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long total = static_cast<long long>(g.hist_frames) + static_cast<long long>(blocks) * g.n_in;
  const long long planeFloats = (total + 3) / 4 * 4;
  const std::size_t need = static_cast<std::size_t>(planeFloats) * channels_ * streams_ * sizeof(float);
  if (need > planarBytes_) {
    Reap(true);
    (void)hipFree(planar_);
    planar_ = nullptr;
    planarBytes_ = 0;
    MI_HIP(hipMalloc(reinterpret_cast<void **>(&planar_), need));
    planarBytes_ = need;
  }
  const int tileFrames = planar_tile_frames(channels_);
  const int tiles = static_cast<int>((total + tileFrames - 1) / tileFrames);
  const std::size_t lds = static_cast<std::size_t>(channels_) * (tileFrames + 1) * sizeof(float);
  IoDesc ioP = io;
  ioP.split_planes = splitPlanar ? 1 : 0;
  ClassMark(0, st, true);
  hipLaunchKernelGGL(planarize_kernel, dim3(static_cast<unsigned>(tiles) * streams_), dim3(256), lds, st, g, ioP, planar_,
                     planeFloats, total, tiles, tileFrames);
  ClassMark(0, st, false);
  if (!HipOk(hipGetLastError(), "planarize_kernel", error)) {
    return false;
  }
  ioF->in = planar_;
  ioF->in_fmt = kF32;
  ioF->in_planar = splitPlanar ? 2 : 1;
  ioF->in_plane_stride = planeFloats * static_cast<long long>(sizeof(float));
  ioF->in_stream_stride = ioF->in_plane_stride * channels_;
  return true;
}

// staging planes of pairs [p0, p0 + np) -> interleaved PCM frames (shared by the fused and the two-level paths)
bool Engine::LaunchFrames(const Geometry &g, const IoDesc &ioF, float *planes, std::size_t p0, std::size_t np, bool split,
                          bool quad, void *stream, std::string *error, int forceTi) {
  hipStream_t ist = static_cast<hipStream_t>(stream);
  const int rows = g.P * channels_;
  int tiledTi = 0;  // many planes, plain layout: the LDS-tiled form (kernels_generic.h), else the quad form
  if (quad && !split && rows >= 16 && !exp_.noTiledInterleave) {  // experiment switch
    tiledTi = rows <= 128 ? 64 : (rows <= 256 ? 32 : 16);
    if (exp_.tileTi > 0) {  // experiment switch (profiles/): tile width 16 / 32 / 64
      tiledTi = exp_.tileTi;
    }
    if (forceTi > 0) {  // cooperative frames: the tile width the transform kernel's workgroups used
      tiledTi = forceTi;
    }
    const int per = 1024 / tiledTi;  // rows per 16-byte word of a 256-thread pass over the tile
    if (rows % per != 0 || rows / per > 8 || rows > 512) {
      tiledTi = 0;
    }
  }
  if (tiledTi) {
    const int tiles = (g.Bc + tiledTi - 1) / tiledTi;
    if (!LaunchInterleaveTiled(g, ioF, planes, static_cast<int>(p0), static_cast<int>(np), tiles, tiledTi,
                               rows * tiledTi / 1024, outFmt_ == kF32, ist)) {
      if (error) {
        *error = "interleave_tiled_kernel: no instantiation for this frame shape";
      }
      return false;
    }
  } else if (quad && (rows == 4 || rows == 8) && !exp_.noRowsInterleave) {  // experiment switch
    const int threads = 256, perWg = threads * (32 / rows);  // interleave_rows_kernel: kDepth = 32 / R
    const int wgsPerPair = (g.Bc + perWg - 1) / perWg;
    const dim3 grid(static_cast<unsigned>(np) * wgsPerPair);
    const int sb0 = static_cast<int>(p0), nbp = static_cast<int>(np);
    if (outFmt_ == kF32) {
      if (rows == 4) {
        hipLaunchKernelGGL((interleave_rows_kernel<kF32, 4>), grid, dim3(threads), 0, ist, g, ioF, planes, sb0, nbp, wgsPerPair);
      } else {
        hipLaunchKernelGGL((interleave_rows_kernel<kF32, 8>), grid, dim3(threads), 0, ist, g, ioF, planes, sb0, nbp, wgsPerPair);
      }
    } else if (rows == 4) {
      hipLaunchKernelGGL((interleave_rows_kernel<kS32, 4>), grid, dim3(threads), 0, ist, g, ioF, planes, sb0, nbp, wgsPerPair);
    } else {
      hipLaunchKernelGGL((interleave_rows_kernel<kS32, 8>), grid, dim3(threads), 0, ist, g, ioF, planes, sb0, nbp, wgsPerPair);
    }
  } else if (quad) {
    const int threads = 256, perWg = threads * 4;  // interleave_quad_kernel: kUnits = 4
    const long long units = static_cast<long long>(g.Bc / 4) * (g.P * channels_ / 4);
    const int wgsPerPair = static_cast<int>((units + perWg - 1) / perWg);
    const dim3 grid(static_cast<unsigned>(np) * wgsPerPair);
    if (outFmt_ == kF32) {
      hipLaunchKernelGGL(interleave_quad_kernel<kF32>, grid, dim3(threads), 0, ist, g, ioF, planes,
                         static_cast<int>(p0), static_cast<int>(np), wgsPerPair);
    } else {
      hipLaunchKernelGGL(interleave_quad_kernel<kS32>, grid, dim3(threads), 0, ist, g, ioF, planes,
                         static_cast<int>(p0), static_cast<int>(np), wgsPerPair);
    }
  } else {
    const long long total = static_cast<long long>(np) * g.B * channels_;
    hipLaunchKernelGGL(interleave_scalar_kernel, dim3(Blocks(total, 256)), dim3(256), 0, ist, g, ioF, planes,
                       static_cast<int>(p0), static_cast<int>(np));
  }
  return true;
}

// class timing (diagnostic): begin/end event around one launch on the stream it is enqueued on
bool Engine::ClassMark(int cls, void *stream, bool begin) {
  if (!classTiming_) {
    return true;
  }
  hipEvent_t ev = nullptr;
  if (!classPool_.empty()) {
    ev = static_cast<hipEvent_t>(classPool_.back());
    classPool_.pop_back();
  } else if (hipEventCreate(&ev) != hipSuccess) {
    return false;
  }
  if (hipEventRecord(ev, static_cast<hipStream_t>(stream)) != hipSuccess) {
    classPool_.push_back(ev);
    return false;
  }
  if (begin) {
    classEv_[cls].emplace_back(ev, nullptr);
  } else {
    classEv_[cls].back().second = ev;
  }
  return true;
}

bool Engine::LastClassMs(double out[4]) {
  bool any = false;
  for (int c = 0; c < 4; ++c) {
    out[c] = -1.0;
    double sum = 0.0;
    for (auto &pr : classEv_[c]) {
      if (!pr.first || !pr.second || hipEventSynchronize(static_cast<hipEvent_t>(pr.second)) != hipSuccess) {
        continue;
      }
      float ms = 0.0f;
      if (hipEventElapsedTime(&ms, static_cast<hipEvent_t>(pr.first), static_cast<hipEvent_t>(pr.second)) == hipSuccess) {
        sum += ms;
        out[c] = sum;
        any = true;
      }
    }
  }
  return any;
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
  if (!OrderAfterLast(hipStream, error)) {
    return false;
  }
  if (classTiming_) {
    for (auto &v : classEv_) {  // the previous call's pairs go back to the pool
      for (auto &pr : v) {
        if (pr.first) classPool_.push_back(pr.first);
        if (pr.second) classPool_.push_back(pr.second);
      }
      v.clear();
    }
  }
  // the filter tables this call reads: one snapshot for the whole call (an EQ change lands between calls, i.e.
  // between blocks), kept alive until the call's last kernel has finished
  const std::shared_ptr<const TableSet> tabs = filter_->tables();
  lastGeneration_ = tabs->generation;

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

  const bool timing = !evStart_.empty() && (timingCalls_++ % timingEvery_) == 0;
  const std::size_t slot = timing ? static_cast<std::size_t>(evCount_ % static_cast<long long>(evStart_.size())) : 0;
  if (timing) {
    MI_HIP(hipEventRecord(static_cast<hipEvent_t>(evStart_[slot]), st));
  }
  lastTwoLevel_ = false;
  lastCoop_ = false;
  if (fused_) {
    // one workgroup per (stream, block, channel group); launches are chunked by whole (stream, block) pairs so that
    // the fp32 staging planes (channels * B floats per pair) stay bounded.
    PickChannelGroup(blocks);
    const bool split = filter_->fusedSplit();
    const bool ext = cg_ < channels_ || split || parts_ > 0;  // the split and phase-split kernels have no epilogue of their own
    const std::size_t pairs = static_cast<std::size_t>(blocks) * streams_;
    const std::size_t perPair = static_cast<std::size_t>(channels_) * g.P * g.Bp * sizeof(float);  // staging planes
    std::size_t budget = static_cast<std::size_t>(1024) << 20;
    if (exp_.chunkMb > 0) {  // experiment switch (profiles/)
      budget = static_cast<std::size_t>(exp_.chunkMb) << 20;
    }
    std::size_t chunk = std::max<std::size_t>(1, std::min<std::size_t>(pairs, budget / perPair));
    if (chunk < pairs && chunk * groups_ > wgCapacity_) {
      // several launches: make each a whole number of full-chip rounds of workgroups
      const std::size_t wgs = chunk * groups_ - (chunk * groups_) % wgCapacity_;
      chunk = std::max<std::size_t>(1, wgs / groups_);
    }
    // Pipelined launches (planes leave the kernel, ext): the call is cut into chunks of a few full-chip rounds of
    // workgroups; the interleave kernel of chunk k runs on the engine's second stream while the transform kernel of
    // chunk k+1 runs on the caller's -- the bandwidth-bound pass fills the issue slots and the tail of the
    // compute-bound one instead of following it. Two plane buffers alternate.
    // Only where several workgroups share a CU (K <= 4096): a K = 16384 workgroup owns its CU's whole LDS and
    // register file, so an interleave workgroup that lands there keeps a transform workgroup out (measured: configs 4
    // and 5 lose 17-22 % with pipelined launches, config 3 gains 4 %; profiles/r02_b_pipelined_launches.txt).
    io.out_vec_ok = (reinterpret_cast<std::uintptr_t>(dOut) % 16 == 0 && outStride % 16 == 0 &&
                     (static_cast<std::size_t>(g.B) * channels_ * 4) % 16 == 0)
                        ? 1
                        : 0;
    const bool quad = io.out_vec_ok && (outFmt_ == kF32 || outFmt_ == kS32) && (g.P * channels_) % 4 == 0 && g.Bc % 4 == 0;
    // Cooperative frames (device/frame_tile.h): the transform kernel's own workgroups assemble the frames of
    // pairs whose planes are complete while the rest of the launch computes; the frame pass behind the kernel takes what
    // is left. Needs the LDS-tiled frame form with ONE tile width that suits both the transform kernel's T threads
    // (1, 2, 4 or 8 sixteen-byte words per thread, the tile in the workgroup's own LDS) and the 256-thread frame pass.
    // WHERE it pays (same-box A/B, profiles/r04_b_coop_frames.txt): only where several transform workgroups share a CU
    // (K <= 4096: a workgroup that assembles frames leaves three others computing) AND the launch is at least four
    // full-chip rounds of workgroups long (the first round has nothing to assemble, the last round's pairs fall to the
    // frame pass): configs[2] at 512 / 1024 / 2048 blocks +6 / +11 / +5 %, at its stated 256 blocks (two rounds) -9 %;
    // K = 8192 (two workgroups per CU) -13 %, K = 16384 (one) -11..-13 %: there the pass alone stays.
    int coopTi = 0, coopEpt = 0;
    const bool coopShape = wgCapacity_ >= 4 * static_cast<std::size_t>(cuCount_) &&
                           std::min<std::size_t>(chunk, pairs) * groups_ >= 4 * wgCapacity_;  // per launch
    if (ext && !split && parts_ == 0 && !filter_->fusedNarrow() && !filter_->fusedR32() && quad && !exp_.noCoopFrames &&
        !exp_.noTiledInterleave && exp_.pipeline != 1 && (coopShape || exp_.forceCoopFrames)) {
      const int rows = g.P * channels_, T = g.K / 32;
      for (int ti : {64, 32, 16}) {
        const int words = rows * (ti / 4), per = 1024 / ti;
        const bool inKernel = T >= 64 && words % T == 0 && (words / T == 1 || words / T == 2 || words / T == 4 || words / T == 8) &&
                              64 + static_cast<long long>(rows) * (ti + 1) * 4 <= static_cast<long long>(g.K) * 8;
        const bool framePass = rows >= 16 && rows % per == 0 && rows / per <= 8 && rows <= 512;
        if (inKernel && framePass) {
          coopTi = ti;
          coopEpt = words / T;
          break;
        }
      }
    }
    lastCoop_ = coopTi > 0;
    bool pipelined = false;
    const bool sharedCus = wgCapacity_ >= 4 * static_cast<std::size_t>(cuCount_) && coopTi == 0;
    // exp_.pipeline: experiment switch (profiles/): 0 = never, 1 = always, -1 = by shape
    if (ext && (exp_.pipeline >= 0 ? exp_.pipeline == 1 : sharedCus)) {
      const std::size_t totalWgs = pairs * groups_;
      const std::size_t rounds = (totalWgs + wgCapacity_ - 1) / wgCapacity_;
      std::size_t chunkRounds = std::max<std::size_t>(1, rounds / 8);
      if (exp_.chunkRounds > 0) {  // experiment switch (profiles/)
        chunkRounds = static_cast<std::size_t>(exp_.chunkRounds);
      }
      std::size_t c = std::max<std::size_t>(1, chunkRounds * wgCapacity_ / groups_);
      c = std::min(c, std::max<std::size_t>(1, (budget / 2) / perPair));
      if (c < pairs) {
        chunk = c;
        pipelined = true;
      }
    }
    const std::size_t planeBytes = chunk * perPair * (pipelined ? 2 : 1);
    if (planeBytes > scratchBytes_) {
      Reap(true);
      (void)hipFree(scratch_);
      scratch_ = nullptr;
      scratchBytes_ = 0;
      MI_HIP(hipMalloc(reinterpret_cast<void **>(&scratch_), planeBytes));
      scratchBytes_ = planeBytes;
    }
    io.scratch = scratch_;
    io.cg = cg_;
    io.groups = groups_;
    if (coopTi && std::min<std::size_t>(chunk, pairs) > fsyncPairs_) {
      Reap(true);
      (void)hipFree(fsync_);
      fsync_ = nullptr;
      fsyncPairs_ = 0;
      MI_HIP(hipMalloc(&fsync_, std::min<std::size_t>(chunk, pairs) * sizeof(FrameSync)));
      fsyncPairs_ = std::min<std::size_t>(chunk, pairs);
    }
    IoDesc ioF = io;  // what the fused kernel reads (io keeps the caller's buffers for the history carry)
    // The split form reads every second complex word per transform half: from interleaved stereo PCM that is 8 useful
    // bytes per 32 (measured: its two first passes were 36 % of the kernel, profiles/r02_d_*); from a split-planar
    // timeline the lanes read consecutive 8-byte words.
    const bool splitPlanar = split && g.Bc % 4 == 0 && g.hist_frames == g.Oc && g.Oc % 4 == 0 &&
                             !exp_.noSplitPlanar;  // experiment switch (profiles/)
    if ((channels_ > 2 || splitPlanar) && channels_ <= kMaxPlanarChannels) {
      if (!PlanarizeInput(g, io, blocks, splitPlanar, st, &ioF, error)) {
        return false;
      }
    }
    ioF.ext_epilogue = ext ? 1 : 0;
    ioF.phase_parts = parts_;
    ioF.split_planes = split ? 1 : 0;
    ioF.park = nullptr;
    if (split && exp_.park) {  // experiment switch (profiles/r03_g_split_park.txt): measured 0.89x
      // the second half transform of a phase takes its first-pass inputs from here (kernel_fused.h phase_inputs2_both)
      const std::size_t words = static_cast<std::size_t>(split_park_words(g.K / 64));  // T = (K/2) / 32 threads
      const std::size_t need = std::min<std::size_t>(chunk, pairs) * groups_ * words * sizeof(f4);
      if (need > parkBytes_) {
        Reap(true);
        (void)hipFree(park_);
        park_ = nullptr;
        parkBytes_ = 0;
        MI_HIP(hipMalloc(reinterpret_cast<void **>(&park_), need));
        parkBytes_ = need;
      }
      ioF.park = static_cast<f4 *>(park_);
    }
    ioF.fsync = nullptr;
    if (coopTi) {
      const int tiles = (g.Bc + coopTi - 1) / coopTi;
      // a workgroup's share of its pair's tiles and a quarter more (measured best of 8 / 12 / 16 / 26 / 52 at a share of
      // 12.5: 16; twice the share costs 10 %: the workgroup holds registers and LDS a transform workgroup would use)
      int cap = (5 * tiles + 4 * groups_ - 1) / (4 * groups_);
      if (exp_.coopCap > 0) {  // experiment switch (profiles/r04_*)
        cap = exp_.coopCap;
      }
      ioF.fsync = static_cast<FrameSync *>(fsync_);
      ioF.ftile_ti = coopTi;
      ioF.ftile_ept = coopEpt;
      ioF.ftiles = tiles;
      ioF.ftile_cap = std::max(1, cap);  // claims are whole batches of min(8, 32 / ept) tiles: a workgroup stops at or past it
    }
    hipStream_t aux = static_cast<hipStream_t>(aux_);
    std::size_t k = 0;
    bool usedHalf[2] = {false, false};
    for (std::size_t p0 = 0; p0 < pairs; p0 += chunk, ++k) {
      const std::size_t np = std::min<std::size_t>(chunk, pairs - p0);
      const int half = pipelined ? static_cast<int>(k & 1) : 0;
      float *planes = scratch_ + static_cast<std::size_t>(half) * chunk * (perPair / sizeof(float));
      hipEvent_t evFused = static_cast<hipEvent_t>(pipeEv_[half]), evFrames = static_cast<hipEvent_t>(pipeEv_[2 + half]);
      if (pipelined && usedHalf[half]) {
        MI_HIP(hipStreamWaitEvent(st, evFrames, 0));  // the frames of chunk k-2 are out of this plane buffer
      }
      ioF.item0 = static_cast<int>(p0 * groups_);
      ioF.scratch = planes;
      if (ioF.fsync) {
        MI_HIP(hipMemsetAsync(fsync_, 0, np * sizeof(FrameSync), st));
      }
      ClassMark(1, st, true);
      if (!DispatchFused(g, ioF, tabs->fused(), split, filter_->fusedNarrow(), filter_->fusedR32(),
                         static_cast<unsigned>(np * groups_ * static_cast<std::size_t>(parts_ ? parts_ : 1)), st, error)) {
        return false;
      }
      ClassMark(1, st, false);
      if (!ext) {
        continue;
      }
      hipStream_t ist = st;
      if (pipelined) {
        MI_HIP(hipEventRecord(evFused, st));
        MI_HIP(hipStreamWaitEvent(aux, evFused, 0));
        ist = aux;
      }
      // staging planes of this chunk -> interleaved PCM frames
      ClassMark(2, ist, true);
      if (!LaunchFrames(g, ioF, planes, p0, np, split, quad, ist, error, coopTi)) {
        return false;
      }
      ClassMark(2, ist, false);
      if (!HipOk(hipGetLastError(), "interleave kernel", error)) {
        return false;
      }
      if (pipelined) {
        MI_HIP(hipEventRecord(evFrames, aux));
        usedHalf[half] = true;
      }
    }
    if (pipelined) {  // join: the caller's stream continues after the last frames are written
      for (int half = 0; half < 2; ++half) {
        if (usedHalf[half]) {
          MI_HIP(hipStreamWaitEvent(st, static_cast<hipEvent_t>(pipeEv_[2 + half]), 0));
        }
      }
    }
  } else if (tiled_covers(g.log2k) && tabs->count[9] == 2 * static_cast<std::size_t>(g.P) * g.K &&
             static_cast<long long>(blocks) * streams_ * channels_ * g.P * 32 < (1ll << 31) &&
             !exp_.noTwoLevel) {  // experiment switch (profiles/): the pass-per-launch form
    // K = 2^15 .. 2^18 outside the fused kernels (the "2m" filters at 2x / 4x / 8x): two-level transforms with the
    // M2-point rows in LDS (device/kernels_tiled.h), frames by the fused path's interleave kernels
    cg_ = 1;
    groups_ = channels_;
    parts_ = 0;
    lastTwoLevel_ = true;
    const std::size_t pairs = static_cast<std::size_t>(blocks) * streams_;
    const std::size_t perPair = static_cast<std::size_t>(channels_) * g.P * g.Bp * sizeof(float);  // staging planes
    const std::size_t perItem = static_cast<std::size_t>(2 + g.P) * g.K * sizeof(cf);              // A, X and the P rows of B
    std::size_t budget = static_cast<std::size_t>(1024) << 20;
    if (exp_.twoLevelBudgetMb > 0) {  // experiment switch (profiles/r03_q_two_level.txt)
      budget = static_cast<std::size_t>(exp_.twoLevelBudgetMb) << 20;
    }
    const std::size_t chunk = std::max<std::size_t>(1, std::min<std::size_t>(pairs, budget / (perItem * channels_ + perPair)));
    if (!EnsureWork(chunk * channels_, error, false)) {
      return false;
    }
    if (chunk * perPair > scratchBytes_) {
      Reap(true);
      (void)hipFree(scratch_);
      scratch_ = nullptr;
      scratchBytes_ = 0;
      MI_HIP(hipMalloc(reinterpret_cast<void **>(&scratch_), chunk * perPair));
      scratchBytes_ = chunk * perPair;
    }
    IoDesc ioT = io;
    ioT.scratch = scratch_;
    ioT.cg = 1;
    ioT.groups = channels_;
    ioT.ext_epilogue = 1;
    ioT.split_planes = 0;
    ioT.out_vec_ok = (reinterpret_cast<std::uintptr_t>(dOut) % 16 == 0 && outStride % 16 == 0 &&
                      (static_cast<std::size_t>(g.B) * channels_ * 4) % 16 == 0)
                         ? 1
                         : 0;
    const bool quad = ioT.out_vec_ok && (outFmt_ == kF32 || outFmt_ == kS32) && (g.P * channels_) % 4 == 0 && g.Bc % 4 == 0;
    // interleaved frames: the column pass of one channel would use 4 bytes of every frame it touches (2x, 8 channels:
    // 160 of 500 us per launch, profiles/r03_q_two_level.txt) -- one fp32 timeline per channel first, as for wide fused frames
    IoDesc ioL = io;
    if (channels_ >= 2 && channels_ <= kMaxPlanarChannels && g.S == 1 && g.hist_frames == g.Oc &&
        !exp_.twoLevelNoPlanar) {  // experiment switch (profiles/)
      if (!PlanarizeInput(g, io, blocks, false, st, &ioL, error)) {
        return false;
      }
    }
    for (std::size_t p0 = 0; p0 < pairs; p0 += chunk) {
      const std::size_t np = std::min<std::size_t>(chunk, pairs - p0);
      ClassMark(1, st, true);
      if (!LaunchTiled(exp_.twoLevelStoreForward ? 0 : 1, g, ioL, *tabs, work_[0], work_[1], work_[2], scratch_, static_cast<int>(p0 * channels_),
                       static_cast<int>(np * channels_), st, error)) {
        return false;
      }
      ClassMark(1, st, false);
      ClassMark(2, st, true);
      if (!LaunchFrames(g, ioT, scratch_, p0, np, false, quad, st, error)) {
        return false;
      }
      ClassMark(2, st, false);
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
      cf *Z = StagedFft<-1>(work_[0], work_[1], tabs->tw, g.log2k, n, st);
      hipLaunchKernelGGL(gen_multiply_kernel, dim3(Blocks(elems, threads)), dim3(threads), 0, st, g, Z, work_[2],
                         tabs->Gs, tabs->Gc, tabs->Wm, n);
      cf *y = StagedFft<+1>(work_[2], work_[3], tabs->tw, g.log2k, static_cast<long long>(n) * g.P, st);
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
    // threads = copy units; the kernel derives the same unit width (16 / 4 / 1 bytes) from the same quantities
    const long long totalIn = static_cast<long long>(blocks) * g.n_in;
    const long long frameBytes = static_cast<long long>(channels_) * pcm_bytes(inFmt_);
    const long long rowBytes = static_cast<long long>(g.hist_frames) * frameBytes;
    const long long fromHist = totalIn >= g.hist_frames ? 0 : (g.hist_frames - totalIn) * frameBytes;
    const unsigned long long mix = static_cast<unsigned long long>(rowBytes) | static_cast<unsigned long long>(fromHist) |
                                   static_cast<unsigned long long>(totalIn * frameBytes) |
                                   static_cast<unsigned long long>(io.in_stream_stride) |
                                   static_cast<unsigned long long>(io.hist_stream_stride) |
                                   reinterpret_cast<std::uintptr_t>(io.in) | reinterpret_cast<std::uintptr_t>(io.hist) |
                                   reinterpret_cast<std::uintptr_t>(hist_[1 - cur_]);
    const int unit = (mix & 15) == 0 ? 16 : ((mix & 3) == 0 ? 4 : 1);
    ClassMark(3, st, true);
    hipLaunchKernelGGL(update_history_kernel, dim3(Blocks(rowBytes / unit * streams_, 256)), dim3(256), 0, st, g, io,
                       hist_[1 - cur_], totalIn);
    ClassMark(3, st, false);
    if (!HipOk(hipGetLastError(), "update_history_kernel", error)) {
      return false;
    }
    cur_ = 1 - cur_;
  }
  return MarkDone(hipStream, tabs, error);
}

// ---- the rule the host paths rest on, and its audit ---------------------------------------------------------------------
// "Never two in-flight asynchronous copies on host ranges that are not page-locked and may share a page" (DESIGN 4,
// profiles/r03_r_multi_fault.txt: the runtime pins a pageable range per copy and the shared page goes with the first copy
// to complete -- a GPU memory fault that aborts the process). HostCopyAudit sees every asynchronous host copy ProcessHost
// issues, with the page span of its host range and whether that range is page-locked, and every host-side wait that
// retires copies; a copy that breaks the rule bumps a process-wide counter (mi_debug_unsafe_host_copies) which the GPU
// tests of the host paths require to stay 0 -- a test of the rule itself instead of reruns that hope to meet the fault.
namespace {
std::atomic<unsigned long long> g_unsafeHostCopies{0};
constexpr std::uintptr_t kHostPage = 4096;
// From this size on the runtime may pin a pageable range for the copy instead of staging it through its own pinned buffers
// (ROCclr GPU_PINNED_MIN_XFER_SIZE: 1 MiB by default; taken eight times lower here). Two such pins alive at once on
// NEIGHBOURING ranges are the fault of profiles/r03_r_multi_fault.txt -- and "neighbouring" is wider than "sharing a 4 KiB
// page": round 4 met it on the one path round 3 had left alone, ONE copy in and ONE copy out (3.2 MB and 26 MB, two numpy
// arrays mapped one after the other, profiles/r04_c_pair_fault.txt). So two large copies on memory that is not page-locked
// are never in flight together, wherever they are.
constexpr std::size_t kRuntimePinsFrom = 128 * 1024;

struct PageSpan {
  std::uintptr_t lo, hi;  // first and last page-aligned address touched
  std::size_t bytes;
  bool touches(const PageSpan &o) const { return lo <= o.hi && o.lo <= hi; }
  // may not be in flight together when neither is page-locked
  bool conflicts(const PageSpan &o) const { return touches(o) || (bytes >= kRuntimePinsFrom && o.bytes >= kRuntimePinsFrom); }
};
PageSpan SpanOf(const void *p, std::size_t bytes) {
  const std::uintptr_t a = reinterpret_cast<std::uintptr_t>(p);
  return PageSpan{a & ~(kHostPage - 1), (a + (bytes ? bytes - 1 : 0)) & ~(kHostPage - 1), bytes};
}

class HostCopyAudit {
 public:
  enum Dir { kIn = 0, kOut = 1 };
  void Issue(Dir d, const void *host, std::size_t bytes, bool pageLocked) {
    if (pageLocked || bytes == 0) {
      return;
    }
    const PageSpan sp = SpanOf(host, bytes);
    for (const auto &v : open_) {
      for (const PageSpan &o : v) {
        if (sp.conflicts(o)) {
          g_unsafeHostCopies.fetch_add(1, std::memory_order_relaxed);
        }
      }
    }
    open_[d].push_back(sp);
  }
  void Retired(Dir d) { open_[d].clear(); }  // the host has waited for every copy of this direction issued so far
  void RetiredAll() {
    open_[0].clear();
    open_[1].clear();
  }

 private:
  std::vector<PageSpan> open_[2];
};
}  // namespace

unsigned long long UnsafeHostCopies() { return g_unsafeHostCopies.load(std::memory_order_relaxed); }

// Both ends of [p, p + bytes) lie in page-locked host memory (hipHostMalloc / hipHostRegister). The first byte alone is not
// enough: a caller may have registered a shorter length, or hand in a view that runs past a pinned allocation.
bool HostRangePageLocked(const void *p, std::size_t bytes) {
  auto locked = [](const void *q) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, q) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    return a.type == hipMemoryTypeHost;
  };
  return p && locked(p) && (bytes <= 1 || locked(static_cast<const char *>(p) + bytes - 1));
}

// Host buffers. The call is cut into consecutive sub-batches of blocks; sub-batch j's H2D copy, its kernels and its D2H
// copy run on three streams, each gated by events, through two device slots per direction: while the kernels of j run,
// j+1 is being copied in and j-1 copied out. (A block depends on earlier blocks only through the input history, which
// ProcessDevice carries from one sub-batch to the next on the compute stream.) Synchronous for the caller.
bool Engine::ProcessHost(const void *hIn, std::size_t inStride, void *hOut, std::size_t outStride,
                         std::size_t blocks, std::string *error, std::size_t inFramePitch, std::size_t outFramePitch) {
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
  const std::size_t inBlock = static_cast<std::size_t>(g.n_in) * channels_ * pcm_bytes(inFmt_);
  const std::size_t outBlock = static_cast<std::size_t>(g.B) * channels_ * pcm_bytes(outFmt_);
  const std::size_t inFrame = static_cast<std::size_t>(channels_) * pcm_bytes(inFmt_);
  const std::size_t outFrame = static_cast<std::size_t>(channels_) * pcm_bytes(outFmt_);
  if ((inFramePitch && inFramePitch < inFrame) || (outFramePitch && outFramePitch < outFrame)) {
    if (error) {
      *error = "frame pitch smaller than this engine's frame";
    }
    return false;
  }
  const bool pitchedIn = inFramePitch > inFrame, pitchedOut = outFramePitch > outFrame;
  // Pitched DMA only for rows the copy engines take whole dwords of: a column group of odd byte width (one s16 channel,
  // packed 24-bit samples) is packed / unpacked on the host instead -- correct, synchronous and slow; the runtime aborted
  // in hipMemcpy2DAsync on 2-byte rows out of hipHostRegister'ed memory (gpurun_out/r03g, recorded in profiles/r03_g_*).
  auto dwordRows = [](const void *base, std::size_t stride, std::size_t pitch, std::size_t width) {
    return reinterpret_cast<std::uintptr_t>(base) % 4 == 0 && stride % 4 == 0 && pitch % 4 == 0 && width % 4 == 0;
  };
  const bool packIn = pitchedIn && !dwordRows(hIn, inStride, inFramePitch, inFrame) && !exp_.pitchedAnyWidth;
  const bool packOut = pitchedOut && !dwordRows(hOut, outStride, outFramePitch, outFrame) && !exp_.pitchedAnyWidth;
  std::vector<char> hostRows;  // the fallback's packed rows of one stream and sub-batch
  // bytes one block takes in the CALLER's buffers
  const std::size_t inBlockHost = pitchedIn ? static_cast<std::size_t>(g.n_in) * inFramePitch : inBlock;
  const std::size_t outBlockHost = pitchedOut ? static_cast<std::size_t>(g.B) * outFramePitch : outBlock;
  if (streams_ > 1 && (inStride < blocks * inBlockHost - (pitchedIn ? inFramePitch - inFrame : 0) ||
                       outStride < blocks * outBlockHost - (pitchedOut ? outFramePitch - outFrame : 0))) {
    if (error) {
      *error = "stream stride smaller than one stream's data";
    }
    return false;
  }
  // sub-batch size: up to 8 per call, but never so small that a launch cannot fill the chip
  std::size_t nsub = std::min<std::size_t>(8, blocks);
  if (exp_.hostSubBatches > 0) {  // experiment switch (profiles/)
    nsub = std::min<std::size_t>(blocks, static_cast<std::size_t>(exp_.hostSubBatches));
  }
  const std::size_t minBlocks = std::max<std::size_t>(1, (static_cast<std::size_t>(cuCount_) + streams_ * channels_ - 1) /
                                                             (static_cast<std::size_t>(streams_) * channels_));
  std::size_t sb = std::max((blocks + nsub - 1) / nsub, std::min(blocks, minBlocks));
  nsub = (blocks + sb - 1) / sb;
  const std::size_t inRow = sb * inBlock, outRow = sb * outBlock;  // device-side stream strides
  if (inRow * streams_ > stageInBytes_ || outRow * streams_ > stageOutBytes_) {
    Reap(true);
    for (int i = 0; i < 2; ++i) {
      (void)hipFree(stageIn_[i]);
      (void)hipFree(stageOut_[i]);
      stageIn_[i] = stageOut_[i] = nullptr;
    }
    stageInBytes_ = stageOutBytes_ = 0;
    for (int i = 0; i < 2; ++i) {
      MI_HIP(hipMalloc(&stageIn_[i], inRow * streams_));
      MI_HIP(hipMalloc(&stageOut_[i], outRow * streams_));
    }
    stageInBytes_ = inRow * streams_;
    stageOutBytes_ = outRow * streams_;
  }
  // A call that issues several copies per buffer (streams are rows of the caller's buffers, sub-batches are pieces of the
  // rows) must not leave them to the runtime's on-the-fly pinning when the buffers are pageable: in-flight copies whose host
  // ranges share a page collide (see FillArray; the first sighting was a two-stream engine, profiles/r03_r_multi_fault.txt).
  // The buffers are page-locked here for the duration of the call -- one registration each, released after the last copy
  // has completed -- unless they already are (mi_host_alloc, mi_host_register, MultiEngine); if the registration is
  // refused, every copy is waited for before the next one is issued.
  struct ScopedPin {
    void *p = nullptr;
    ~ScopedPin() {
      if (p) {
        (void)hipHostUnregister(p);
      }
    }
  } pinIn, pinOut;
  // EVERY exit that is not the successful one -- each MI_HIP below returns at once -- first waits for the three streams:
  // copies of earlier sub-batches may still be in flight on h2d / d2h, and the pins above must not be released (nor the
  // caller handed back a buffer it may free) under a running DMA: that is the very fault the pins exist to prevent.
  // Declared after the pins, so it runs before them.
  struct DrainOnFailure {
    hipStream_t st[3] = {nullptr, nullptr, nullptr};
    bool armed = false;
    ~DrainOnFailure() {
      if (armed) {
        for (hipStream_t q : st) {
          if (q) {
            (void)hipStreamSynchronize(q);
          }
        }
        (void)hipGetLastError();
      }
    }
  } drain;
  HostCopyAudit audit;
  bool serialCopies = false;
  bool lockedIn = false, lockedOut = false;  // the caller's ranges are page-locked for the whole call (theirs or ours)
  const std::size_t inExtent = (static_cast<std::size_t>(streams_) - 1) * inStride + blocks * inBlockHost -
                               (pitchedIn ? inFramePitch - inFrame : 0);
  const std::size_t outExtent = (static_cast<std::size_t>(streams_) - 1) * outStride + blocks * outBlockHost -
                                (pitchedOut ? outFramePitch - outFrame : 0);
  bool waitBetween = false;  // single copy in, single copy out (below): wait for the first before issuing the second
  if (streams_ > 1 || nsub > 1) {
    lockedIn = packIn || HostRangePageLocked(hIn, inExtent);
    lockedOut = packOut || HostRangePageLocked(hOut, outExtent);
    // A buffer whose HEAD is page-locked but whose extent is not (registered for a shorter length, a view that runs past a
    // pinned allocation): the runtime resolves the pointer to that registration and refuses copies that run past it
    // ("invalid argument", measured), and it cannot be registered a second time either. Say what is wrong instead.
    if ((!lockedIn && !packIn && HostRangePageLocked(hIn, 1)) || (!lockedOut && !packOut && HostRangePageLocked(hOut, 1))) {
      if (error) {
        *error = "host buffer is page-locked for only part of its extent: register (mi_host_register) the whole buffer or none of it";
      }
      drain.armed = false;
      return false;
    }
    if (!lockedIn) {
      if (hipHostRegister(const_cast<void *>(hIn), inExtent, hipHostRegisterDefault) == hipSuccess) {
        pinIn.p = const_cast<void *>(hIn);
        lockedIn = true;
      } else {
        (void)hipGetLastError();
        serialCopies = true;
      }
    }
    if (!lockedOut) {
      if (hipHostRegister(hOut, outExtent, hipHostRegisterDefault) == hipSuccess) {
        pinOut.p = hOut;
        lockedOut = true;
      } else {
        (void)hipGetLastError();
        serialCopies = true;
      }
    }
  } else if (!packIn && !packOut && SpanOf(hIn, inExtent).conflicts(SpanOf(hOut, outExtent))) {
    // One copy in and one copy out (the reference's call shape, mi_ups_process_block): the pair itself is two copies. They
    // conflict when they share a page (heap neighbours do) or when both are large enough for the runtime to pin them.
    // Only then ask the runtime (two attribute queries stay off the 69 us path of a one-channel block: 51 KB in, 204 KB
    // out, separate pages); unless both ends are page-locked the copy in is waited for before the copy out is issued.
    lockedIn = HostRangePageLocked(hIn, inExtent);
    lockedOut = HostRangePageLocked(hOut, outExtent);
    waitBetween = !(lockedIn && lockedOut);
  }
  hipStream_t own = static_cast<hipStream_t>(own_), h2d = static_cast<hipStream_t>(h2d_), d2h = static_cast<hipStream_t>(d2h_);
  drain.st[0] = h2d;
  drain.st[1] = own;
  drain.st[2] = d2h;
  drain.armed = true;
  // One sub-batch (the reference's own call shape: one block per call) has nothing to overlap: copy in, kernels and copy
  // out follow each other on the engine's stream, without the three cross-stream events of the pipelined form
  // (mi_ups_process_block p50 132 -> see profiles/r03_n_step_overhead.txt).
  const bool oneStream = nsub == 1 && hostOneStream_ && !exp_.hostThreeStreams;  // experiment switch (profiles/)
  if (oneStream) {
    h2d = d2h = own;
  }
  if (!OrderAfterLast(own_, error)) {
    return false;
  }
  // pipeEv_[4 + slot]: input of the slot copied in; [6 + slot]: kernels of the slot done; [8 + slot]: output copied out
  for (std::size_t j = 0; j < nsub; ++j) {
    const int slot = static_cast<int>(j & 1);
    const std::size_t b0 = j * sb, nb = std::min(sb, blocks - b0);
    hipEvent_t evIn = static_cast<hipEvent_t>(pipeEv_[4 + slot]), evRun = static_cast<hipEvent_t>(pipeEv_[6 + slot]),
               evOut = static_cast<hipEvent_t>(pipeEv_[8 + slot]);
    if (j >= 2) {
      MI_HIP(hipStreamWaitEvent(h2d, evRun, 0));  // the kernels of j-2 have read this input slot
    }
    if (packIn && j >= 2) {
      MI_HIP(hipEventSynchronize(evRun));  // the kernels of j-2 have read this input slot (the copy below is synchronous)
    }
    for (int s = 0; s < streams_; ++s) {
      if (packIn) {
        const std::size_t rows = nb * static_cast<std::size_t>(g.n_in);
        hostRows.resize(rows * inFrame);
        const char *src = static_cast<const char *>(hIn) + s * inStride + b0 * inBlockHost;
        for (std::size_t f = 0; f < rows; ++f) {
          std::memcpy(hostRows.data() + f * inFrame, src + f * inFramePitch, inFrame);
        }
        MI_HIP(hipMemcpy(static_cast<char *>(stageIn_[slot]) + s * inRow, hostRows.data(), rows * inFrame, hipMemcpyHostToDevice));
      } else if (pitchedIn) {
        const char *src = static_cast<const char *>(hIn) + s * inStride + b0 * inBlockHost;
        audit.Issue(HostCopyAudit::kIn, src, nb * inBlockHost - (inFramePitch - inFrame), lockedIn);
        MI_HIP(hipMemcpy2DAsync(static_cast<char *>(stageIn_[slot]) + s * inRow, inFrame, src, inFramePitch, inFrame,
                                nb * static_cast<std::size_t>(g.n_in), hipMemcpyHostToDevice, h2d));
      } else {
        const char *src = static_cast<const char *>(hIn) + s * inStride + b0 * inBlock;
        audit.Issue(HostCopyAudit::kIn, src, nb * inBlock, lockedIn);
        MI_HIP(hipMemcpyAsync(static_cast<char *>(stageIn_[slot]) + s * inRow, src, nb * inBlock, hipMemcpyHostToDevice, h2d));
      }
      if ((serialCopies || waitBetween) && !packIn) {
        MI_HIP(hipStreamSynchronize(h2d));
        audit.Retired(HostCopyAudit::kIn);
        if (oneStream) {
          audit.RetiredAll();  // one stream: everything issued so far has completed
        }
      }
    }
    if (failAtSubBatch_ == static_cast<int>(j)) {  // test hook (mi_debug_fail_host_call_at): as if the runtime had refused a call
      failAtSubBatch_ = -1;
      if (error) {
        *error = "host copy failed (injected by test hook)";
      }
      return false;
    }
    if (!oneStream) {
      MI_HIP(hipEventRecord(evIn, h2d));
      MI_HIP(hipStreamWaitEvent(own, evIn, 0));
    }
    if (j >= 2) {
      MI_HIP(hipStreamWaitEvent(own, evOut, 0));  // the output of j-2 has left this output slot
    }
    if (!ProcessDevice(stageIn_[slot], inRow, stageOut_[slot], outRow, nb, own_, error)) {
      return false;  // `drain` waits for the copies in flight before the pins go
    }
    if (!oneStream) {
      MI_HIP(hipEventRecord(evRun, own));
      MI_HIP(hipStreamWaitEvent(d2h, evRun, 0));
    }
    if (packOut) {  // kernels done: the synchronous copies below may read the output slot
      if (oneStream) {
        MI_HIP(hipStreamSynchronize(own));
      } else {
        MI_HIP(hipEventSynchronize(evRun));
      }
    }
    for (int s = 0; s < streams_; ++s) {
      if (packOut) {
        const std::size_t rows = nb * static_cast<std::size_t>(g.B);
        hostRows.resize(rows * outFrame);
        MI_HIP(hipMemcpy(hostRows.data(), static_cast<const char *>(stageOut_[slot]) + s * outRow, rows * outFrame,
                         hipMemcpyDeviceToHost));
        char *dst = static_cast<char *>(hOut) + s * outStride + b0 * outBlockHost;
        for (std::size_t f = 0; f < rows; ++f) {
          std::memcpy(dst + f * outFramePitch, hostRows.data() + f * outFrame, outFrame);
        }
      } else if (pitchedOut) {
        char *dst = static_cast<char *>(hOut) + s * outStride + b0 * outBlockHost;
        audit.Issue(HostCopyAudit::kOut, dst, nb * outBlockHost - (outFramePitch - outFrame), lockedOut);
        MI_HIP(hipMemcpy2DAsync(dst, outFramePitch, static_cast<const char *>(stageOut_[slot]) + s * outRow, outFrame, outFrame,
                                nb * static_cast<std::size_t>(g.B), hipMemcpyDeviceToHost, d2h));
      } else {
        char *dst = static_cast<char *>(hOut) + s * outStride + b0 * outBlock;
        audit.Issue(HostCopyAudit::kOut, dst, nb * outBlock, lockedOut);
        MI_HIP(hipMemcpyAsync(dst, static_cast<const char *>(stageOut_[slot]) + s * outRow, nb * outBlock, hipMemcpyDeviceToHost,
                              d2h));
      }
      if (serialCopies && !packOut) {
        MI_HIP(hipStreamSynchronize(d2h));
        audit.Retired(HostCopyAudit::kOut);
        if (oneStream) {
          audit.RetiredAll();
        }
      }
    }
    if (!oneStream) {
      MI_HIP(hipEventRecord(evOut, d2h));
    }
  }
  if (!oneStream) {
    MI_HIP(hipStreamSynchronize(d2h));
  }
  MI_HIP(hipStreamSynchronize(own));
  drain.armed = false;  // everything has completed: nothing to wait for
  return true;
}

void *HostAlloc(std::size_t bytes, std::string *error) {
  void *p = nullptr;
  if (!HipOk(hipHostMalloc(&p, std::max<std::size_t>(bytes, 1), hipHostMallocDefault), "hipHostMalloc", error)) {
    return nullptr;
  }
  return p;
}

void HostFree(void *p) {
  if (p) {
    (void)hipHostFree(p);
  }
}

bool HostRegister(void *p, std::size_t bytes, std::string *error) {
  if (!p || bytes == 0) {
    if (error) {
      *error = "null buffer";
    }
    return false;
  }
  return HipOk(hipHostRegister(p, bytes, hipHostRegisterDefault), "hipHostRegister", error);
}

void HostUnregister(void *p) {
  if (p) {
    (void)hipHostUnregister(p);
  }
}

namespace {
// Copy kernels for the measured ceiling. U independent 16-byte loads per lane in flight, then U stores. kPersist: a grid of
// 8 workgroups per CU walks the buffer; else one short-lived workgroup per 256 * U * 16-byte piece (the shape of this
// library's own frame-assembly pass). DeviceCopyRate reports the best of the forms (profiles/r03_h_copy_forms.txt).
template <int U, bool kPersist>
__global__ void copy16_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, std::size_t n) {
  const std::size_t piece = static_cast<std::size_t>(256) * U;
  const std::size_t stride = kPersist ? static_cast<std::size_t>(gridDim.x) * piece : n;
  for (std::size_t base = static_cast<std::size_t>(blockIdx.x) * piece; base < n; base += stride) {
    uint4 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const std::size_t j = base + threadIdx.x + static_cast<std::size_t>(k) * 256;
      if (j < n) {
        v[k] = src[j];
      }
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const std::size_t j = base + threadIdx.x + static_cast<std::size_t>(k) * 256;
      if (j < n) {
        dst[j] = v[k];
      }
    }
    if (!kPersist) {
      break;
    }
  }
}
}  // namespace

bool DeviceCopyRate(int device, std::size_t bytes, int iters, double *gbps, std::string *error) {
  if (!UseDevice(device, error)) {
    return false;
  }
  if (bytes < 16 || iters < 1 || !gbps) {
    if (error) {
      *error = "bad copy-rate arguments";
    }
    return false;
  }
  const std::size_t n = bytes / 16;
  void *a = nullptr, *b = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  bool ok = HipOk(hipMalloc(&a, n * 16), "hipMalloc", error) && HipOk(hipMalloc(&b, n * 16), "hipMalloc", error) &&
            HipOk(hipMemset(a, 1, n * 16), "hipMemset", error) && HipOk(hipEventCreate(&e0), "hipEventCreate", error) &&
            HipOk(hipEventCreate(&e1), "hipEventCreate", error);
  double best = 0.0;
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
  const unsigned persist = static_cast<unsigned>(cus) * 8;
  auto pieces = [&](int u) { return static_cast<unsigned>((n + 256u * u - 1) / (256u * u)); };
  for (int form = 0; ok && form < 5; ++form) {
    for (int i = 0; ok && i < iters + 1; ++i) {  // the first launch of a form is a warm-up
      ok = HipOk(hipEventRecord(e0, nullptr), "hipEventRecord", error);
      const uint4 *src = static_cast<const uint4 *>(a);
      uint4 *dst = static_cast<uint4 *>(b);
      switch (form) {
        case 0: hipLaunchKernelGGL((copy16_kernel<1, true>), dim3(persist), dim3(256), 0, nullptr, src, dst, n); break;
        case 1: hipLaunchKernelGGL((copy16_kernel<4, true>), dim3(persist), dim3(256), 0, nullptr, src, dst, n); break;
        case 2: hipLaunchKernelGGL((copy16_kernel<4, false>), dim3(pieces(4)), dim3(256), 0, nullptr, src, dst, n); break;
        case 3: hipLaunchKernelGGL((copy16_kernel<8, false>), dim3(pieces(8)), dim3(256), 0, nullptr, src, dst, n); break;
        default: hipLaunchKernelGGL((copy16_kernel<2, false>), dim3(pieces(2)), dim3(256), 0, nullptr, src, dst, n); break;
      }
      ok = ok && HipOk(hipEventRecord(e1, nullptr), "hipEventRecord", error) &&
           HipOk(hipEventSynchronize(e1), "hipEventSynchronize", error);
      float ms = 0.0f;
      ok = ok && HipOk(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime", error);
      if (ok && i > 0 && ms > 0.0f) {
        const double rate = 2.0 * static_cast<double>(n) * 16.0 / (static_cast<double>(ms) * 1e-3) / 1e9;
        best = std::max(best, rate);
        if (std::getenv("MIUPS_EXP_COPY_VERBOSE")) {  // experiment switch (profiles/r03_h_copy_forms.txt)
          std::fprintf(stderr, "copy form %d: %.1f GB/s\n", form, rate);
        }
      }
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(a);
  (void)hipFree(b);
  if (ok) {
    *gbps = best;
  }
  return ok;
}

bool Engine::EnableTiming(int slots, std::string *error) {
  if (!UseDevice(filter_->device(), error)) {
    return false;
  }
  Reap(true);  // no recorded pair is still pending when the ring is replaced
  for (void *e : evStart_) {
    (void)hipEventDestroy(static_cast<hipEvent_t>(e));
  }
  for (void *e : evStop_) {
    (void)hipEventDestroy(static_cast<hipEvent_t>(e));
  }
  evStart_.clear();
  evStop_.clear();
  evCount_ = 0;
  timingCalls_ = 0;
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
