#include "spectrum.h"

#include <cmath>

#include "../device/fft_radix.h"
#include "host_fft.h"

namespace miups {
namespace {

int Log2(std::size_t v) {
  int r = 0;
  while ((static_cast<std::size_t>(1) << r) < v) {
    ++r;
  }
  return r;
}

void BuildFusedLayout(FilterTables *t, bool r32) {
  const Geometry &g = t->geo;
  t->hasFused = false;
  if (g.S != 1 || g.log2k < 5 || g.log2k > 14) {
    return;
  }
  t->fusedR32 = r32;
  const int K = g.K, P = g.P, J = K / 16, T = K / 32;
  std::vector<int> blockOfSet(J);
  for (int b = 0; b < J; ++b) {
    blockOfSet[FusedSetOfBlock(b, g.log2k, r32)] = b;
  }
  t->WmT.assign(T, cf{1.0f, 0.0f});
  t->blockB.assign(T, 0);
  t->GT.assign(static_cast<std::size_t>(P) * 16 * T, f4{0.0f, 0.0f, 0.0f, 0.0f});
  t->G0.assign(static_cast<std::size_t>(P) * 17, f4{0.0f, 0.0f, 0.0f, 0.0f});
  t->Wb = t->Wm[J / 2];
  auto pair = [&](int p, int k) {
    const cf gs = t->Gs[static_cast<std::size_t>(p) * K + k];
    const cf gc = t->Gc[static_cast<std::size_t>(p) * K + k];
    return f4{gs.x, gs.y, gc.x, gc.y};
  };
  t->blockB[0] = blockOfSet[J / 2];
  for (int tau = 1; tau < T; ++tau) {
    const int a = FusedSetOfBlock(FusedBlockA(tau, g.log2k, r32), g.log2k, r32);  // 0 < a < J/2
    t->WmT[tau] = t->Wm[a];
    t->blockB[tau] = blockOfSet[J - a];
    for (int p = 0; p < P; ++p) {
      for (int s = 0; s < 16; ++s) {
        t->GT[(static_cast<std::size_t>(p) * 16 + s) * T + tau] = pair(p, a + s * J);
      }
    }
  }
  for (int p = 0; p < P; ++p) {
    for (int s = 0; s <= 8; ++s) {
      t->G0[static_cast<std::size_t>(p) * 17 + s] = pair(p, s * J);
    }
    for (int s = 0; s < 8; ++s) {
      t->G0[static_cast<std::size_t>(p) * 17 + 9 + s] = pair(p, J / 2 + s * J);
    }
    // thread 0 runs the generic sixteen slots on column 0 of GT (its 17th pair stays in G0[16])
    for (int s = 0; s < 16; ++s) {
      t->GT[(static_cast<std::size_t>(p) * 16 + s) * T + 0] = t->G0[static_cast<std::size_t>(p) * 17 + s];
    }
  }
  {
    // twiddle base of thread 0's slots 9..15: W_M^(J/2) * W_32^(s-9) = Wself * W_32^s, M = 32 J
    const double pi = 3.14159265358979323846264338327950288;
    const double a = 2.0 * pi * 17.0 / 64.0;
    t->Wself = cf{static_cast<float>(std::cos(a)), static_cast<float>(std::sin(a))};
  }
  t->hasFused = true;
}

// Narrow layout (fused_kernel<log2k, EXT, 1>, kernel_fused.h "narrow form"): T = K/16 lanes, lane tid of wave w holds
// set S_a (lanes 0..31: tau = 32 w + lane) or its mirror S_(J-a) (lanes 32..63, same tau); tau = 0 holds the two
// self-mirrored sets S_0 and S_(J/2). Per lane: its LDS block in the stride-1 passes (blockB), W_M^k0 (WmT), and for
// its eight pairs j the spectra of bin k0 + j*J (GT [P][8][T]); G0 [P] = lane 0's ninth pair (bin 8 J = K/2).
void BuildFusedNarrowLayout(FilterTables *t) {
  const Geometry &g = t->geo;
  const int K = g.K, P = g.P, J = K / 16, T = K / 16;
  std::vector<int> blockOfSet(J);
  for (int b = 0; b < J; ++b) {
    blockOfSet[FusedSetOfBlock(b, g.log2k, false)] = b;
  }
  t->WmT.assign(T, cf{1.0f, 0.0f});
  t->blockB.assign(T, 0);
  t->GT.assign(static_cast<std::size_t>(P) * 8 * T, f4{0.0f, 0.0f, 0.0f, 0.0f});
  t->G0.assign(static_cast<std::size_t>(P), f4{0.0f, 0.0f, 0.0f, 0.0f});
  t->Wb = t->Wm[J / 2];
  auto pair = [&](int p, int k) {
    const cf gs = t->Gs[static_cast<std::size_t>(p) * K + k];
    const cf gc = t->Gc[static_cast<std::size_t>(p) * K + k];
    return f4{gs.x, gs.y, gc.x, gc.y};
  };
  for (int tid = 0; tid < T; ++tid) {
    const int tau = 32 * (tid >> 6) + (tid & 31), mirror = (tid >> 5) & 1;
    const int a = tau == 0 ? 0 : FusedSetOfBlock(FusedBlockA(tau, g.log2k, false), g.log2k, false);  // 0 <= a < J/2
    const int k0 = mirror ? (tau == 0 ? J / 2 : J - a) : a;
    t->blockB[tid] = blockOfSet[k0];
    t->WmT[tid] = t->Wm[k0];
    for (int p = 0; p < P; ++p) {
      for (int j = 0; j < 8; ++j) {
        t->GT[(static_cast<std::size_t>(p) * 8 + j) * T + tid] = pair(p, k0 + j * J);
      }
    }
  }
  for (int p = 0; p < P; ++p) {
    t->G0[p] = pair(p, 8 * J);
  }
  t->hasFused = true;
  t->fusedNarrow = true;
}

// Split layout (fused_split_kernel<log2k - 1>, kernel_fused.h "split form"): thread sets of
// the half-length transform (Kh = K/2, Jh = Kh/16, T = Kh/32); every half-length mirror
// pair (k, Kh-k) carries the two full-length pairs (k, K-k) and (Kh-k, Kh+k).
bool BuildFusedSplitLayout(FilterTables *t) {
  const Geometry &g = t->geo;
  // (half length >= 1024: the kernel spreads thread 0's sets over 17 lanes, so it needs 17 threads)
  if (g.S != 1 || g.log2k < 11 || g.log2k > 15 || g.Oc % 4 != 0 || g.Bc % 4 != 0) {
    return false;
  }
  const int lh = g.log2k - 1;
  const int K = g.K, Kh = K / 2, P = g.P, J = Kh / 16, T = Kh / 32;
  std::vector<int> blockOfSet(J);
  for (int b = 0; b < J; ++b) {
    blockOfSet[FusedSetOfBlock(b, lh, false)] = b;  // the split form runs the classic plan
  }
  t->WmT.assign(T, cf{1.0f, 0.0f});
  t->blockB.assign(T, 0);
  t->GT.assign(static_cast<std::size_t>(P) * 32 * T, f4{0.0f, 0.0f, 0.0f, 0.0f});
  t->G0.assign(static_cast<std::size_t>(P) * 34, f4{0.0f, 0.0f, 0.0f, 0.0f});
  t->selfW.assign(17, cf{1.0f, 0.0f});
  t->Wb = t->Wm[J / 2];
  auto pair = [&](int p, int k) {
    const cf gs = t->Gs[static_cast<std::size_t>(p) * K + k];
    const cf gc = t->Gc[static_cast<std::size_t>(p) * K + k];
    return f4{gs.x, gs.y, gc.x, gc.y};
  };
  t->blockB[0] = blockOfSet[J / 2];
  for (int tau = 1; tau < T; ++tau) {
    const int a = FusedSetOfBlock(FusedBlockA(tau, lh, false), lh, false);  // 0 < a < J/2
    t->WmT[tau] = t->Wm[a];
    t->blockB[tau] = blockOfSet[J - a];
    for (int p = 0; p < P; ++p) {
      for (int s = 0; s < 16; ++s) {
        const int k = a + s * J;
        t->GT[((static_cast<std::size_t>(p) * 2 + 0) * 16 + s) * T + tau] = pair(p, k);
        t->GT[((static_cast<std::size_t>(p) * 2 + 1) * 16 + s) * T + tau] = pair(p, Kh - k);
      }
    }
  }
  // self lanes (thread 0's sets S_0 and S_{J/2}): lane l <= 8: k = l*J, lane l >= 9: k = J/2 + (l-9)*J;
  // pair 1 = (k, K-k), pair 2 = (Kh-k, Kh+k)  (k = 0: bins 0 and Kh; k = Kh/2: the same pair twice)
  for (int l = 0; l < 17; ++l) {
    const int k = l <= 8 ? l * J : J / 2 + (l - 9) * J;
    t->selfW[l] = t->Wm[k];
    for (int p = 0; p < P; ++p) {
      t->G0[(static_cast<std::size_t>(p) * 2 + 0) * 17 + l] = pair(p, k);
      t->G0[(static_cast<std::size_t>(p) * 2 + 1) * 17 + l] = pair(p, Kh - k);
    }
  }
  t->hasFused = true;
  t->fusedSplit = true;
  return true;
}

}  // namespace

// radices of the fused kernel's forward passes for K = 2^log2k
std::vector<int> FusedRadices(int log2k, bool r32) {
  std::vector<int> r;
  if (r32) {
    r.push_back((1 << log2k) / 512);
    r.push_back(32);
    r.push_back(16);
    return r;
  }
  if (log2k % 4) {
    r.push_back(1 << (log2k % 4));
  }
  for (int i = 0; i < log2k / 4; ++i) {
    r.push_back(16);
  }
  return r;
}

int FusedSetOfBlock(int block, int log2k, bool r32) {
  // block digits, most significant first, are the output digits u_0, u_1, ..
  // of the passes before the last; the set index has them least significant first
  const std::vector<int> radices = FusedRadices(log2k, r32);
  int stride = (1 << log2k);
  int set = 0, weight = 1;
  for (std::size_t i = 0; i + 1 < radices.size(); ++i) {
    stride /= radices[i];
    const int digit = (block / (stride / 16)) % radices[i];
    set += digit * weight;
    weight *= radices[i];
  }
  return set;
}

int FusedBlockA(int tau, int log2k, bool r32) {
  const std::vector<int> radices = FusedRadices(log2k, r32);
  const int rl = radices.size() >= 2 ? radices[radices.size() - 2] : 2;
  return (tau / (rl / 2)) * rl + (tau % (rl / 2));
}

bool BuildGeometry(const FilterConfig &config, Geometry *geo, std::string *errorMessage) {
  const std::size_t N = config.fftSize, B = config.blockSize;
  const std::size_t L = config.upsampleFactor > 1 ? config.upsampleFactor : 1;
  if (N < 2 || B == 0 || B >= N || (N & (N - 1)) != 0 || B % L != 0) {
    if (errorMessage) {
      *errorMessage = "invalid filter geometry";
    }
    return false;
  }
  if (N > (static_cast<std::size_t>(1) << 28)) {
    if (errorMessage) {
      *errorMessage = "fft_size too large";
    }
    return false;
  }
  // Frequency-domain polyphase needs the whole time buffer to be an L-fold
  // zero-stuffing of a compact sequence, i.e. L | N (then L | O too because
  // L | B). Otherwise (non power-of-two factors) keep the stuffing in the time
  // domain: P = 1, S = L. Same kernels, same formulas.
  const std::size_t P = (N % L == 0) ? L : 1;
  const std::size_t S = L / P;
  const std::size_t O = N - B;
  geo->P = static_cast<int>(P);
  geo->S = static_cast<int>(S);
  geo->M = static_cast<int>(N / P);
  geo->K = geo->M / 2;
  geo->log2k = Log2(static_cast<std::size_t>(geo->K));
  geo->Oc = static_cast<int>(O / P);
  geo->Bc = static_cast<int>(B / P);
  geo->n_in = static_cast<int>(B / L);
  geo->B = static_cast<int>(B);
  geo->hist_frames = static_cast<int>((static_cast<std::size_t>(geo->Oc) + S - 1) / S);
  geo->Bp = (geo->Bc + 31) & ~31;
  return true;
}

bool BuildTables(const FilterConfig &config, const std::vector<float> &taps, const std::vector<double> *totalFir,
                 int flags, FilterTables *out, std::string *errorMessage) {
  if (!BuildGeometry(config, &out->geo, errorMessage)) {
    return false;
  }
  const Geometry &g = out->geo;
  const std::size_t N = config.fftSize;
  const int P = g.P, M = g.M, K = g.K;
  if (taps.size() != config.taps || taps.size() > N) {
    if (errorMessage) {
      *errorMessage = "taps must be <= fft_size for minimal overlap-save";
    }
    return false;
  }
  if (totalFir && totalFir->size() != taps.size()) {
    if (errorMessage) {
      *errorMessage = "the EQ-folded filter must keep the tap count (fft_size - block_size == taps - 1)";
    }
    return false;
  }

  // ---- total impulse response on the N-point circle -----------------------
  // Without the compat flag it is the FIR itself: `taps` samples, zero beyond -- a linear convolution, EQ included.
  std::vector<double> h(N, 0.0);
  const bool compat = (flags & kLoadRefCompatSpectrum) != 0;
  if (!compat) {
    for (std::size_t i = 0; i < taps.size(); ++i) {
      h[i] = totalFir ? (*totalFir)[i] : static_cast<double>(taps[i]);
    }
  } else {
    // what the reference would multiply by if it were handed these coefficients: its fp32 recurrence FFT of them
    std::vector<std::complex<float>> H32(N, std::complex<float>(0.0f, 0.0f));
    for (std::size_t i = 0; i < taps.size(); ++i) {
      H32[i] = std::complex<float>(totalFir ? static_cast<float>((*totalFir)[i]) : taps[i], 0.0f);
    }
    FftRefCompat32(H32, false);
    std::vector<std::complex<double>> H(N);
    for (std::size_t i = 0; i < N; ++i) {
      H[i] = std::complex<double>(H32[i].real(), H32[i].imag());
    }
    // The reference keeps Re(IFFT(X*H)); for real x that equals filtering with
    // the real part of IFFT(H), whatever asymmetry rounding left in H.
    Fft64(H, true);
    for (std::size_t i = 0; i < N; ++i) {
      h[i] = H[i].real();
    }
  }

  // ---- phase spectra -------------------------------------------------------
  out->Gs.assign(static_cast<std::size_t>(P) * K, cf{0.0f, 0.0f});
  out->Gc.assign(static_cast<std::size_t>(P) * K, cf{0.0f, 0.0f});
  const double scale = 1.0 / (2.0 * static_cast<double>(M));
  std::vector<std::complex<double>> G(M);
  for (int p = 0; p < P; ++p) {
    for (int i = 0; i < M; ++i) {
      G[i] = std::complex<double>(h[static_cast<std::size_t>(i) * P + p], 0.0);
    }
    Fft64(G, false);
    for (int k = 0; k < K; ++k) {
      const std::complex<double> a = G[k] * scale;
      const std::complex<double> b = std::conj(G[K - k]) * scale;
      out->Gs[static_cast<std::size_t>(p) * K + k] = cf{static_cast<float>(a.real()), static_cast<float>(a.imag())};
      out->Gc[static_cast<std::size_t>(p) * K + k] = cf{static_cast<float>(b.real()), static_cast<float>(b.imag())};
    }
  }

  // ---- twiddles --------------------------------------------------------------
  const double pi = 3.14159265358979323846264338327950288;
  out->Wm.resize(K);
  for (int k = 0; k < K; ++k) {
    const double a = -2.0 * pi * static_cast<double>(k) / static_cast<double>(M);
    out->Wm[k] = cf{static_cast<float>(std::cos(a)), static_cast<float>(std::sin(a))};
  }
  out->tw.assign(K > 1 ? K - 1 : 1, cf{1.0f, 0.0f});
  for (int q = 1; q <= g.log2k; ++q) {
    const int half = 1 << (q - 1);
    for (int k = 0; k < half; ++k) {
      const double a = -2.0 * pi * static_cast<double>(k) / static_cast<double>(1 << q);
      out->tw[tw_offset(q) + k] = cf{static_cast<float>(std::cos(a)), static_cast<float>(std::sin(a))};
    }
  }
  out->fusedSplit = false;
  out->fusedNarrow = false;
  out->fusedR32 = false;
  if (((flags & kLoadInternalForceSplit) || out->geo.log2k == 15) && BuildFusedSplitLayout(out)) {
    return true;
  }
  if (out->geo.S == 1 && out->geo.log2k >= 10 && out->geo.log2k <= 14 && (flags & kLoadInternalNarrow)) {
    BuildFusedNarrowLayout(out);  // one butterfly per thread, four waves per SIMD (experiment, see kernel_fused.h)
    return true;
  }
  BuildFusedLayout(out, fused_plan_r32_exists(out->geo.log2k, 2) && (flags & kLoadInternalR32) != 0);
  return true;
}

}  // namespace miups
