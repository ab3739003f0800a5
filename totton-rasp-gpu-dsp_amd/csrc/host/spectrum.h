// Load-time construction of everything the kernels read besides audio:
// geometry, polyphase filter spectra (with the EQ folded in), twiddles.
// Runs on the host in fp64, once per LoadFilter / EQ change
// (reference counterpart: PrepareSpectrum, vulkan_streaming_upsampler.cpp:726-753).
#pragma once

#include <complex>
#include <string>
#include <vector>

#include "../device/common.h"
#include "filter_config.h"

namespace miups {

enum LoadFlags : int {
  kLoadDefault = 0,
  // Build the spectrum from the reference's own fp32 recurrence-twiddle FFT of
  // the taps (what both reference paths multiply by), instead of the exact one.
  kLoadRefCompatSpectrum = 1,
};

struct FilterTables {
  Geometry geo{};
  std::vector<cf> Gs;  // [P][K]  G_p[k] / (2M)
  std::vector<cf> Gc;  // [P][K]  conj(G_p[K-k]) / (2M)   (index 0 holds the Nyquist bin)
  std::vector<cf> Wm;  // [K]     exp(-2 pi i k / M)
  std::vector<cf> tw;  // see tw_offset() in device/fft_radix.h
  // The same spectra in the fused kernel's thread order (FusedTables in
  // device/common.h); empty when the geometry is outside the fused kernel.
  bool hasFused = false;
  // fusedSplit: the layout is for fused_split_kernel<log2k - 1> (two half-length transforms
  // per block transform): T = K/64 threads, GT [P][2][16][T], G0 [P][2][17], selfW [17]. Chosen when
  // K = 32768 (one size past the LDS) and the history length is a multiple of 4.
  bool fusedSplit = false;
  // fusedNarrow: the layout is for fused_kernel<log2k, EXT, 1> (one butterfly per thread, T = K/16 lanes): WmT [T],
  // blockB [T] (each lane's own block), GT [P][8][T], G0 [P]. Experiment (kLoadInternalNarrow), 1024 <= K <= 16384.
  bool fusedNarrow = false;
  // fusedR32: the plain wide layout follows the radix-32 pass plan (K = 8192, 16384: passes K/512, 32, 16) instead of the
  // classic one (2^(log2k mod 4), 16, .., 16): another digit reversal, hence another block <-> set table. The kernel
  // instantiation must match (fused_kernel<.., R32>).
  bool fusedR32 = false;
  std::vector<cf> WmT;      // [T]
  std::vector<int> blockB;  // [T]
  std::vector<f4> GT;       // [P][16][T]
  std::vector<f4> G0;       // [P][17]
  cf Wb{1.0f, 0.0f};
  cf Wself{1.0f, 0.0f};     // plain layout: thread 0's twiddle base for its slots 9..15
  std::vector<cf> selfW;    // split layout only
};

// BuildTables flag for the emulation driver only (tests/emu): build the split layout for
// any K the split kernel covers, so that it can be checked at small sizes.
constexpr int kLoadInternalForceSplit = 0x100;
// ... and: the narrow (one butterfly per thread) layout for 1024 <= K <= 16384 (emulation tests and the
// MIUPS_EXP_NARROW experiment switch; the product uses the wide form, which measured faster: profiles/r02_d_*)
constexpr int kLoadInternalNarrow = 0x200;

// ... and: the radix-32 pass plan for the wide form at K = 8192 / 16384 (MIUPS_EXP_R32 experiment switch on a
// -DMIUPS_WITH_R32 build, emulation tests; measured slower than the classic plan: profiles/r03_b_radix32.txt)
constexpr int kLoadInternalR32 = 0x400;

// Frequency layout of the fused kernel's in-place FFT (decimation in frequency, radices of the pass plan: classic
// R0,16,..,16 or K/512,32,16): after the forward transform LDS block b holds the bins {SetOfBlock(b) + t*K/16}.
// Exposed for the layout tests. r32: the radix-32 plan (see fused_plan_r32_exists in device/common.h).
std::vector<int> FusedRadices(int log2k, bool r32);
int FusedSetOfBlock(int block, int log2k, bool r32);
int FusedBlockA(int tau, int log2k, bool r32);  // first block of thread tau in the pairing passes

bool BuildGeometry(const FilterConfig &config, Geometry *geo, std::string *errorMessage);

// totalFir: optional replacement of the taps by the EQ-folded FIR (fp64, SAME length: eq::FoldCascadeIntoTaps), so the
// product stays a linear convolution with at most `taps` samples and fft_size - block_size == taps - 1 keeps holding.
bool BuildTables(const FilterConfig &config, const std::vector<float> &taps, const std::vector<double> *totalFir,
                 int flags, FilterTables *out, std::string *errorMessage);

}  // namespace miups
