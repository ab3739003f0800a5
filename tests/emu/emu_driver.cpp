// TEST INFRASTRUCTURE ONLY -- CPU thread-emulation of the HIP kernels.
//
// Compiles the product's device headers (csrc/device/*.h) with
// -DMIUPS_HOST_EMU, where blockIdx/threadIdx/__syncthreads/dynamic LDS are
// provided by the shim below (one OS thread per GPU thread, one workgroup at a
// time), and runs them under AddressSanitizer/UBSan on exact-size buffers.
// Purpose: catch indexing and synchronisation mistakes on the CPU before a
// kernel is launched on a GPU box (a faulting kernel can take the node down).
// It is a separate executable under tests/; libmi_upsampler.so does not contain
// it and has no code path to it.
//
// usage: emu_driver <filter.json> <flags> <streams> <channels> <in_fmt> <out_fmt>
//                   <blocks_per_call> <calls> <in.bin> <out.bin> <fused|staged|auto>
//   in.bin : [call][stream][frame][channel] samples of in_fmt
//   out.bin: [call][stream][frame][channel] samples of out_fmt
#include <pthread.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "device/kernels_generic.h"
#include "device/kernels_tiled.h"
#include "host/filter_config.h"
#include "host/spectrum.h"

#include "emu_launch.h"

namespace miups_emu {
thread_local Dim3 t_threadIdx;
thread_local Dim3 t_blockIdx;
Dim3 g_blockDim;
Dim3 g_gridDim;
static pthread_barrier_t g_bar;
static bool g_use_bar = false;
static std::vector<char> g_shared;

void barrier() {
  if (g_use_bar) {
    pthread_barrier_wait(&g_bar);
  }
}
void *dyn_shared() { return g_shared.data(); }

// cross-lane exchange of the narrow kernel (v_permlane32_swap on the GPU): every thread of the block calls it
static std::vector<float> g_xch;
float xchg_xor(float v, unsigned mask) {
  const unsigned me = t_threadIdx.x;
  g_xch[me] = v;
  barrier();
  const float r = g_xch[me ^ mask];
  barrier();
  return r;
}

// run f() once per (block, thread); with barriers every thread of a block is a
// real OS thread, otherwise threads run back to back.
void launch(unsigned grid, unsigned block, size_t shmem, bool barriers, const std::function<void()> &f) {
  g_gridDim.x = grid;
  g_blockDim.x = block;
  g_shared.assign(shmem, 0);
  g_xch.assign(block, 0.0f);
  g_use_bar = barriers && block > 1;
  for (unsigned b = 0; b < grid; ++b) {
    if (!g_use_bar) {
      for (unsigned t = 0; t < block; ++t) {
        t_blockIdx.x = b;
        t_threadIdx.x = t;
        f();
      }
      continue;
    }
    // one OS thread per GPU thread, kept for ALL workgroups of the launch (a thread per (workgroup, thread) spent most of
    // the suite's time in clone/join): every thread walks the workgroups in order, a barrier between two workgroups
    pthread_barrier_init(&g_bar, nullptr, block);
    std::vector<std::thread> threads;
    threads.reserve(block);
    for (unsigned t = 0; t < block; ++t) {
      threads.emplace_back([&, t]() {
        for (unsigned wg = 0; wg < grid; ++wg) {
          t_blockIdx.x = wg;
          t_threadIdx.x = t;
          f();
          pthread_barrier_wait(&g_bar);  // the workgroup is done: its LDS may be reused
        }
      });
    }
    for (auto &th : threads) {
      th.join();
    }
    pthread_barrier_destroy(&g_bar);
    break;
  }
}
}  // namespace miups_emu

using namespace miups;

namespace {

unsigned Blocks(long long total, int threads) { return static_cast<unsigned>((total + threads - 1) / threads); }

template <int DIR>
void EmuPass(int R, const cf *in, cf *out, const cf *tw, int K, int Ns, int log2NsR, long long rows) {
  const unsigned grid = Blocks(rows * (K / R), 64);
  auto run = [&](auto kernel) { miups_emu::launch(grid, 64, 0, false, [&]() { kernel(in, out, tw, K, Ns, log2NsR, rows); }); };
  switch (R) {
    case 2: run(gen_pass_kernel<DIR, 2>); break;
    case 4: run(gen_pass_kernel<DIR, 4>); break;
    case 8: run(gen_pass_kernel<DIR, 8>); break;
    default: run(gen_pass_kernel<DIR, 16>); break;
  }
}

template <int DIR>
cf *EmuFft(cf *a, cf *b, const cf *tw, int log2k, long long rows) {
  const int K = 1 << log2k;
  int done = 0;
  cf *src = a, *dst = b;
  auto pass = [&](int log2r) {
    EmuPass<DIR>(1 << log2r, src, dst, tw, K, 1 << done, done + log2r, rows);
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

// interleave_tiled_kernel with the engine's 256-thread shape (EPT = rows * TI / 1024)
template <int FMT, int TI>
void EmuInterleaveTiledEpt(const Geometry &g, const IoDesc &io, const float *planes, int sb0, int nb, int tiles, int ept) {
  const int rows = g.P * io.channels;
  const unsigned grid = static_cast<unsigned>(nb) * static_cast<unsigned>(tiles);
  const size_t lds = static_cast<size_t>(rows) * (TI + 1) * sizeof(float);
  auto run = [&](auto kernel) { miups_emu::launch(grid, 256, lds, true, [&]() { kernel(g, io, planes, sb0, nb, tiles); }); };
  switch (ept) {
    case 1: run(interleave_tiled_kernel<FMT, TI, 1>); break;
    case 2: run(interleave_tiled_kernel<FMT, TI, 2>); break;
    case 3: run(interleave_tiled_kernel<FMT, TI, 3>); break;
    case 4: run(interleave_tiled_kernel<FMT, TI, 4>); break;
    case 5: run(interleave_tiled_kernel<FMT, TI, 5>); break;
    case 6: run(interleave_tiled_kernel<FMT, TI, 6>); break;
    case 7: run(interleave_tiled_kernel<FMT, TI, 7>); break;
    default: run(interleave_tiled_kernel<FMT, TI, 8>); break;
  }
}
void EmuInterleaveTiled(const Geometry &g, const IoDesc &io, const float *planes, int sb0, int nb, int tiles, int ti, int ept,
                        bool f32) {
  if (f32) {
    if (ti == 64) EmuInterleaveTiledEpt<kF32, 64>(g, io, planes, sb0, nb, tiles, ept);
    else if (ti == 32) EmuInterleaveTiledEpt<kF32, 32>(g, io, planes, sb0, nb, tiles, ept);
    else EmuInterleaveTiledEpt<kF32, 16>(g, io, planes, sb0, nb, tiles, ept);
  } else {
    if (ti == 64) EmuInterleaveTiledEpt<kS32, 64>(g, io, planes, sb0, nb, tiles, ept);
    else if (ti == 32) EmuInterleaveTiledEpt<kS32, 32>(g, io, planes, sb0, nb, tiles, ept);
    else EmuInterleaveTiledEpt<kS32, 16>(g, io, planes, sb0, nb, tiles, ept);
  }
}

bool DispatchFused(const Geometry &g, const IoDesc &io, const FilterTables &t, unsigned items) {
  if (t.fusedSplit) {  // two half-length transforms per block transform
    switch (g.log2k - 1) {
      case 10: EmuFusedSplitK10(g, io, t, items); return true;
      case 11: EmuFusedSplitK11(g, io, t, items); return true;
      case 12: EmuFusedSplitK12(g, io, t, items); return true;
      case 13: EmuFusedSplitK13(g, io, t, items); return true;
      case 14: EmuFusedSplitK14(g, io, t, items); return true;
      default: return false;
    }
  }
  switch (g.log2k) {
    case 5: EmuFusedK5(g, io, t, items); return true;
    case 6: EmuFusedK6(g, io, t, items); return true;
    case 7: EmuFusedK7(g, io, t, items); return true;
    case 8: EmuFusedK8(g, io, t, items); return true;
    case 9: EmuFusedK9(g, io, t, items); return true;
    case 10: EmuFusedK10(g, io, t, items); return true;
    case 11: EmuFusedK11(g, io, t, items); return true;
    case 12: EmuFusedK12(g, io, t, items); return true;
    case 13: EmuFusedK13(g, io, t, items); return true;
    case 14: EmuFusedK14(g, io, t, items); return true;
    default: return false;
  }
}

std::vector<char> ReadAll(const std::string &path) {
  std::ifstream f(path, std::ios::binary);
  return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

}  // namespace

int main(int argc, char **argv) {
  if (argc != 12) {
    std::cerr << "bad usage\n";
    return 2;
  }
  const std::string json = argv[1];
  int flags = std::atoi(argv[2]);
  if (std::getenv("EMU_SPLIT")) {  // tests: the split layout at sizes the emulation can run
    flags |= kLoadInternalForceSplit;
  }
  if (std::getenv("EMU_NARROW")) {  // tests: one butterfly per thread (experiment form) where it exists
    flags |= kLoadInternalNarrow;
  }
  if (std::getenv("EMU_R32")) {  // tests: the radix-32 pass plan (experiment) at K = 8192 / 16384
    flags |= kLoadInternalR32;
  }
  const int streams = std::atoi(argv[3]), channels = std::atoi(argv[4]);
  const int inFmt = std::atoi(argv[5]), outFmt = std::atoi(argv[6]);
  const int blocks = std::atoi(argv[7]), calls = std::atoi(argv[8]);
  const std::string inPath = argv[9], outPath = argv[10], path = argv[11];

  FilterConfig config;
  std::vector<float> taps;
  std::string error;
  if (!ReadFilter(json, &config, &taps, &error)) {
    std::cerr << error << "\n";
    return 1;
  }
  FilterTables t;
  if (!BuildTables(config, taps, nullptr, flags, &t, &error)) {
    std::cerr << error << "\n";
    return 1;
  }
  const Geometry g = t.geo;
  const bool fusedOk = t.hasFused;
  if (std::getenv("EMU_SPLIT") && !t.fusedSplit) {
    std::cerr << "split layout not available for this geometry\n";
    return 3;
  }
  const bool tiled = path == "tiled";  // the two-level path (device/kernels_tiled.h)
  if (tiled && !tiled_covers(g.log2k)) {
    std::cerr << "two-level path does not cover this geometry\n";
    return 3;
  }
  const bool fused = path == "fused" ? true : ((path == "staged" || tiled) ? false : fusedOk);
  if (fused && !fusedOk) {
    std::cerr << "fused path does not cover this geometry\n";
    return 3;
  }

  const size_t inRow = static_cast<size_t>(blocks) * g.n_in * channels * pcm_bytes(inFmt);
  const size_t outRow = static_cast<size_t>(blocks) * g.B * channels * pcm_bytes(outFmt);
  const std::vector<char> all = ReadAll(inPath);
  if (all.size() != inRow * streams * calls) {
    std::cerr << "input size mismatch: " << all.size() << " vs " << inRow * streams * calls << "\n";
    return 1;
  }
  const size_t histRow = static_cast<size_t>(g.hist_frames) * channels * pcm_bytes(inFmt);
  std::vector<char> hist(histRow * streams, 0), hist2(histRow * streams, 0);
  std::ofstream out(outPath, std::ios::binary | std::ios::trunc);

  for (int call = 0; call < calls; ++call) {
    // exact-size copies so ASan sees every out-of-range access
    std::vector<char> in(all.begin() + call * inRow * streams, all.begin() + (call + 1) * inRow * streams);
    std::vector<char> o(outRow * streams, 0);
    IoDesc io{};
    io.in = in.data();
    io.hist = hist.data();
    io.out = o.data();
    io.in_stream_stride = static_cast<long long>(inRow);
    io.hist_stream_stride = static_cast<long long>(histRow);
    io.out_stream_stride = static_cast<long long>(outRow);
    io.channels = channels;
    io.streams = streams;
    io.in_fmt = inFmt;
    io.out_fmt = outFmt;
    io.blocks = blocks;
    const unsigned items = static_cast<unsigned>(blocks) * streams * channels;
    if (fused) {
      // same rule as Engine::PickChannelGroup / ProcessDevice: whole frames per workgroup for
      // mono/stereo, else one channel per workgroup and the frames written by interleave_*_kernel
      int cg = channels <= 2 ? channels : 1;
      if (const char *force = std::getenv("EMU_CG")) {  // tests: other group widths the kernel supports
        const int v = std::atoi(force);
        if (v > 0 && channels % v == 0) {
          cg = v;
        }
      }
      // EMU_PARTS=d: the engine's small-call form -- d workgroups per (block, stream, channel), P / d phases each
      int parts = 0;
      if (const char *pp = std::getenv("EMU_PARTS")) {
        parts = std::atoi(pp);
        if (parts < 2 || (t.fusedSplit ? 2 * g.P : g.P) % parts != 0 || t.fusedNarrow || t.fusedR32 ||
            (t.fusedSplit && std::getenv("EMU_PARK"))) {
          std::fprintf(stderr, "EMU_PARTS: %d does not fit this geometry\n", parts);
          return 2;
        }
        cg = 1;
      }
      // EMU_INKERNEL=1: keep the in-kernel epilogue for groups narrower than a frame
      const bool ext = t.fusedSplit || parts > 0 || (cg < channels && !std::getenv("EMU_INKERNEL"));
      const unsigned groups = static_cast<unsigned>(channels / cg);
      const unsigned pairs = static_cast<unsigned>(blocks) * streams;
      const unsigned chunk = pairs > 1 ? (pairs + 1) / 2 : pairs;  // exercise the chunked launch (item0 > 0)
      std::vector<float> scratch(static_cast<size_t>(chunk) * channels * g.P * g.Bp);
      io.scratch = scratch.data();
      io.cg = cg;
      io.groups = channels / cg;
      io.out_vec_ok = (reinterpret_cast<uintptr_t>(o.data()) % 16 == 0 && outRow % 16 == 0 &&
                       (static_cast<size_t>(g.B) * channels * 4) % 16 == 0)
                          ? 1
                          : 0;
      IoDesc ioF = io;
      std::vector<float> planar;
      // same rules as Engine::ProcessDevice: de-interleave wide frames first; the split form from a split-planar timeline
      const bool splitPlanar = t.fusedSplit && g.Bc % 4 == 0 && g.hist_frames == g.Oc && g.Oc % 4 == 0 &&
                               std::getenv("EMU_NO_SPLIT_PLANAR") == nullptr;
      if (channels > 2 || splitPlanar) {
        const long long total = static_cast<long long>(g.hist_frames) + static_cast<long long>(blocks) * g.n_in;
        const long long planeFloats = (total + 3) / 4 * 4;
        planar.assign(static_cast<size_t>(planeFloats) * channels * streams, 0.0f);
        const int tileFrames = planar_tile_frames(channels);
        const int tiles = static_cast<int>((total + tileFrames - 1) / tileFrames);
        IoDesc ioP = io;
        ioP.split_planes = splitPlanar ? 1 : 0;
        miups_emu::launch(static_cast<unsigned>(tiles) * streams, 256,
                          static_cast<size_t>(channels) * (tileFrames + 1) * sizeof(float), true,
                          [&]() { planarize_kernel(g, ioP, planar.data(), planeFloats, total, tiles, tileFrames); });
        ioF.in = planar.data();
        ioF.in_fmt = kF32;
        ioF.in_planar = splitPlanar ? 2 : 1;
        ioF.in_plane_stride = planeFloats * static_cast<long long>(sizeof(float));
        ioF.in_stream_stride = ioF.in_plane_stride * channels;
      }
      ioF.ext_epilogue = ext ? 1 : 0;
      ioF.phase_parts = parts;
      ioF.split_planes = t.fusedSplit ? 1 : 0;
      std::vector<f4> park;
      ioF.park = nullptr;
      if (t.fusedSplit && std::getenv("EMU_PARK")) {  // same rule as the engine (MIUPS_EXP_PARK: experiment, off by default)
        park.assign(static_cast<size_t>(chunk) * groups * split_park_words(g.K / 64), f4{0.0f, 0.0f, 0.0f, 0.0f});
        ioF.park = park.data();
      }
      const bool quad = ioF.out_vec_ok && (outFmt == kF32 || outFmt == kS32) && (g.P * channels) % 4 == 0 && g.Bc % 4 == 0;
      // EMU_COOP=1: cooperative frames, same rule as Engine::ProcessDevice (device/frame_tile.h). The emulation runs the
      // workgroups of a launch one after the other, so a pair's last workgroup finds it complete and assembles up to its
      // cap; later workgroups take older pairs' tiles; the frame pass behind the kernel takes the rest.
      int coopTi = 0, coopEpt = 0;
      std::vector<FrameSync> fsync;
      if (std::getenv("EMU_COOP") && ext && !t.fusedSplit && parts == 0 && !t.fusedNarrow && !t.fusedR32 && quad &&
          !std::getenv("EMU_NO_TILED_INTERLEAVE")) {
        const int rows = g.P * channels, T = g.K / 32;
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
        if (!coopTi) {
          std::fprintf(stderr, "EMU_COOP: no tile width fits this shape\n");
          return 2;
        }
        fsync.resize(chunk);
        const int tiles = (g.Bc + coopTi - 1) / coopTi;
        int cap = 2 * ((tiles + static_cast<int>(groups) - 1) / static_cast<int>(groups));
        if (const char *c = std::getenv("EMU_COOP_CAP")) {
          cap = std::atoi(c);
        }
        ioF.fsync = fsync.data();
        ioF.ftile_ti = coopTi;
        ioF.ftile_ept = coopEpt;
        ioF.ftiles = tiles;
        ioF.ftile_cap = std::max(1, cap);
      }
      for (unsigned p0 = 0; p0 < pairs; p0 += chunk) {
        const unsigned np = std::min(chunk, pairs - p0);
        ioF.item0 = static_cast<int>(p0 * groups);
        if (coopTi) {
          std::fill(fsync.begin(), fsync.end(), FrameSync{});
        }
        DispatchFused(g, ioF, t, np * groups * static_cast<unsigned>(parts ? parts : 1));
        if (!ext) {
          continue;
        }
        if (coopTi) {  // how much of the frame work the transform kernel's workgroups took (stderr: the test reads it)
          unsigned long long claimed = 0;
          const unsigned tiles = static_cast<unsigned>(ioF.ftiles);
          for (unsigned j = 0; j < np; ++j) {
            claimed += std::min(fsync[j].next, tiles);
          }
          std::fprintf(stderr, "EMU_COOP: %llu of %llu tiles assembled inside the transform kernel\n", claimed,
                       static_cast<unsigned long long>(np) * tiles);
        }
        const int rows = g.P * channels;
        int tiledTi = 0;  // same rule as the engine
        if (quad && !t.fusedSplit && rows >= 16 && !std::getenv("EMU_NO_TILED_INTERLEAVE")) {
          tiledTi = rows <= 128 ? 64 : (rows <= 256 ? 32 : 16);
          if (coopTi) {
            tiledTi = coopTi;
          }
          const int per = 1024 / tiledTi;
          if (rows % per != 0 || rows / per > 8 || rows > 512) {
            tiledTi = 0;
          }
        }
        if (tiledTi) {
          const int tiles = (g.Bc + tiledTi - 1) / tiledTi;
          EmuInterleaveTiled(g, ioF, scratch.data(), static_cast<int>(p0), static_cast<int>(np), tiles, tiledTi,
                             rows * tiledTi / 1024, outFmt == kF32);
        } else if (quad && (rows == 4 || rows == 8) && !std::getenv("EMU_NO_ROWS_INTERLEAVE")) {
          const int threads = 32, perWg = threads * (32 / rows);
          const int wgsPerPair = (g.Bc + perWg - 1) / perWg;
          auto run = [&](auto kernel) {
            miups_emu::launch(np * wgsPerPair, threads, 0, false, [&]() {
              kernel(g, ioF, scratch.data(), static_cast<int>(p0), static_cast<int>(np), wgsPerPair);
            });
          };
          if (outFmt == kF32) {
            if (rows == 4) run(interleave_rows_kernel<kF32, 4>);
            else run(interleave_rows_kernel<kF32, 8>);
          } else {
            if (rows == 4) run(interleave_rows_kernel<kS32, 4>);
            else run(interleave_rows_kernel<kS32, 8>);
          }
        } else if (quad) {
          const int threads = 32, perWg = threads * 4;
          const long long units = static_cast<long long>(g.Bc / 4) * (g.P * channels / 4);
          const int wgsPerPair = static_cast<int>((units + perWg - 1) / perWg);
          auto run = [&](auto kernel) {
            miups_emu::launch(np * wgsPerPair, threads, 0, false, [&]() {
              kernel(g, ioF, scratch.data(), static_cast<int>(p0), static_cast<int>(np), wgsPerPair);
            });
          };
          if (outFmt == kF32) {
            run(interleave_quad_kernel<kF32>);
          } else {
            run(interleave_quad_kernel<kS32>);
          }
        } else {
          const long long total = static_cast<long long>(np) * g.B * channels;
          miups_emu::launch(Blocks(total, 64), 64, 0, false, [&]() {
            interleave_scalar_kernel(g, ioF, scratch.data(), static_cast<int>(p0), static_cast<int>(np));
          });
        }
      }
    } else if (tiled) {
      // same sequence as Engine::ProcessDevice's two-level branch, in two chunks of pairs (item0 > 0)
      const int K1 = 1 << tiled_log2k1(g.log2k), M2 = g.K / K1;
      std::vector<f4> tGsc(t.Gs.size());  // {Gs, Gc} per bin and phase, as DeviceFilter::StageTables builds them
      std::vector<cf> tWm(t.Wm.size());
      for (int k1 = 0; k1 < K1; ++k1) {
        for (int k2 = 0; k2 < M2; ++k2) {
          const size_t to = static_cast<size_t>(k1) * M2 + k2, from = static_cast<size_t>(k1) + static_cast<size_t>(K1) * k2;
          tWm[to] = t.Wm[from];
          for (int p = 0; p < g.P; ++p) {
            const cf a = t.Gs[static_cast<size_t>(p) * g.K + from], b = t.Gc[static_cast<size_t>(p) * g.K + from];
            tGsc[static_cast<size_t>(p) * g.K + to] = f4{a.x, a.y, b.x, b.y};
          }
        }
      }
      const unsigned pairs = static_cast<unsigned>(blocks) * streams;
      const unsigned chunk = pairs > 1 ? (pairs + 1) / 2 : pairs;
      const size_t nmax = static_cast<size_t>(chunk) * channels;
      std::vector<cf> A(nmax * g.K), X(nmax * g.K), Bw(nmax * g.K * g.P);
      std::vector<float> planes(nmax * g.P * g.Bp);
      IoDesc ioT = io;
      ioT.scratch = planes.data();
      ioT.cg = 1;
      ioT.groups = channels;
      ioT.ext_epilogue = 1;
      ioT.out_vec_ok = (reinterpret_cast<uintptr_t>(o.data()) % 16 == 0 && outRow % 16 == 0 &&
                        (static_cast<size_t>(g.B) * channels * 4) % 16 == 0)
                           ? 1
                           : 0;
      // the engine's rule: interleaved frames (two or more channels, S = 1) go through planarize_kernel first
      IoDesc ioL = io;
      std::vector<float> planarT;
      if (channels >= 2 && g.S == 1 && g.hist_frames == g.Oc && !std::getenv("EMU_TILED_NO_PLANAR")) {
        const long long total = static_cast<long long>(g.hist_frames) + static_cast<long long>(blocks) * g.n_in;
        const long long planeFloats = (total + 3) / 4 * 4;
        planarT.assign(static_cast<size_t>(planeFloats) * channels * streams, 0.0f);
        const int tileFrames = planar_tile_frames(channels);
        const int tiles = static_cast<int>((total + tileFrames - 1) / tileFrames);
        IoDesc ioP = io;
        ioP.split_planes = 0;
        miups_emu::launch(static_cast<unsigned>(tiles) * streams, 256,
                          static_cast<size_t>(channels) * (tileFrames + 1) * sizeof(float), true,
                          [&]() { planarize_kernel(g, ioP, planarT.data(), planeFloats, total, tiles, tileFrames); });
        ioL.in = planarT.data();
        ioL.in_fmt = kF32;
        ioL.in_planar = 1;
        ioL.in_plane_stride = planeFloats * static_cast<long long>(sizeof(float));
        ioL.in_stream_stride = ioL.in_plane_stride * channels;
      }
      auto rows_kernels = [&](auto cfgTag, auto k1Tag, int item0, int n) {
        constexpr int LOG2M = decltype(cfgTag)::value, KK1 = decltype(k1Tag)::value;
        using Cfg = TiledRowCfg<LOG2M>;
        miups_emu::launch(Blocks(static_cast<long long>(n) * M2, 64), 64, 0, false,
                          [&]() { tiled_load_kernel<KK1>(g, ioL, t.tw.data(), A.data(), item0, n); });
        TiledRowSrc plain{A.data(), nullptr, nullptr};
        miups_emu::launch(static_cast<unsigned>(n) * KK1, Cfg::T, Cfg::LDS_BYTES, true,
                          [&]() { tiled_row_forward_kernel<LOG2M, KK1>(g, plain, t.tw.data(), X.data()); });
        TiledRowSrc spectral{X.data(), tWm.data(), tGsc.data()};
        miups_emu::launch(static_cast<unsigned>(n) * g.P * KK1, Cfg::T, Cfg::LDS_BYTES, true,
                          [&]() { tiled_row_inverse_kernel<LOG2M, KK1>(g, spectral, t.tw.data(), Bw.data()); });
        const long long rows = static_cast<long long>(n) * g.P;
        miups_emu::launch(Blocks(rows * M2, 64), 64, 0, false,
                          [&]() { tiled_store_kernel<KK1>(g, t.tw.data(), Bw.data(), planes.data(), rows, 1); });
      };
      for (unsigned p0 = 0; p0 < pairs; p0 += chunk) {
        const unsigned np = std::min(chunk, pairs - p0);
        const int item0 = static_cast<int>(p0) * channels, n = static_cast<int>(np) * channels;
        switch (g.log2k) {
          case 15: rows_kernels(std::integral_constant<int, 11>(), std::integral_constant<int, 16>(), item0, n); break;
          case 16: rows_kernels(std::integral_constant<int, 12>(), std::integral_constant<int, 16>(), item0, n); break;
          case 17: rows_kernels(std::integral_constant<int, 13>(), std::integral_constant<int, 16>(), item0, n); break;
          default: rows_kernels(std::integral_constant<int, 13>(), std::integral_constant<int, 32>(), item0, n); break;
        }
        const long long total = static_cast<long long>(np) * g.B * channels;
        miups_emu::launch(Blocks(total, 64), 64, 0, false, [&]() {
          interleave_scalar_kernel(g, ioT, planes.data(), static_cast<int>(p0), static_cast<int>(np));
        });
      }
    } else {
      const size_t row = g.K;
      std::vector<cf> w0(items * row), w1(items * row), w2(items * row * g.P), w3(items * row * g.P);
      const long long elems = static_cast<long long>(items) * g.K;
      miups_emu::launch(Blocks(elems, 64), 64, 0, false, [&]() { gen_load_kernel(g, io, w0.data(), 0, items); });
      cf *Z = EmuFft<-1>(w0.data(), w1.data(), t.tw.data(), g.log2k, items);
      miups_emu::launch(Blocks(elems, 64), 64, 0, false, [&]() {
        gen_multiply_kernel(g, Z, w2.data(), t.Gs.data(), t.Gc.data(), t.Wm.data(), items);
      });
      cf *y = EmuFft<+1>(w2.data(), w3.data(), t.tw.data(), g.log2k, static_cast<long long>(items) * g.P);
      miups_emu::launch(Blocks(elems * g.P, 64), 64, 0, false, [&]() { gen_store_kernel(g, io, y, 0, items); });
    }
    const long long histBytes = static_cast<long long>(histRow) * streams;
    if (histBytes > 0) {
      miups_emu::launch(Blocks(histBytes, 64), 64, 0, false, [&]() {
        update_history_kernel(g, io, hist2.data(), static_cast<long long>(blocks) * g.n_in);
      });
      hist.swap(hist2);
    }
    out.write(o.data(), static_cast<std::streamsize>(o.size()));
  }
  return 0;
}
