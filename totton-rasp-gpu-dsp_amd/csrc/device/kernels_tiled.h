// Two-level path for transforms past the fused kernels: K = K1 * M2 with the M2-point transforms in LDS.
//
// The any-size staged path (kernels_generic.h) runs every radix-16 pass of a K-point transform as a launch of its own
// over HBM: 4-5 round trips per transform, (1 + P) transforms per channel-block, plus one pass each to load, multiply and
// store. The "2m" filters the selector prefers when present (640 001 taps, N = 2^20: alsa_filter_selector.cpp:74-96) land
// there at 2x / 4x / 8x (K = 262144 / 131072 / 65536) and ran at 23-29 Gsamples/s against 120-230 on the fused kernels
// (scripts/staged_rate.py). Here the transform is the classic four-step form, bin k = k1 + K1 k2, sample n = n1 M2 + n2:
//
//   forward   A[k1][n2] = W_K^(n2 k1) * sum_n1 z[n1 M2 + n2] W_K1^(n1 k1)      tiled_load_kernel  (radix-K1 in registers,
//                                                                               straight from the PCM frames)
//             X[k1][k2] = sum_n2 A[k1][n2] W_M2^(n2 k2)                         tiled_row_kernel   (M2 points in LDS)
//   per phase Z'[k1][k2] = spectral_bin(X[k], conj X[K-k], ...)                 fused into the load of ...
//             B[k1][n2]  = sum_k2 Z'[k1][k2] W_M2^(-n2 k2)                       tiled_row_kernel   (inverse)
//             y[n1 M2 + n2] = sum_k1 B[k1][n2] W_K^(-n2 k1) W_K1^(-n1 k1)        tiled_store_kernel (radix-K1, kept samples
//                                                                               phase-planar into the staging planes)
//   frames    interleave_*_kernel of the fused path.
//
// The spectrum stays in the [k1][k2] order between the two row transforms (a pointwise stage does not care; the mirror
// bin K - k = (K1 - k1) + K1 (M2 - 1 - k2) is a row read backwards), so every global access of every kernel is
// lane-contiguous and a transform costs two round trips instead of four or five; the spectral product is never written.
// Row transforms: Stockham autosort passes (the staged path's own index algebra, gen_pass_kernel) through ONE LDS buffer
// (a pass reads, waits for every thread to have read, then writes), first pass from global memory, last pass to global
// memory; LDS word indices are XOR-swizzled (tiled_swz) so that the first pass's stride-R stores spread over the banks.
//
// K1 = 16 (K = 32768 .. 131072: M2 = 2048 / 4096 / 8192) or 32 (K = 262144: M2 = 8192).
#pragma once

#include "kernels_generic.h"

namespace miups {

MI_HD constexpr bool tiled_covers(int log2k) { return log2k >= 15 && log2k <= 18; }
MI_HD constexpr int tiled_log2k1(int log2k) { return log2k == 18 ? 5 : 4; }
MI_HD constexpr int tiled_log2m2(int log2k) { return log2k - tiled_log2k1(log2k); }

// work item of this path: item = (stream * blocks + block) * channels + channel -- the order of the staging planes
// ([pair][channel][phase][Bp]) that the interleave kernels read
MI_DEVICE void tiled_item(const IoDesc &io, int item, int &s, int &blk, int &c) {
  const int sb = item / io.channels;
  c = item - sb * io.channels;
  s = sb / io.blocks;
  blk = sb - s * io.blocks;
}

// ---- forward column pass, straight from the PCM frames ------------------------------------------------------------
// thread = (item, n2); A[item][k1][n2]
template <int K1>
MI_GLOBAL void tiled_load_kernel(Geometry g, IoDesc io, const cf *MI_RESTRICT tw, cf *MI_RESTRICT A, int item0, int nitems) {
  const int M2 = g.K / K1;
  const long long gid = static_cast<long long>(MI_BID_X) * MI_BDIM_X + MI_TID_X;
  if (gid >= static_cast<long long>(nitems) * M2) {
    return;
  }
  const int it = static_cast<int>(gid / M2);
  const int n2 = static_cast<int>(gid - static_cast<long long>(it) * M2);
  int s, blk, c;
  tiled_item(io, item0 + it, s, blk, c);
  cf v[K1];
  if (io.in_planar == 1) {
    // one fp32 timeline per channel (planarize_kernel; S = 1 and hist_frames = Oc): compact sample n of block blk is
    // timeline sample blk * Bc + n -- lanes read consecutive 8-byte words
    const float *tl = reinterpret_cast<const float *>(static_cast<const char *>(io.in) + s * io.in_stream_stride +
                                                      c * io.in_plane_stride) +
                      static_cast<long long>(blk) * g.Bc;
    MI_UNROLL
    for (int n1 = 0; n1 < K1; ++n1) {
      const int n = 2 * (n1 * M2 + n2);
      v[n1] = mk(tl[n], tl[n + 1]);
    }
  } else {
    MI_UNROLL
    for (int n1 = 0; n1 < K1; ++n1) {
      const int n = n1 * M2 + n2;  // complex word of the compact block: samples 2n, 2n + 1
      v[n1] = mk(compact_sample(g, io, s, c, blk, 2 * n), compact_sample(g, io, s, c, blk, 2 * n + 1));
    }
  }
  dftR<-1, K1>(v);
  if (n2 > 0) {
    apply_twiddles_out<-1, K1>(v, tw[tw_offset(g.log2k) + n2]);  // W_K^(n2 k1); n2 < M2 <= K/2
  }
  cf *dst = A + static_cast<long long>(it) * g.K + n2;
  MI_UNROLL
  for (int k1 = 0; k1 < K1; ++k1) {
    dst[static_cast<long long>(k1) * M2] = v[out_pos<K1>(k1)];
  }
}

// ---- row transforms in LDS ------------------------------------------------------------------------------------------
// LDS word index of row element i: the low four bits XORed with the next four -- the first pass's stride-R stores (lane j
// writes words 16 j + u) and the later passes' 16-word groups both spread over all banks, and the buffer stays M2 words
// (a padded layout, i + i/16, took 34.8 KB at M2 = 4096: four workgroups per CU instead of five)
MI_HD constexpr int tiled_swz(int i) { return i ^ ((i >> 4) & 15); }
template <int LOG2M>
struct TiledRowCfg {
  static constexpr int M = 1 << LOG2M;
  static constexpr int T = M / 16;               // threads: sixteen points each in every pass
  static constexpr int R0 = 1 << (LOG2M % 4);    // first pass radix (1 = none: 4096 = 16^3)
  static constexpr int NPASS = LOG2M / 4 + (R0 > 1 ? 1 : 0);
  static constexpr int BUF_WORDS = M;            // swizzled, not padded (tiled_swz)
  // ONE buffer: a pass between LDS and LDS reads its sixteen words, waits for every thread to have read, then writes
  static constexpr int LDS_BYTES = BUF_WORDS * 8;
};

// What a row transform reads. Plain: a row of A. Spectral: the spectral stage of phase p on the fly (gen_multiply_kernel's
// arithmetic on the [k1][k2] layout; tables in the same layout).
struct TiledRowSrc {
  const cf *rows;   // plain: A; spectral: X   (both [item][K1][M2])
  const cf *WmT;    // spectral: [K1][M2]   W_M^k, k = k1 + K1 k2
  const f4 *GscT;   // spectral: [P][K1][M2] {Gs, Gc} of the bin as ONE 16-byte word (the L2 -> CU path is bound per load
                    // instruction, not per byte: scripts/ubench/load_width.hip -- four operand streams instead of five)
};

template <int LOG2M, int DIR, bool SPECTRAL, int K1>
struct TiledRow {
  using Cfg = TiledRowCfg<LOG2M>;
  static constexpr int M = Cfg::M, T = Cfg::T, R0 = Cfg::R0;

  // this workgroup's rows, as workgroup-uniform pointers (scalar registers): the per-element address is then one 32-bit
  // offset instead of (it * K1 + k1) * M + i in 64 bits per element (fewer instructions; the kernel's time did not move:
  // profiles/r03_q_two_level.txt)
  struct Rows {
    const cf *x;    // plain: the row of A; spectral: the row of X
    const cf *xm;   // spectral: the mirror row of X (read backwards)
    const cf *w;    // spectral: untangle twiddles of the row
    const f4 *gsc;  // spectral: the phase's {Gs, Gc} words of the row
    int k1;
  };
  static MI_DEVICE Rows make_rows(const Geometry &g, const TiledRowSrc &src, long long it, int p, int k1) {
    Rows r{};
    r.k1 = k1;
    r.x = src.rows + (it * K1 + k1) * M;
    if constexpr (SPECTRAL) {
      const int m1 = (K1 - k1) & (K1 - 1);
      r.xm = src.rows + (it * K1 + m1) * M;
      r.w = src.WmT + static_cast<long long>(k1) * M;
      r.gsc = src.GscT + static_cast<long long>(p) * g.K + static_cast<long long>(k1) * M;
    }
    return r;
  }
  // element i of this workgroup's row
  static MI_DEVICE cf fetch(const Rows &r, int i) {
    if constexpr (!SPECTRAL) {
      return r.x[i];
    } else {
      const int m2 = r.k1 ? M - 1 - i : ((M - i) & (M - 1));
      const f4 gg = r.gsc[i];
      return spectral_bin(r.x[i], r.xm[m2], r.w[i], mk(gg.x, gg.y), mk(gg.z, gg.w));
    }
  }

  // one Stockham pass, radix R, Ns = product of the radices before it: butterfly j reads j + t*M/R, writes
  // (j-k)*R + k + u*Ns with k = j mod Ns (gen_pass_kernel's statement).
  // FROM_GLOBAL: inputs through fetch(); TO_GLOBAL: outputs to dst (lane-contiguous: the last pass has Ns = M/16 >= T);
  // otherwise the LDS buffer, in place: every thread has read before any thread writes (the barrier in the middle).
  template <int R, int NS, bool FROM_GLOBAL, bool TO_GLOBAL>
  static MI_DEVICE void pass(const Rows &rows, const cf *MI_RESTRICT tw, cf *MI_RESTRICT lds, cf *MI_RESTRICT dst, int tid) {
    constexpr int PER = 16 / R;  // butterflies per thread
    constexpr int LOG2NSR = __builtin_ctz(NS * R);
    static_assert(FROM_GLOBAL || PER == 1, "LDS to LDS passes are radix 16");
    cf v[PER][R];
    MI_UNROLL
    for (int b = 0; b < PER; ++b) {
      const int j = tid + b * T;
      MI_UNROLL
      for (int t = 0; t < R; ++t) {
        const int i = j + t * (M / R);
        if constexpr (FROM_GLOBAL) {
          v[b][t] = fetch(rows, i);
        } else {
          v[b][t] = lds[tiled_swz(i)];
        }
      }
    }
    if constexpr (!FROM_GLOBAL && !TO_GLOBAL) {
      MI_SYNC();
    }
    MI_UNROLL
    for (int b = 0; b < PER; ++b) {
      const int j = tid + b * T;
      const int k = j & (NS - 1);
      if constexpr (NS > 1) {
        apply_twiddles<DIR, R>(v[b], tw[tw_offset(LOG2NSR) + k]);
      }
      dftR<DIR, R>(v[b]);
      const int base = (j - k) * R + k;
      MI_UNROLL
      for (int u = 0; u < R; ++u) {
        if constexpr (TO_GLOBAL) {
          dst[base + u * NS] = v[b][out_pos<R>(u)];
        } else {
          lds[tiled_swz(base + u * NS)] = v[b][out_pos<R>(u)];
        }
      }
    }
  }

  static MI_DEVICE void run(const Geometry &g, const TiledRowSrc &src, const cf *MI_RESTRICT tw, cf *MI_RESTRICT out_rows,
                            cf *lds) {
    const int tid = MI_TID_X;
    // workgroup = (item, [phase,] k1)
    const long long wg = MI_BID_X;
    const int k1 = static_cast<int>(wg % K1);
    const long long ip = wg / K1;
    const int p = SPECTRAL ? static_cast<int>(ip % g.P) : 0;
    const long long it = SPECTRAL ? ip / g.P : ip;
    cf *dst = out_rows + (ip * K1 + k1) * M;  // plain: X[item][k1]; spectral: B[item][p][k1]
    const Rows rows = make_rows(g, src, it, p, k1);
    if constexpr (R0 > 1) {
      pass<R0, 1, true, false>(rows, tw, lds, nullptr, tid);
      MI_SYNC();
      pass<16, R0, false, false>(rows, tw, lds, nullptr, tid);
      MI_SYNC();
      if constexpr (Cfg::NPASS == 4) {
        pass<16, R0 * 16, false, false>(rows, tw, lds, nullptr, tid);
        MI_SYNC();
      }
      pass<16, M / 16, false, true>(rows, tw, lds, dst, tid);
    } else {
      pass<16, 1, true, false>(rows, tw, lds, nullptr, tid);
      MI_SYNC();
      pass<16, M / 256, false, false>(rows, tw, lds, nullptr, tid);
      MI_SYNC();
      pass<16, M / 16, false, true>(rows, tw, lds, dst, tid);
    }
  }
};

// workgroup = (item, k1): X[item][k1][.] = FFT_M2(A[item][k1][.])
template <int LOG2M, int K1>
MI_GLOBAL MI_LAUNCH_BOUNDS((TiledRowCfg<LOG2M>::T), 1) void tiled_row_forward_kernel(Geometry g, TiledRowSrc src,
                                                                                     const cf *MI_RESTRICT tw,
                                                                                     cf *MI_RESTRICT X) {
  MI_DYN_SHARED(cf, lds);
  TiledRow<LOG2M, -1, false, K1>::run(g, src, tw, X, lds);
}
// workgroup = (item, phase, k1): B[item][p][k1][.] = IFFT_M2(spectral stage of phase p on X[item][k1][.])
template <int LOG2M, int K1>
MI_GLOBAL MI_LAUNCH_BOUNDS((TiledRowCfg<LOG2M>::T), 1) void tiled_row_inverse_kernel(Geometry g, TiledRowSrc src,
                                                                                     const cf *MI_RESTRICT tw,
                                                                                     cf *MI_RESTRICT B) {
  MI_DYN_SHARED(cf, lds);
  TiledRow<LOG2M, +1, true, K1>::run(g, src, tw, B, lds);
}

// ---- inverse column pass + overlap-discard into the staging planes ----------------------------------------------------
// thread = (item * P + p, n2); planes[item - item0][p][i] = y_p[Oc + i]  (the fused path's plane layout)
template <int K1>
MI_GLOBAL void tiled_store_kernel(Geometry g, const cf *MI_RESTRICT tw, const cf *MI_RESTRICT B, float *MI_RESTRICT planes,
                                  long long nrows, int newest_first) {
  const int M2 = g.K / K1;
  const long long gid = static_cast<long long>(MI_BID_X) * MI_BDIM_X + MI_TID_X;
  if (gid >= nrows * M2) {
    return;
  }
  // (item, phase) rows are walked NEWEST FIRST: B is larger than the 256 MB Infinity Cache and was written in row order just
  // before -- its tail may still be cached, its head is not (measured +1..2 % on the whole call: profiles/r03_q_two_level.txt)
  const long long slot = gid / M2;
  const int n2 = static_cast<int>(gid - slot * M2);
  const long long ip = newest_first ? nrows - 1 - slot : slot;
  const cf *src = B + ip * g.K + n2;
  cf v[K1];
  MI_UNROLL
  for (int k1 = 0; k1 < K1; ++k1) {
    v[k1] = src[static_cast<long long>(k1) * M2];
  }
  if (n2 > 0) {
    apply_twiddles<+1, K1>(v, tw[tw_offset(g.log2k) + n2]);  // conj(W_K^(n2 k1))
  }
  dftR<+1, K1>(v);
  float *plane = planes + ip * g.Bp;
  if ((g.Oc & 1) == 0) {
    // even history length (every shipped geometry): a complex word is kept or dropped whole -- one 8-byte store
    MI_UNROLL
    for (int n1 = 0; n1 < K1; ++n1) {
      const int n = 2 * (n1 * M2 + n2);  // compact samples n, n + 1
      if (n >= g.Oc) {
        *reinterpret_cast<cf *>(plane + (n - g.Oc)) = v[out_pos<K1>(n1)];
      }
    }
    return;
  }
  MI_UNROLL
  for (int n1 = 0; n1 < K1; ++n1) {
    const int n = 2 * (n1 * M2 + n2);  // compact samples n, n + 1
    const cf y = v[out_pos<K1>(n1)];
    if (n >= g.Oc) {
      plane[n - g.Oc] = y.x;
    }
    if (n + 1 >= g.Oc) {
      plane[n + 1 - g.Oc] = y.y;
    }
  }
}

}  // namespace miups
