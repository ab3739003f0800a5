// Fused overlap-save upsampler kernel for gfx950: ONE workgroup = one
// channel-block, everything between the PCM load and the PCM store stays in
// registers and LDS.
//
//   load   interleaved PCM frames (history + new) -> z[n] = x[2n] + j x[2n+1]
//   FFT_K  Stockham radix-(R0,16,..,16), passes staged through LDS (in place,
//          XOR-swizzled), first pass fed straight from HBM, last pass left in
//          registers as the butterfly sets {j + t*K/16}
//   split  real-FFT untangle of mirror pairs (k, K-k), kept in registers
//   for each output phase p (P = upsample factor):
//     multiply by the phase spectrum G_p (L2-resident), re-tangle, inverse
//     FFT_K radix-(16,..,16,R0) through the same LDS buffer, last pass stores
//     y_p[n], n >= Oc, to out frame (n-Oc)*P + p with the PCM conversion fused.
//
// Replaces, per channel-block, the reference's ProcessBlock body
// (src/vulkan/vulkan_streaming_upsampler.cpp:528-572): zero-stuff/overlap
// assembly, pack, forward C2C FFT_N, CPU spectral multiply, inverse C2C FFT_N,
// gather and overlap update -- with N-point transforms replaced by the exact
// polyphase identity: one K-point forward and P K-point
// inverse transforms, K = N / (2P).
//
// Thread layout: T = K/32 threads, each owns TWO radix-16 butterflies per
// radix-16 pass. In the two passes adjacent to the spectral stage thread tau
// owns butterfly sets S_tau and S_{J-tau} (J = K/16), which are mirror images
// under k -> K-k, so the untangle needs no data from another thread. Thread 0
// owns the two self-mirrored sets S_0 and S_{J/2}.
#pragma once

#include "common.h"
#include "fft_radix.h"
#include "kernels_generic.h"
#include "pcm.h"

namespace miups {

MI_DEVICE int lds_swz(int i) { return i ^ ((i >> 4) & 15); }

// exp(-2*pi*i*t/32), t = 0..16
MI_DEVICE cf w32(int t) {
  constexpr float c[17] = {1.0f,
                           0.98078528040323044913f,
                           0.92387953251128675613f,
                           0.83146961230254523708f,
                           0.70710678118654752440f,
                           0.55557023301960222474f,
                           0.38268343236508977173f,
                           0.19509032201612826785f,
                           0.0f,
                           -0.19509032201612826785f,
                           -0.38268343236508977173f,
                           -0.55557023301960222474f,
                           -0.70710678118654752440f,
                           -0.83146961230254523708f,
                           -0.92387953251128675613f,
                           -0.98078528040323044913f,
                           -1.0f};
  constexpr float s[17] = {0.0f,
                           0.19509032201612826785f,
                           0.38268343236508977173f,
                           0.55557023301960222474f,
                           0.70710678118654752440f,
                           0.83146961230254523708f,
                           0.92387953251128675613f,
                           0.98078528040323044913f,
                           1.0f,
                           0.98078528040323044913f,
                           0.92387953251128675613f,
                           0.83146961230254523708f,
                           0.70710678118654752440f,
                           0.55557023301960222474f,
                           0.38268343236508977173f,
                           0.19509032201612826785f,
                           0.0f};
  return mk(c[t], -s[t]);
}

// ---- mirror-pair algebra (see gen_multiply_kernel for the per-bin form) ---
//   xa = (u+v) - jW(u-v), xb = (u+v) + jW(u-v),  v = conj(zm)
MI_DEVICE void pair_split(cf u, cf zm, cf W, cf &xa, cf &xb) {
  const cf v = cconj(zm);
  const cf s = cadd(u, v);
  const cf d = cmulj(cmul(W, csub(u, v)));
  xa = csub(s, d);
  xb = cadd(s, d);
}
//   P = xa*gs, Q = xb*gc ; zk = (P+Q) + j conj(W)(P-Q) ; zkm = conj((P+Q) - j conj(W)(P-Q))
MI_DEVICE void pair_phase(cf xa, cf xb, cf W, cf gs, cf gc, cf &zk, cf &zkm) {
  const cf P = cmul(xa, gs);
  const cf Q = cmul(xb, gc);
  const cf S = cadd(P, Q);
  const cf D = cmulj(cmulc(csub(P, Q), W));
  zk = cadd(S, D);
  zkm = cconj(csub(S, D));
}

template <int LOG2K>
struct FusedCfg {
  static constexpr int K = 1 << LOG2K;
  static constexpr int J = K / 16;       // radix-16 butterflies per pass
  static constexpr int T = K / 32;       // threads per workgroup
  static constexpr int R0 = 1 << (LOG2K % 4);
  static constexpr int LOG2R0 = LOG2K % 4;
  static constexpr int N16 = LOG2K / 4;  // radix-16 passes
  static constexpr int LDS_BYTES = K * 8;
  static_assert(LOG2K >= 5 && LOG2K <= 14, "fused kernel covers K = 32 .. 16384");
};

template <int LOG2K>
struct FusedKernel {
  using Cfg = FusedCfg<LOG2K>;
  static constexpr int K = Cfg::K, J = Cfg::J, T = Cfg::T, R0 = Cfg::R0, N16 = Cfg::N16;
  static constexpr int LOG2R0 = Cfg::LOG2R0;

  // ---- LDS access for one radix-R butterfly -------------------------------
  template <int R>
  static MI_DEVICE void lds_read(const cf *lds, int j, cf *v) {
    MI_UNROLL
    for (int t = 0; t < R; ++t) {
      v[t] = lds[lds_swz(j + t * (K / R))];
    }
  }
  template <int R, int NS>
  static MI_DEVICE void lds_write(cf *lds, int j, const cf *v) {
    const int k = j & (NS - 1);
    const int base = (j - k) * R + k;
    MI_UNROLL
    for (int u = 0; u < R; ++u) {
      lds[lds_swz(base + u * NS)] = v[out_pos<R>(u)];
    }
  }
  template <int DIR, int R, int NS, int LOG2NSR>
  static MI_DEVICE void butterfly(cf *v, int j, const cf *tw) {
    if constexpr (NS > 1) {
      apply_twiddles<DIR, R>(v, tw[tw_offset(LOG2NSR) + (j & (NS - 1))]);
    }
    dftR<DIR, R>(v);
  }

  // ---- global load of one radix-R butterfly's inputs (forward pass 0) -----
  template <int R>
  static MI_DEVICE void global_read(const Geometry &g, const IoDesc &io, int s, int c, int blk, int j, cf *v) {
    MI_UNROLL
    for (int t = 0; t < R; ++t) {
      const int n = j + t * (K / R);
      v[t] = mk(compact_sample(g, io, s, c, blk, 2 * n), compact_sample(g, io, s, c, blk, 2 * n + 1));
    }
  }
  // ---- global store of one radix-R butterfly's outputs (inverse last pass) -
  template <int R>
  static MI_DEVICE void global_write(const Geometry &g, const IoDesc &io, int s, int c, int blk, int p, int j,
                                     const cf *v) {
    MI_UNROLL
    for (int u = 0; u < R; ++u) {
      const int n = j + u * (K / R);
      const cf y = v[out_pos<R>(u)];
      store_output(g, io, s, c, blk, p, 2 * n, y.x);
      store_output(g, io, s, c, blk, p, 2 * n + 1, y.y);
    }
  }

  // forward radix-16 pass number P16 (0-based among the radix-16 passes)
  template <int P16>
  static MI_DEVICE void fwd16(const Geometry &g, const IoDesc &io, int s, int c, int blk, cf *lds, const cf *tw,
                              int tid, cf *A, cf *B) {
    constexpr int NS = R0 * (1 << (4 * P16));
    constexpr int LOG2NSR = LOG2R0 + 4 * P16 + 4;
    constexpr bool kFromGlobal = (R0 == 1 && P16 == 0);
    constexpr bool kLast = (P16 == N16 - 1);
    const int jA = tid;
    const int jB = kLast ? (tid == 0 ? T : J - tid) : tid + T;
    if constexpr (kFromGlobal) {
      global_read<16>(g, io, s, c, blk, jA, A);
      global_read<16>(g, io, s, c, blk, jB, B);
    } else {
      lds_read<16>(lds, jA, A);
      lds_read<16>(lds, jB, B);
      MI_SYNC();  // every read of this pass done before anyone overwrites
    }
    butterfly<-1, 16, NS, LOG2NSR>(A, jA, tw);
    butterfly<-1, 16, NS, LOG2NSR>(B, jB, tw);
    if constexpr (!kLast) {
      lds_write<16, NS>(lds, jA, A);
      lds_write<16, NS>(lds, jB, B);
      MI_SYNC();
    }
  }

  // inverse radix-16 pass number P16; pass 0 takes its inputs from A/B
  template <int P16>
  static MI_DEVICE void inv16(const Geometry &g, const IoDesc &io, int s, int c, int blk, int p, cf *lds,
                              const cf *tw, int tid, cf *A, cf *B) {
    constexpr int NS = 1 << (4 * P16);
    constexpr int LOG2NSR = 4 * P16 + 4;
    constexpr bool kFirst = (P16 == 0);
    constexpr bool kToGlobal = (R0 == 1 && P16 == N16 - 1);
    const int jA = tid;
    const int jB = kFirst ? (tid == 0 ? T : J - tid) : tid + T;
    if constexpr (!kFirst) {
      lds_read<16>(lds, jA, A);
      lds_read<16>(lds, jB, B);
      if constexpr (!kToGlobal) {
        MI_SYNC();
      }
    }
    butterfly<+1, 16, NS, LOG2NSR>(A, jA, tw);
    butterfly<+1, 16, NS, LOG2NSR>(B, jB, tw);
    if constexpr (kToGlobal) {
      global_write<16>(g, io, s, c, blk, p, jA, A);
      global_write<16>(g, io, s, c, blk, p, jB, B);
      MI_SYNC();  // LDS free for the next phase
    } else {
      lds_write<16, NS>(lds, jA, A);
      lds_write<16, NS>(lds, jB, B);
      MI_SYNC();
    }
  }

  // spectral stage for one phase. kSelf = thread 0 (self-mirrored sets).
  // On entry Xa/Xb hold the split spectrum, on exit A/B hold the inputs of
  // inverse pass 0 in natural order (element t of butterflies jA / jB).
  template <bool kSelf>
  static MI_DEVICE void phase_inputs(int tid, const cf *Xa, const cf *Xb, cf Wa, cf Wb, const cf *MI_RESTRICT gs,
                                     const cf *MI_RESTRICT gc, cf *A, cf *B) {
    if constexpr (!kSelf) {
      // pair t: k = tid + t*J  <->  K-k = (J-tid) + (15-t)*J
      MI_UNROLL
      for (int t = 0; t < 16; ++t) {
        const int k = tid + t * J;
        pair_phase(Xa[t], Xb[t], cmul(Wa, w32(t)), gs[k], gc[k], A[t], B[15 - t]);
      }
    } else {
      // set S_0: k = t*J <-> (16-t)*J, t = 0..8 (t = 0 pairs DC with Nyquist
      // through gc[0] = conj Gs[K]; t = 8 is its own mirror)
      MI_UNROLL
      for (int t = 0; t <= 8; ++t) {
        cf zk, zkm;
        pair_phase(Xa[t], Xb[t], w32(t), gs[t * J], gc[t * J], zk, zkm);
        A[t] = zk;
        if (t >= 1 && t <= 7) {
          A[16 - t] = zkm;
        }
      }
      // set S_T: k = T + t*J <-> T + (15-t)*J, t = 0..7
      MI_UNROLL
      for (int t = 0; t < 8; ++t) {
        const int k = T + t * J;
        pair_phase(Xa[9 + t], Xb[9 + t], cmul(Wb, w32(t)), gs[k], gc[k], B[t], B[15 - t]);
      }
    }
  }

  template <bool kSelf>
  static MI_DEVICE void split_spectrum(int tid, const cf *A, const cf *B, cf Wa, cf Wb, cf *Xa, cf *Xb) {
    // A[out_pos<16>(t)] = Z[jA + t*J], B likewise for jB
    if constexpr (!kSelf) {
      MI_UNROLL
      for (int t = 0; t < 16; ++t) {
        pair_split(A[out_pos<16>(t)], B[out_pos<16>(15 - t)], cmul(Wa, w32(t)), Xa[t], Xb[t]);
      }
    } else {
      MI_UNROLL
      for (int t = 0; t <= 8; ++t) {
        pair_split(A[out_pos<16>(t)], A[out_pos<16>((16 - t) & 15)], w32(t), Xa[t], Xb[t]);
      }
      MI_UNROLL
      for (int t = 0; t < 8; ++t) {
        pair_split(B[out_pos<16>(t)], B[out_pos<16>(15 - t)], cmul(Wb, w32(t)), Xa[9 + t], Xb[9 + t]);
      }
    }
  }

  static MI_DEVICE void run(const Geometry &g, const IoDesc &io, const cf *MI_RESTRICT tw, const cf *MI_RESTRICT Wm,
                            const cf *MI_RESTRICT Gs, const cf *MI_RESTRICT Gc, cf *lds) {
    const int tid = MI_TID_X;
    const int item = MI_BID_X;
    const int sc_count = io.streams * io.channels;
    const int blk = item / sc_count;
    const int sc = item % sc_count;
    const int s = sc / io.channels, c = sc % io.channels;

    cf A[16], B[16];

    // ------------------------------ forward ------------------------------
    if constexpr (R0 > 1) {
      MI_UNROLL
      for (int i = 0; i < 32 / R0; ++i) {
        const int j = tid + i * T;
        cf v[R0];
        global_read<R0>(g, io, s, c, blk, j, v);
        dftR<-1, R0>(v);
        lds_write<R0, 1>(lds, j, v);
      }
      MI_SYNC();
    }
    fwd16<0>(g, io, s, c, blk, lds, tw, tid, A, B);
    if constexpr (N16 >= 2) {
      fwd16<1>(g, io, s, c, blk, lds, tw, tid, A, B);
    }
    if constexpr (N16 >= 3) {
      fwd16<2>(g, io, s, c, blk, lds, tw, tid, A, B);
    }

    // ------------------------- split (once per block) --------------------
    cf Xa[17], Xb[17];
    const cf Wa = Wm[tid];  // W_M^tid ; thread 0: W_M^0 = 1 (unused)
    const cf Wb = Wm[T];    // W_M^T   ; used by thread 0 only
    if (tid == 0) {
      split_spectrum<true>(tid, A, B, Wa, Wb, Xa, Xb);
    } else {
      split_spectrum<false>(tid, A, B, Wa, Wb, Xa, Xb);
    }

    // --------------------------- per output phase ------------------------
    for (int p = 0; p < g.P; ++p) {
      const cf *gs = Gs + static_cast<long long>(p) * K;
      const cf *gc = Gc + static_cast<long long>(p) * K;
      if (tid == 0) {
        phase_inputs<true>(tid, Xa, Xb, Wa, Wb, gs, gc, A, B);
      } else {
        phase_inputs<false>(tid, Xa, Xb, Wa, Wb, gs, gc, A, B);
      }
      inv16<0>(g, io, s, c, blk, p, lds, tw, tid, A, B);
      if constexpr (N16 >= 2) {
        inv16<1>(g, io, s, c, blk, p, lds, tw, tid, A, B);
      }
      if constexpr (N16 >= 3) {
        inv16<2>(g, io, s, c, blk, p, lds, tw, tid, A, B);
      }
      if constexpr (R0 > 1) {
        MI_UNROLL
        for (int i = 0; i < 32 / R0; ++i) {
          const int j = tid + i * T;
          cf v[R0];
          lds_read<R0>(lds, j, v);
          butterfly<+1, R0, K / R0, LOG2K>(v, j, tw);
          global_write<R0>(g, io, s, c, blk, p, j, v);
        }
        MI_SYNC();  // LDS free for the next phase
      }
    }
  }
};

template <int LOG2K>
MI_GLOBAL MI_LAUNCH_BOUNDS((FusedCfg<LOG2K>::T < 64 ? 64 : FusedCfg<LOG2K>::T), 1) void fused_kernel(
    Geometry g, IoDesc io, const cf *MI_RESTRICT tw, const cf *MI_RESTRICT Wm, const cf *MI_RESTRICT Gs,
    const cf *MI_RESTRICT Gc) {
  MI_DYN_SHARED(cf, lds);
  FusedKernel<LOG2K>::run(g, io, tw, Wm, Gs, Gc, lds);
}

}  // namespace miups
