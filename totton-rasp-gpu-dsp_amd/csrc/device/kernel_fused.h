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
//
// PCM formats are a run-time property of the engine; the format switch is
// hoisted around the first forward pass and the last inverse pass (the only
// code that touches PCM), so no per-sample branch is executed.
#pragma once

#include "common.h"
#include "fft_radix.h"
#include "kernels_generic.h"
#include "pcm.h"

#if defined(MIUPS_HOST_EMU)
#define MI_SCHED_FENCE()
#define MI_OPAQUE_VGPR(x)
#else
#define MI_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// Makes `x` look freshly defined: address arithmetic derived from it cannot be
// hoisted out of the phase loop (hipcc otherwise precomputes every LDS / output
// offset and store predicate of all passes and keeps >100 registers live).
#define MI_OPAQUE_VGPR(x) asm volatile("" : "+v"(x))
#endif

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

// ---- format-typed sample access (p points AT the sample) -------------------
template <int FMT>
MI_DEVICE float sample_load(const char *p) {
  if constexpr (FMT == kF32) {
    return *reinterpret_cast<const float *>(p);
  } else if constexpr (FMT == kS32) {
    return static_cast<float>(*reinterpret_cast<const int32_t *>(p)) * (1.0f / 2147483648.0f);
  } else if constexpr (FMT == kS16) {
    return static_cast<float>(*reinterpret_cast<const int16_t *>(p)) * (1.0f / 32768.0f);
  } else {
    return pcm_load(p, kS24_3LE, 0);
  }
}
template <int FMT>
MI_DEVICE void sample_store(char *p, float v) {
  pcm_store(p, FMT, 0, v);  // FMT is a constant here: the switch folds away
}

// Per-workgroup addressing (fused path requires S == 1, so compact sample n of
// this block is input frame f0 + n, f0 = blk*Bc - Oc; frames < 0 are history).
struct BlockIo {
  const char *pin;    // where compact sample 0 would be in `in`   (valid for n >= n_hist)
  const char *phist;  // where compact sample 0 is in the history  (valid for n <  n_hist)
  // byte step between consecutive frames of one channel; the host only selects
  // the fused path when M * step < 2^31
  int in_step;
  int n_hist;
  int Oc;
};

MI_DEVICE BlockIo make_block_io(const Geometry &g, const IoDesc &io, int s, int c, int blk) {
  BlockIo b;
  const long long ib = pcm_bytes(io.in_fmt);
  const long long f0 = static_cast<long long>(blk) * g.Bc - g.Oc;
  b.in_step = static_cast<int>(ib * io.channels);
  b.pin = static_cast<const char *>(io.in) + s * io.in_stream_stride + (f0 * io.channels + c) * ib;
  b.phist = static_cast<const char *>(io.hist) + s * io.hist_stream_stride +
            ((g.hist_frames + f0) * io.channels + c) * ib;
  b.n_hist = f0 >= 0 ? 0 : (-f0 > g.M ? g.M : static_cast<int>(-f0));
  b.Oc = g.Oc;
  return b;
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
  // The swizzle only permutes within aligned groups of 16, so when the element
  // stride is a multiple of 256 it is the same for every element.
  template <int R>
  static MI_DEVICE void lds_read(const cf *lds, int j, cf *v) {
    constexpr int stride = K / R;
    if constexpr (stride % 256 == 0) {
      const cf *p = lds + lds_swz(j);
      MI_UNROLL
      for (int t = 0; t < R; ++t) {
        v[t] = p[t * stride];
      }
    } else {
      MI_UNROLL
      for (int t = 0; t < R; ++t) {
        v[t] = lds[lds_swz(j + t * stride)];
      }
    }
  }
  template <int R, int NS>
  static MI_DEVICE void lds_write(cf *lds, int j, const cf *v) {
    const int k = j & (NS - 1);
    const int base = (j - k) * R + k;
    if constexpr (NS % 256 == 0) {
      cf *p = lds + lds_swz(base);
      MI_UNROLL
      for (int u = 0; u < R; ++u) {
        p[u * NS] = v[out_pos<R>(u)];
      }
    } else {
      MI_UNROLL
      for (int u = 0; u < R; ++u) {
        lds[lds_swz(base + u * NS)] = v[out_pos<R>(u)];
      }
    }
  }
  // Twiddle loads are issued at the top of a pass, ahead of the LDS reads and the
  // barrier, so their L2 round trip overlaps those instead of following them.
  template <int NS, int LOG2NSR>
  static MI_DEVICE cf load_tw(const cf *tw, int j) {
    if constexpr (NS > 1) {
      return tw[tw_offset(LOG2NSR) + (j & (NS - 1))];
    } else {
      return mk(1.0f, 0.0f);
    }
  }
  template <int DIR, int R, int NS>
  static MI_DEVICE void butterfly(cf *v, cf w) {
    if constexpr (NS > 1) {
      apply_twiddles<DIR, R>(v, w);
    }
    dftR<DIR, R>(v);
  }

  // ---- global load of one radix-R butterfly's inputs (forward pass 0) -----
  // kHist = false: the whole block lies in `in` (true for all but the first
  // blocks of a call), one uniform base + 32-bit offsets.
  template <int FMT, int R, bool kHist>
  static MI_DEVICE void global_read(const BlockIo &b, int j, cf *v) {
    MI_UNROLL
    for (int t = 0; t < R; ++t) {
      const int n = 2 * (j + t * (K / R));
      const unsigned o0 = static_cast<unsigned>(n) * static_cast<unsigned>(b.in_step);
      const unsigned o1 = o0 + static_cast<unsigned>(b.in_step);
      if constexpr (kHist) {
        const char *p0 = (n < b.n_hist ? b.phist : b.pin) + o0;
        const char *p1 = (n + 1 < b.n_hist ? b.phist : b.pin) + o1;
        v[t] = mk(sample_load<FMT>(p0), sample_load<FMT>(p1));
      } else {
        v[t] = mk(sample_load<FMT>(b.pin + o0), sample_load<FMT>(b.pin + o1));
      }
    }
  }
  // ---- store of one radix-R butterfly's outputs (inverse last pass) ---------
  // overlap-discard (:566-569): compact samples n < Oc are dropped; the kept
  // ones go, still fp32 and phase-planar, to this workgroup's staging plane
  // (plane[i] = y_p[Oc + i]) with lane-contiguous 8-byte stores. The epilogue
  // turns the planes into interleaved PCM frames.
  template <int R, bool kEvenOc>
  static MI_DEVICE void plane_write(float *plane, int Oc, int j, const cf *v) {
    MI_UNROLL
    for (int u = 0; u < R; ++u) {
      const int n = 2 * (j + u * (K / R));
      const cf y = v[out_pos<R>(u)];
      if constexpr (kEvenOc) {
        if (n >= Oc) {
          *reinterpret_cast<cf *>(plane + (n - Oc)) = y;
        }
      } else {
        if (n >= Oc) {
          plane[n - Oc] = y.x;
        }
        if (n + 1 >= Oc) {
          plane[n + 1 - Oc] = y.y;
        }
      }
    }
  }

  // ---- forward pass 0 when it is a radix-R0 pass (R0 > 1) -------------------
  template <int FMT, bool kHist>
  static MI_DEVICE void fwd_r0_impl(const BlockIo &b, cf *lds, int tid) {
    MI_UNROLL
    for (int i = 0; i < 32 / R0; ++i) {
      const int j = tid + i * T;
      cf v[R0];
      global_read<FMT, R0, kHist>(b, j, v);
      dftR<-1, R0>(v);
      lds_write<R0, 1>(lds, j, v);
    }
  }
  template <int FMT>
  static MI_DEVICE void fwd_r0(const BlockIo &b, cf *lds, int tid) {
    if (b.n_hist == 0) {
      fwd_r0_impl<FMT, false>(b, lds, tid);
    } else {
      fwd_r0_impl<FMT, true>(b, lds, tid);
    }
  }
  // ---- forward pass 0 when it is a radix-16 pass (R0 == 1, N16 >= 2) --------
  template <int FMT>
  static MI_DEVICE void fwd16_from_global(const BlockIo &b, cf *lds, int tid, cf *A, cf *B) {
    if (b.n_hist == 0) {
      global_read<FMT, 16, false>(b, tid, A);
      global_read<FMT, 16, false>(b, tid + T, B);
    } else {
      global_read<FMT, 16, true>(b, tid, A);
      global_read<FMT, 16, true>(b, tid + T, B);
    }
    dft16<-1>(A);
    lds_write<16, 1>(lds, tid, A);
    dft16<-1>(B);
    lds_write<16, 1>(lds, tid + T, B);
  }

  // forward radix-16 pass P16 reading LDS (every pass except a global pass 0)
  template <int P16>
  static MI_DEVICE void fwd16(cf *lds, const cf *tw, int tid, cf *A, cf *B) {
    constexpr int NS = R0 * (1 << (4 * P16));
    constexpr int LOG2NSR = LOG2R0 + 4 * P16 + 4;
    constexpr bool kLast = (P16 == N16 - 1);
    const int jA = tid;
    const int jB = kLast ? (tid == 0 ? T : J - tid) : tid + T;
    const cf wA = load_tw<NS, LOG2NSR>(tw, jA);
    // jB = jA + T indexes the same entry whenever the table period NS divides T
    const cf wB = (!kLast && T % NS == 0) ? wA : load_tw<NS, LOG2NSR>(tw, jB);
    lds_read<16>(lds, jA, A);
    lds_read<16>(lds, jB, B);
    MI_SYNC();  // every read of this pass done before anyone overwrites
    butterfly<-1, 16, NS>(A, wA);
    if constexpr (!kLast) {
      lds_write<16, NS>(lds, jA, A);
    }
    MI_SCHED_FENCE();
    butterfly<-1, 16, NS>(B, wB);
    if constexpr (!kLast) {
      lds_write<16, NS>(lds, jB, B);
      MI_SYNC();
    }
  }

  // inverse radix-16 pass P16 that ends in LDS; pass 0 takes its inputs from A/B
  template <int P16>
  static MI_DEVICE void inv16(cf *lds, const cf *tw, int tid, cf *A, cf *B) {
    constexpr int NS = 1 << (4 * P16);
    constexpr int LOG2NSR = 4 * P16 + 4;
    constexpr bool kFirst = (P16 == 0);
    const int jA = tid;
    const int jB = kFirst ? (tid == 0 ? T : J - tid) : tid + T;
    const cf wA = load_tw<NS, LOG2NSR>(tw, jA);
    const cf wB = (T % NS == 0) ? wA : load_tw<NS, LOG2NSR>(tw, jB);
    if constexpr (!kFirst) {
      lds_read<16>(lds, jA, A);
      lds_read<16>(lds, jB, B);
      MI_SYNC();
    }
    butterfly<+1, 16, NS>(A, wA);
    lds_write<16, NS>(lds, jA, A);
    MI_SCHED_FENCE();
    butterfly<+1, 16, NS>(B, wB);
    lds_write<16, NS>(lds, jB, B);
    MI_SYNC();
  }
  // inverse last pass when it is a radix-16 pass (R0 == 1): LDS -> staging plane
  template <bool kEvenOc>
  static MI_DEVICE void inv16_to_plane(float *plane, int Oc, const cf *lds, const cf *tw, int tid, cf *A, cf *B) {
    constexpr int P16 = N16 - 1;
    constexpr int NS = 1 << (4 * P16);
    const cf wA = load_tw<NS, 4 * P16 + 4>(tw, tid);
    const cf wB = (T % NS == 0) ? wA : load_tw<NS, 4 * P16 + 4>(tw, tid + T);
    lds_read<16>(lds, tid, A);
    lds_read<16>(lds, tid + T, B);
    butterfly<+1, 16, NS>(A, wA);
    plane_write<16, kEvenOc>(plane, Oc, tid, A);
    MI_SCHED_FENCE();
    butterfly<+1, 16, NS>(B, wB);
    plane_write<16, kEvenOc>(plane, Oc, tid + T, B);
  }
  // inverse last pass when it is the radix-R0 pass: LDS -> staging plane
  template <bool kEvenOc>
  static MI_DEVICE void inv_r0_to_plane(float *plane, int Oc, const cf *lds, const cf *tw, int tid) {
    // W_K^(tid + i*T) = W_K^tid * W_32^i : one table load for all 32/R0 butterflies
    const cf w0 = load_tw<K / R0, LOG2K>(tw, tid);
    MI_UNROLL
    for (int i = 0; i < 32 / R0; ++i) {
      const int j = tid + i * T;
      cf v[R0];
      lds_read<R0>(lds, j, v);
      butterfly<+1, R0, K / R0>(v, i == 0 ? w0 : cmul(w0, w32(i)));
      plane_write<R0, kEvenOc>(plane, Oc, j, v);
      if ((i & 3) == 3) {
        MI_SCHED_FENCE();  // keep at most 4 butterflies' registers in flight
      }
    }
  }

  // ---- epilogue: staging planes -> interleaved PCM frames ---------------------
  // scr[(cc*P + p)*Bc + i] = y_p[Oc + i] of channel c0+cc; output frame
  // blk*B + i*P + p, channel c0+cc (reference: interleave + ConvertFloatToPcm,
  // alsa_streamer_main.cpp:327-329,550-552; alsa_common.cpp:87-127).
  //
  // Fast form (this group is the whole frame, 4-byte samples, 16-byte aligned):
  // a thread gathers VPT = cg*pg values (pg consecutive phases of one i, every
  // channel) = VPT consecutive output samples and writes them as 16-byte vectors;
  // consecutive lanes write consecutive memory.
  template <int FMT, int VPT>
  static MI_DEVICE void epilogue_vec(const Geometry &g, const IoDesc &io, char *out_blk, const float *scr, int tid) {
    const int cg = io.cg, pg = VPT / cg, qn = g.P / pg;
    const int units = g.Bc * qn;
    // kDepth units per step so that kDepth*VPT plane loads are in flight per lane
    // (the planes come back from L2 / Infinity Cache, ~1-2 us round trip)
    constexpr int kDepth = VPT >= 16 ? 2 : 4;
    for (int base = tid; base < units; base += T * kDepth) {
      float v[kDepth][VPT];
      MI_UNROLL
      for (int d = 0; d < kDepth; ++d) {
        const int unit = base + d * T;
        if (unit < units) {
          const int i = unit / qn, q = unit - i * qn;
          MI_UNROLL
          for (int e = 0; e < VPT; ++e) {
            const int pp = e / cg, cc = e - pp * cg;
            v[d][e] = scr[(cc * g.P + q * pg + pp) * g.Bc + i];
          }
        }
      }
      MI_UNROLL
      for (int d = 0; d < kDepth; ++d) {
        const int unit = base + d * T;
        if (unit < units) {
          char *dst = out_blk + static_cast<long long>(unit) * (VPT * 4);
          MI_UNROLL
          for (int e = 0; e < VPT; e += 4) {
            if constexpr (FMT == kF32) {
              struct alignas(16) F4 { float a, b, c, d; };
              *reinterpret_cast<F4 *>(dst + 4 * e) = F4{v[d][e], v[d][e + 1], v[d][e + 2], v[d][e + 3]};
            } else {
              struct alignas(16) I4 { int32_t a, b, c, d; };
              I4 w;
              w.a = static_cast<int32_t>(pcm_clamp(v[d][e], 0.9999999f) * 2147483648.0f);
              w.b = static_cast<int32_t>(pcm_clamp(v[d][e + 1], 0.9999999f) * 2147483648.0f);
              w.c = static_cast<int32_t>(pcm_clamp(v[d][e + 2], 0.9999999f) * 2147483648.0f);
              w.d = static_cast<int32_t>(pcm_clamp(v[d][e + 3], 0.9999999f) * 2147483648.0f);
              *reinterpret_cast<I4 *>(dst + 4 * e) = w;
            }
          }
        }
      }
    }
  }
  // General form: one output sample per thread and step, lanes in output order.
  template <int FMT>
  static MI_DEVICE void epilogue_scalar(const Geometry &g, const IoDesc &io, char *out_blk, const float *scr, int tid) {
    const int cg = io.cg;
    const long long total = static_cast<long long>(g.B) * cg;
    for (long long e = tid; e < total; e += T) {
      const int m = static_cast<int>(e / cg), cc = static_cast<int>(e - static_cast<long long>(m) * cg);
      const int i = m / g.P, p = m - i * g.P;
      pcm_store(out_blk, FMT, static_cast<long long>(m) * io.channels + cc, scr[(cc * g.P + p) * g.Bc + i]);
    }
  }
  static MI_DEVICE void epilogue(const Geometry &g, const IoDesc &io, int s, int c0, int blk, const float *scr, int tid) {
    const int ob = pcm_bytes(io.out_fmt);
    char *out_blk = static_cast<char *>(io.out) + s * io.out_stream_stride +
                    (static_cast<long long>(blk) * g.B * io.channels + c0) * ob;
    const int cg = io.cg;
    const bool pow2 = (cg & (cg - 1)) == 0 && (g.P & (g.P - 1)) == 0;
    const int pg = pow2 ? (cg * g.P <= 16 ? g.P : (16 / cg > 0 ? 16 / cg : 1)) : 1;
    const int vpt = cg * pg;
    const bool vec = pow2 && io.out_vec_ok && cg == io.channels && cg <= 16 && vpt >= 4 &&
                     (io.out_fmt == kF32 || io.out_fmt == kS32);
    if (vec) {
      if (io.out_fmt == kF32) {
        if (vpt == 4) epilogue_vec<kF32, 4>(g, io, out_blk, scr, tid);
        else if (vpt == 8) epilogue_vec<kF32, 8>(g, io, out_blk, scr, tid);
        else epilogue_vec<kF32, 16>(g, io, out_blk, scr, tid);
      } else {
        if (vpt == 4) epilogue_vec<kS32, 4>(g, io, out_blk, scr, tid);
        else if (vpt == 8) epilogue_vec<kS32, 8>(g, io, out_blk, scr, tid);
        else epilogue_vec<kS32, 16>(g, io, out_blk, scr, tid);
      }
      return;
    }
    switch (io.out_fmt) {
      case kS32: epilogue_scalar<kS32>(g, io, out_blk, scr, tid); break;
      case kF32: epilogue_scalar<kF32>(g, io, out_blk, scr, tid); break;
      case kS16: epilogue_scalar<kS16>(g, io, out_blk, scr, tid); break;
      default: epilogue_scalar<kS24_3LE>(g, io, out_blk, scr, tid); break;
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
      const cf *ps = gs + tid, *pc = gc + tid;
      MI_UNROLL
      for (int t = 0; t < 16; ++t) {
        pair_phase(Xa[t], Xb[t], cmul(Wa, w32(t)), ps[t * J], pc[t * J], A[t], B[15 - t]);
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

  // One channel-block: forward FFT, split, then per phase multiply + inverse FFT
  // into this channel's staging planes (scr_c = [P][Bc] floats).
  static MI_DEVICE void channel_block(const Geometry &g, const IoDesc &io, const BlockIo &b, float *scr_c,
                                      const cf *MI_RESTRICT tw, const cf *MI_RESTRICT Wm, const cf *MI_RESTRICT Gs,
                                      const cf *MI_RESTRICT Gc, cf *lds, int tid) {
    cf A[16], B[16];

    // ------------------------------ forward ------------------------------
    if constexpr (R0 > 1) {
      switch (io.in_fmt) {
        case kS32: fwd_r0<kS32>(b, lds, tid); break;
        case kF32: fwd_r0<kF32>(b, lds, tid); break;
        case kS16: fwd_r0<kS16>(b, lds, tid); break;
        default: fwd_r0<kS24_3LE>(b, lds, tid); break;
      }
      MI_SYNC();
      fwd16<0>(lds, tw, tid, A, B);
    } else {
      switch (io.in_fmt) {
        case kS32: fwd16_from_global<kS32>(b, lds, tid, A, B); break;
        case kF32: fwd16_from_global<kF32>(b, lds, tid, A, B); break;
        case kS16: fwd16_from_global<kS16>(b, lds, tid, A, B); break;
        default: fwd16_from_global<kS24_3LE>(b, lds, tid, A, B); break;
      }
      MI_SYNC();
    }
    if constexpr (N16 >= 2) {
      fwd16<1>(lds, tw, tid, A, B);
    }
    if constexpr (N16 >= 3) {
      fwd16<2>(lds, tw, tid, A, B);
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
    const bool evenOc = (b.Oc & 1) == 0;
    for (int p = 0; p < g.P; ++p) {
      const cf *gs = Gs + static_cast<long long>(p) * K;
      const cf *gc = Gc + static_cast<long long>(p) * K;
      float *plane = scr_c + static_cast<long long>(p) * g.Bc;
      int tl = tid;  // per-phase copy of the thread index (see MI_OPAQUE_VGPR)
      MI_OPAQUE_VGPR(tl);
      if (tid == 0) {
        phase_inputs<true>(tl, Xa, Xb, Wa, Wb, gs, gc, A, B);
      } else {
        phase_inputs<false>(tl, Xa, Xb, Wa, Wb, gs, gc, A, B);
      }
      constexpr int kLdsInv = (R0 > 1) ? N16 : N16 - 1;  // inverse radix-16 passes that end in LDS
      if constexpr (kLdsInv >= 1) {
        inv16<0>(lds, tw, tl, A, B);
      }
      if constexpr (kLdsInv >= 2) {
        MI_OPAQUE_VGPR(tl);
        inv16<1>(lds, tw, tl, A, B);
      }
      if constexpr (kLdsInv >= 3) {
        MI_OPAQUE_VGPR(tl);
        inv16<2>(lds, tw, tl, A, B);
      }
      MI_OPAQUE_VGPR(tl);
      if constexpr (R0 > 1) {
        if (evenOc) {
          inv_r0_to_plane<true>(plane, b.Oc, lds, tw, tl);
        } else {
          inv_r0_to_plane<false>(plane, b.Oc, lds, tw, tl);
        }
      } else {
        if (evenOc) {
          inv16_to_plane<true>(plane, b.Oc, lds, tw, tl, A, B);
        } else {
          inv16_to_plane<false>(plane, b.Oc, lds, tw, tl, A, B);
        }
      }
      MI_SYNC();  // LDS free for the next phase
    }
  }

  // work item = (block, stream, channel group); it = (blk*streams + s)*groups + grp
  static MI_DEVICE void run(const Geometry &g, const IoDesc &io, const cf *MI_RESTRICT tw, const cf *MI_RESTRICT Wm,
                            const cf *MI_RESTRICT Gs, const cf *MI_RESTRICT Gc, cf *lds) {
    const int tid = MI_TID_X;
    const int local = MI_BID_X;
    const int item = io.item0 + local;
    const int per_blk = io.streams * io.groups;
    const int blk = item / per_blk;
    const int rem = item - blk * per_blk;
    const int s = rem / io.groups;
    const int c0 = (rem - s * io.groups) * io.cg;
    float *scr = io.scratch + static_cast<long long>(local) * io.cg * g.B;
    for (int cc = 0; cc < io.cg; ++cc) {
      const BlockIo b = make_block_io(g, io, s, c0 + cc, blk);
      int tc = tid;  // fresh copy per channel: keeps address arithmetic inside the loop body
      MI_OPAQUE_VGPR(tc);
      channel_block(g, io, b, scr + static_cast<long long>(cc) * g.B, tw, Wm, Gs, Gc, lds, tc);
    }
    // every plane store of this workgroup is complete and visible to it
    // (the loop ends in a workgroup barrier, which carries the release/acquire)
    epilogue(g, io, s, c0, blk, scr, tid);
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
