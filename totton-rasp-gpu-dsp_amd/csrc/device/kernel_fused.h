// Fused overlap-save upsampler kernel for gfx950: ONE workgroup = one
// (block, stream, channel group); everything between the PCM load and the PCM
// store of a channel-block stays in registers and LDS.
//
//   load   interleaved PCM frames (history + new) -> z[n] = x[2n] + j x[2n+1]
//   FFT_K  decimation-in-frequency, radices (R0,16,..,16), IN PLACE in LDS
//          (XOR-swizzled, bank-conflict free): natural order in, digit-reversed
//          order out; the first pass is fed straight from HBM, the last pass is
//          left in registers as two sixteen-bin sets {k + t*K/16}
//   split  real-FFT untangle of mirror pairs (k, K-k), kept in registers
//   for each output phase p (P = upsample factor):
//     multiply by the phase spectrum G_p (L2-resident, stored in thread order),
//     re-tangle, inverse FFT_K decimation-in-time (digit-reversed in, natural
//     out) through the same LDS buffer; the last pass writes y_p[n], n >= Oc,
//     phase-planar to the workgroup's fp32 staging planes
//   epilogue: staging planes of every channel of the group -> whole interleaved
//     PCM frames, lane-contiguous 16-byte stores.
//
// Because every pass is a true in-place butterfly (each thread reads and writes
// the same LDS words) a pass needs no barrier inside it, only one between
// passes, and waves drift apart inside a pass so LDS traffic of one wave
// overlaps butterflies of another.
//
// Replaces, per channel-block, the reference's ProcessBlock body
// (src/vulkan/vulkan_streaming_upsampler.cpp:528-572): zero-stuff/overlap
// assembly, pack, forward C2C FFT_N, CPU spectral multiply, inverse C2C FFT_N,
// gather and overlap update -- with the N-point transforms replaced by the exact
// polyphase identity: one K-point forward and P K-point inverse
// transforms, K = N / (2P).
//
// Thread layout: T = K/32 threads, each owns TWO radix-16 butterflies per
// radix-16 pass. After the forward transform LDS block b (words 16b..16b+15)
// holds the set S_a = {a + t*K/16}, a = digit-reverse(b). In the two passes
// adjacent to the spectral stage thread tau owns the blocks of S_a and S_{J-a}
// (J = K/16), mirror images under k -> K-k, so the untangle needs no data from
// another thread. Thread 0 owns the self-mirrored S_0 and S_{J/2}.
//
// PCM formats are a run-time property of the engine; the format switch is
// hoisted around the first forward pass and the epilogue (the only code that
// touches PCM), so no per-sample branch is executed.
#pragma once

#include "common.h"
#include "fft_radix.h"
#include "frame_tile.h"
#include "pcm.h"


// Diagnostic build only (-DMIUPS_STAMPS, library variant under lib_ablate/): lane 0 of
// every wave of the first 32 workgroups records s_memtime at fixed points so that
// scripts/stamps_report.py can print where a workgroup's cycles go. The stamps land in a
// buffer of their own; no product build contains them.
#if defined(MIUPS_STAMPS) && !defined(MIUPS_HOST_EMU)
__device__ unsigned long long mi_stamps[32][8][192];
#if !defined(MIUPS_STAMP_BASE)
#define MIUPS_STAMP_BASE 0  // first of the 32 recorded workgroups (a later round: -DMIUPS_STAMP_BASE=1024)
#endif
#define MI_STAMP(id)                                                                                   \
  do {                                                                                                 \
    if ((MI_TID_X & 63) == 0 && MI_TID_X < 512 && MI_BID_X >= MIUPS_STAMP_BASE && MI_BID_X < MIUPS_STAMP_BASE + 32) { \
      mi_stamps[MI_BID_X - MIUPS_STAMP_BASE][MI_TID_X >> 6][(id)] = __builtin_amdgcn_s_memtime();     \
    }                                                                                                  \
  } while (0)
#else
#define MI_STAMP(id)
#endif

#if !defined(MIUPS_WIDE_WAVES)
#define MIUPS_WIDE_WAVES 2  // experiment switch (profiles/): waves per SIMD the wide form is compiled for (3 = 168 registers)
#endif
#if !defined(MIUPS_ROWS_DEPTH)
#define MIUPS_ROWS_DEPTH 1  // experiment switch (profiles/): multiplies the frame epilogue's units in flight per thread
#endif
#if defined(MIUPS_EXP_SKIP_PAIR_SYNC)  // timing experiment (profiles/r02_d_*): no barrier inside a pair of passes (WRONG results)
#define MI_SYNC_PAIR() do {} while (0)
#else
#define MI_SYNC_PAIR() MI_SYNC()
#endif
namespace miups {

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

// exp(-2*pi*i*t/64), t = 0..16 (split form: M = 64 J)
MI_DEVICE cf w64(int t) {
  constexpr float c[17] = {1.0f,
                           0.99518472667219692873f,
                           0.98078528040323043058f,
                           0.95694033573220882438f,
                           0.92387953251128673848f,
                           0.88192126434835504956f,
                           0.83146961230254523567f,
                           0.77301045336273699338f,
                           0.70710678118654752440f,
                           0.63439328416364548779f,
                           0.55557023301960228867f,
                           0.47139673682599780857f,
                           0.38268343236508983729f,
                           0.29028467725446233105f,
                           0.19509032201612833135f,
                           0.09801714032956077016f,
                           0.0f};
  constexpr float s[17] = {0.0f,
                           0.09801714032956060363f,
                           0.19509032201612824808f,
                           0.29028467725446233105f,
                           0.38268343236508978178f,
                           0.47139673682599764204f,
                           0.55557023301960217765f,
                           0.63439328416364548779f,
                           0.70710678118654752440f,
                           0.77301045336273699338f,
                           0.83146961230254523567f,
                           0.88192126434835493853f,
                           0.92387953251128673848f,
                           0.95694033573220893540f,
                           0.98078528040323043058f,
                           0.99518472667219681771f,
                           1.0f};
  return mk(c[t], -s[t]);
}

// ---- mirror-pair algebra (see gen_multiply_kernel for the per-bin form) ---
//   xa = (u+v) - jW(u-v), xb = (u+v) + jW(u-v),  v = conj(zm)
// packed: with c = (1,-1), v = zm*c, s = u+v, e = u-v, jW e = (W e).yx * (-1,1) -> folded into the two fma's
MI_DEVICE void pair_split(cf u_, cf zm_, cf W_, cf &xa, cf &xb) {
  const v2 u = V(u_), W = V(W_);
  const v2 s = v2fma(V(zm_), v2mk(1.0f, -1.0f), u);   // u + conj(zm)
  const v2 e = v2fma(V(zm_), v2mk(-1.0f, 1.0f), u);   // u - conj(zm)
  const v2 d = v2swap(vmul(e, W));                    // (W e).yx ; j*(W e) = d * (-1, 1)
  xa = C(v2fma(d, v2mk(1.0f, -1.0f), s));              // s - j W e
  xb = C(v2fma(d, v2mk(-1.0f, 1.0f), s));              // s + j W e
}
//   P = xa*gs, Q = xb*gc ; zk = (P+Q) + j conj(W)(P-Q) ; zkm = conj((P+Q) - j conj(W)(P-Q))
MI_DEVICE void pair_phase(cf xa, cf xb, cf W_, f4 g, cf &zk, cf &zkm) {
  const v2 P = vmul(V(xa), v2mk(g.x, g.y));
  const v2 Q = vmul(V(xb), v2mk(g.z, g.w));
  const v2 S = P + Q;
  const v2 d = v2swap(vmulc(P - Q, V(W_)));           // (conj(W)(P-Q)).yx ; j*(..) = d * (-1, 1)
  zk = C(v2fma(d, v2mk(-1.0f, 1.0f), S));              // S + j conj(W)(P-Q)
  // conj(S - j conj(W)(P-Q)) = (S - D)*(1,-1), D = d*(-1,1):  S*(1,-1) + d*(1,1)... = S*(1,-1) + d
  zkm = C(v2fma(S, v2mk(1.0f, -1.0f), d));
}

// ---- format-typed sample load (p points AT the sample) ----------------------
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

// Per-channel-block input addressing (fused path requires S == 1, so compact
// sample n of this block is input frame f0 + n, f0 = blk*Bc - Oc; frames < 0
// are history).
struct BlockIo {
  const char *pin;    // where compact sample 0 would be in `in`   (valid for n >= n_hist)
  const char *phist;  // where compact sample 0 is in the history  (valid for n <  n_hist)
  // byte step between consecutive frames of one channel; the host only selects
  // the fused path when M * step < 2^31
  int in_step;
  int n_hist;
  int Oc;
  // 0: one load per sample. 1: mono, 4-byte samples, 8-byte aligned -> the two
  // samples of a complex input word are one 8-byte load. 2: stereo, 4-byte
  // samples, 16-byte aligned -> two whole frames are one 16-byte load (lanes read
  // consecutive 16-byte words: full cache lines) and the channel is picked from it.
  // 3: split-planar fp32 timeline (IoDesc::in_planar == 2): the complex words of transform half h are contiguous,
  // word q at pin + h * half_stride + 8 q -- lanes read consecutive 8-byte words.
  int vec_mode;
  long long half_stride;
  int chan;
  // split form (K = 2 * the LDS transform length): the transform in flight takes the complex
  // words 2n + half, i.e. compact samples 4n + noff, 4n + noff + 1 (noff = 0 or 2)
  int noff;
};

MI_DEVICE BlockIo make_block_io(const Geometry &g, const IoDesc &io, int s, int c, int blk) {
  BlockIo b;
  const long long ib = pcm_bytes(io.in_fmt);
  const long long f0 = static_cast<long long>(blk) * g.Bc - g.Oc;
  b.half_stride = 0;
  if (io.in_planar == 2) {
    // split-planar fp32 timeline (planarize_kernel with io.split_planes): sample i of the timeline sits at float
    // (i & 2 ? H : 0) + 2 (i >> 2) + (i & 1), H = half a plane. The host takes this layout only when every block's
    // window starts at a multiple of four samples.
    const long long i0 = g.hist_frames + f0;
    b.in_step = 4;
    b.pin = static_cast<const char *>(io.in) + s * io.in_stream_stride + c * io.in_plane_stride + (i0 >> 1) * 4;
    b.phist = b.pin;
    b.half_stride = io.in_plane_stride >> 1;
    b.n_hist = 0;
    b.Oc = g.Oc;
    b.chan = 0;
    b.noff = 0;
    b.vec_mode = 3;
    return b;
  }
  if (io.in_planar) {
    // planar fp32 timeline: index hist_frames + f holds frame f of this call
    b.in_step = 4;
    b.pin = static_cast<const char *>(io.in) + s * io.in_stream_stride + c * io.in_plane_stride +
            (g.hist_frames + f0) * 4;
    b.phist = b.pin;
    b.n_hist = 0;
    b.Oc = g.Oc;
    b.chan = 0;
    b.noff = 0;
    b.vec_mode = (reinterpret_cast<unsigned long long>(b.pin) & 7) == 0 ? 1 : 0;
    return b;
  }
  b.in_step = static_cast<int>(ib * io.channels);
  b.pin = static_cast<const char *>(io.in) + s * io.in_stream_stride + (f0 * io.channels + c) * ib;
  b.phist = static_cast<const char *>(io.hist) + s * io.hist_stream_stride +
            ((g.hist_frames + f0) * io.channels + c) * ib;
  b.n_hist = f0 >= 0 ? 0 : (-f0 > g.M ? g.M : static_cast<int>(-f0));
  b.Oc = g.Oc;
  b.chan = c;
  b.noff = 0;
  b.vec_mode = 0;
  if (b.n_hist == 0 && (io.in_fmt == kS32 || io.in_fmt == kF32)) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(b.pin);
    if (io.channels == 1 && (a & 7) == 0) {
      b.vec_mode = 1;
    } else if (io.channels == 2 && ((a - 4ull * c) & 15) == 0) {
      b.vec_mode = 2;
    }
  }
  return b;
}

// W = radix-16 butterflies per thread and pass.
//   W = 2 ("wide"):   T = K/32 threads with up to 256 registers: two waves per SIMD. A thread owns both sets of its
//                     mirror pairs. Used for K < 1024 and by the split form.
//   W = 1 ("narrow"): T = K/16 threads with at most 128 registers: FOUR waves per SIMD (K = 16384: one 1024-thread
//                     workgroup; K = 4096: four 256-thread workgroups per CU). Half the live state per thread, and a
//                     SIMD with four waves issues a VALU instruction every 1.72 cycles instead of every 2.47 with two
//                     (profiles/r02_a_ubench_valu_lds_rates.txt) -- the passes are VALU-issue bound. The two sets of a
//                     mirror pair live in lanes l and l ^ 32 of one wave and trade eight values through
//                     v_permlane32_swap (xchg32) on either side of the spectral product.
//
// Pass plans (forward order; the inverse runs them backwards):
//   classic: R0 = 2^(LOG2K mod 4), then radix 16 with strides .., 256, 16, 1      (K = 16384: 4 16 16 16)
//   R32:     K/512 (16 or 32, stride 512), radix 32 (stride 16), radix 16 (stride 1) -- K = 8192 and 16384 only,
//            wide form only: one pass, one LDS round trip and one barrier fewer per transform (16384: 32 32 16,
//            8192: 16 32 16). A thread's radix-32 butterfly is its two radix-16 butterflies joined by one radix-2
//            step in registers (fft_radix.h vdft32); the two passes next to the spectral stage are unchanged.
//            EXPERIMENT (MIUPS_EXP_R32=1 on a -DMIUPS_WITH_R32 build): measured 0.72x of the classic plan
//            (profiles/r03_b_radix32.txt) -- the radix-4 pass it removes is the cheap one (290 of ~2400 VALU
//            instructions per phase), an unshared 31-power twiddle tree per butterfly replaces a shared 15-power one,
//            and the 64-register butterfly beside the 68 registers of split spectrum spills ~70 registers.
template <int LOG2K, int W = 2, bool R32_ = false>
struct FusedCfg {
  static constexpr int K = 1 << LOG2K;
  static constexpr int J = K / 16;       // sixteen-word LDS blocks / radix-16 butterflies per pass
  static constexpr int T = K / (16 * W); // threads per workgroup
  static constexpr bool R32 = R32_;
  static constexpr int R0 = 1 << (LOG2K % 4);
  static constexpr int LOG2R0 = LOG2K % 4;
  static constexpr int N16 = LOG2K / 4;  // radix-16 passes
  // radix and stride of the pass next to HBM (first forward, last inverse); RF == 1: the classic plan's first pass is
  // its stride-K/16 radix-16 pass
  static constexpr int RF = R32 ? K / 512 : R0;
  // radix of the pass before the last one: its digit is the low digit of the block index
  static constexpr int RL = R32 ? 32 : ((N16 >= 2) ? 16 : R0);
  static_assert(!R32 || fused_plan_r32_exists(LOG2K, W), "the radix-32 plan covers the wide form at K = 8192, 16384");
  static constexpr int LDS_BYTES = K * 8;
  static constexpr int LDS_BYTES_SPLIT = K * 8 + 64 * 8;  // + FusedKernel::kXchWords
  static_assert(LOG2K >= 5 && LOG2K <= 14, "fused kernel covers K = 32 .. 16384");
  static_assert(W == 2 || (W == 1 && LOG2K >= 10), "the narrow form needs whole waves: K >= 1024");
  // LDS block of the thread's first set in the pairing passes: the tau-th block
  // whose low digit is < RL/2 (exactly the sets S_a with a < J/2)
  static MI_HD constexpr int block_a(int tau) { return (tau / (RL / 2)) * RL + (tau % (RL / 2)); }
};

// LDS word indices of the R words {base + t*S} of butterfly q in a pass with
// stride S (sub-transform length L = S*R): base = (q / S) * L + q % S.
// The four strides that occur (K/R0 = 16^N16, 256, 16, 1) each get the cheapest
// closed form of lds_swz(base + t*S).
// byte(t) = 8 * at(t) in the form that costs the fewest address instructions (the passes are VALU-issue bound and the
// generic form spent ~2 instructions per LDS access, a fifth of a pass): strides 1 and 16: one XOR with a constant
// per access; stride 256: the swizzle term takes four values, so four bases per butterfly and every
// access is base + immediate offset; stride >= 1024: one base + immediate offsets.
//   radix 32, stride 16 (sub-transform length 512): word a*512 + t*16 + r; every swizzle term is an XOR with a
//   constant of t or a bit of a -> byte(t) = p ^ C(t), one XOR per access;
//   stride 512 (K = 512 R, one sub-transform): the swizzle only sees t through bit 9 -> two bases + immediate offsets.
template <int R, int S>
struct Bfly {
  int p;  // precomputed per butterfly
  int w;  // S == 16 only
  int b4[S == 256 ? 4 : (S == 512 ? 2 : 1)];  // S == 256 / 512: byte address of (p ^ swizzle term k)
  MI_DEVICE explicit Bfly(int q) {
    if constexpr (S == 1) {
      p = lds_swz(16 * q);
      w = 0;
    } else if constexpr (S == 16 && R == 32) {
      const int a = q >> 4, r = q & 15;
      p = ((a * 512 + r) * 8) ^ ((a & 1) << 7);  // BYTE address of (a, r) with a's share of the swizzle
      w = 0;
    } else if constexpr (S == 512) {
      p = lds_swz(q);  // q < 512
      w = 0;
    } else if constexpr (S == 16) {
      const int a = q >> 4, r = q & 15;
      p = a * (16 * R);
      // the swizzle masks only depend on a (bits 8, 9 of the word index) and t
      w = (r ^ ((a & 1) << 3)) | (((a ^ (a >> 1)) & 1) << 4);
      if (R < 16) {
        w = r;  // R < 16 only happens for the first pass (a == 0)
      }
    } else {
      const int base = (q / S) * (S * R) + (q % S);
      p = lds_swz(base);
      w = 0;
    }
    if constexpr (S == 256) {
      MI_UNROLL
      for (int k = 0; k < 4; ++k) {
        b4[k] = (p ^ (((k & 1) << 3) | ((k >> 1) << 4))) * 8;
      }
    } else if constexpr (S == 512) {
      b4[0] = p * 8;
      b4[1] = (p ^ 16) * 8;
    } else {
      b4[0] = 0;
    }
  }
  // radix 32, stride 16: t*16 words, t's low four bits into the word's low four bits, t's bit 4 into bits 3 and 4
  static MI_DEVICE constexpr int c32(int t) { return (t * 128) ^ ((t & 15) * 8) ^ ((t >> 4) * 192); }
  MI_DEVICE int at(int t) const {
    if constexpr (S == 1) {
      return p ^ t;
    } else if constexpr (S == 16 && R == 32) {
      return (p ^ c32(t)) >> 3;
    } else if constexpr (S == 512) {
      return (b4[t & 1] >> 3) + t * S;
    } else if constexpr (S == 16) {
      return p + ((17 * t) ^ w);
    } else if constexpr (S == 256) {
      // bits 8, 9 of the word index are t's low bits: they flip bits 3 and 4
      return (p ^ (((t & 1) << 3) | (((t ^ (t >> 1)) & 1) << 4))) + t * S;
    } else {
      static_assert(S % 1024 == 0, "unexpected pass stride");
      return p + t * S;
    }
  }
  MI_DEVICE int byte(int t) const {
    if constexpr (S == 1) {
      return (p * 8) ^ (t * 8);
    } else if constexpr (S == 16 && R == 32) {
      return p ^ c32(t);
    } else if constexpr (S == 512) {
      return b4[t & 1] + t * (S * 8);
    } else if constexpr (S == 16) {
      return ((p + w) * 8) ^ (136 * t);  // p holds bits >= 8 only and (17 t) ^ w < 256: the sum is an XOR
    } else if constexpr (S == 256) {
      return b4[(t & 1) | (((t ^ (t >> 1)) & 1) << 1)] + t * (S * 8);
    } else {
      return p * 8 + t * (S * 8);
    }
  }
};

template <int LOG2K, int W = 2, bool R32 = false>
struct FusedKernel {
  using Cfg = FusedCfg<LOG2K, W, R32>;
  static constexpr int K = Cfg::K, J = Cfg::J, T = Cfg::T, N16 = Cfg::N16;
  // R0 / S0 below = radix and stride of the pass next to HBM under the kernel's plan (the classic plan's R0, or K/512)
  static constexpr int R0 = Cfg::RF;
  static constexpr int S0 = K / R0;  // stride of that pass (classic: 16^N16; R32: 512)

  static MI_DEVICE const cf &lds_ref(const cf *lds, int byte) {
    return *reinterpret_cast<const cf *>(reinterpret_cast<const char *>(lds) + byte);
  }
  static MI_DEVICE cf &lds_ref(cf *lds, int byte) { return *reinterpret_cast<cf *>(reinterpret_cast<char *>(lds) + byte); }
  template <int R, int S>
  static MI_DEVICE void lds_get(const cf *lds, const Bfly<R, S> &b, cf *v) {
    MI_UNROLL
    for (int t = 0; t < R; ++t) {
      v[t] = lds_ref(lds, b.byte(t));
    }
  }
  // natural order in v
  template <int R, int S>
  static MI_DEVICE void lds_put(cf *lds, const Bfly<R, S> &b, const cf *v) {
    MI_UNROLL
    for (int t = 0; t < R; ++t) {
      lds_ref(lds, b.byte(t)) = v[t];
    }
  }
  // v holds dftR output (element u at v[out_pos<R>(u)])
  template <int R, int S>
  static MI_DEVICE void lds_put_dft(cf *lds, const Bfly<R, S> &b, const cf *v) {
    MI_UNROLL
    for (int u = 0; u < R; ++u) {
      lds_ref(lds, b.byte(u)) = v[out_pos<R>(u)];
    }
  }

  // One butterfly's DFT and the store of its outputs, four at a time as the last butterfly stage produces them
  // (fft_radix.h vdft16_emit): the LDS write path (~80 B/clk) is the slowest resource of a pass, and sixteen stores
  // issued in one burst behind a thread's last butterfly drain while the whole workgroup waits at the barrier.
  // kOut: output u is multiplied by w^u first (decimation in frequency); tw[u - 1] = w^u for u < 16, and for radix 32
  // w16 = w^16 (the powers above sixteen are formed where they are used: sixteen live powers instead of thirty-one).
  template <int DIR, int R, int S, bool kOut>
  static MI_DEVICE void dft_put(cf *lds, const Bfly<R, S> &b, const cf *c, const v2 *tw, v2 w16 = v2{1.0f, 0.0f}) {
    static_assert(R == 16 || R == 32, "emitting DFTs exist for radix 16 and 32");
    v2 v[R];
    MI_UNROLL
    for (int i = 0; i < R; ++i) {
      v[i] = V(c[i]);
    }
    auto emit = [&](int u, v2 y) {
      if constexpr (kOut) {
        if (u >= 1 && u < 16) {
          y = vmul(y, tw[u - 1]);
        } else if (u == 16) {
          y = vmul(y, w16);
        } else if (u > 16) {
          y = vmul(y, vmul(w16, tw[u - 17]));
        }
      }
      lds_ref(lds, b.byte(u)) = C(y);
    };
#if defined(MIUPS_EXP_NO_TRICKLE)  // experiment switch (profiles/r03_b_*): all stores behind the whole butterfly
    vdftR<DIR, R>(v);
    MI_UNROLL
    for (int u = 0; u < R; ++u) {
      emit(u, v[out_pos<R>(u)]);
    }
#else
    if constexpr (R == 16) {
      vdft16_emit<DIR>(v, emit);
    } else {
      vdft32_emit<DIR>(v, emit);
    }
#endif
  }

  template <int LOG2L>
  static MI_DEVICE cf load_tw(const cf *tw, int r) {
    return tw[tw_offset(LOG2L) + r];
  }

  // ---- global load of one radix-R butterfly's inputs (forward pass 0) -----
  // kHist = false: the whole block lies in `in` (true for all but the first
  // blocks of a call), one uniform base + 32-bit offsets.
  // MODE: BlockIo::vec_mode (1 and 2 imply !kHist and a 4-byte format)
  // SP: split form, word q of this transform is complex word 2q + noff/2 of the block
  template <int FMT, int R, bool kHist, int MODE, int SP = 0>
  static MI_DEVICE void global_read(const BlockIo &b, int q, cf *v) {
    MI_UNROLL
    for (int t = 0; t < R; ++t) {
      const int n = SP ? 4 * (q + t * (K / R)) + b.noff : 2 * (q + t * (K / R));
      if constexpr (MODE == 3) {
        static_assert(MODE != 3 || (SP == 1 && FMT == kF32), "split-planar input is fp32 and belongs to the split form");
        const cf w = *reinterpret_cast<const cf *>(b.pin + (b.noff >> 1) * b.half_stride +
                                                   static_cast<unsigned>(q + t * (K / R)) * 8u);
        v[t] = w;
        continue;
      } else if constexpr (MODE == 1) {
        struct alignas(8) W2 {
          int32_t a, b;
        };
        const W2 w = *reinterpret_cast<const W2 *>(b.pin + static_cast<unsigned>(n) * 4u);
        if constexpr (FMT == kF32) {
          v[t] = mk(__builtin_bit_cast(float, w.a), __builtin_bit_cast(float, w.b));
        } else {
          v[t] = mk(static_cast<float>(w.a) * (1.0f / 2147483648.0f), static_cast<float>(w.b) * (1.0f / 2147483648.0f));
        }
        continue;
      } else if constexpr (MODE == 2) {
        struct alignas(16) W4 {
          int32_t a, b, c, d;
        };
        const W4 w = *reinterpret_cast<const W4 *>(b.pin - 4 * b.chan + static_cast<unsigned>(n) * 8u);
        const int32_t lo = b.chan ? w.b : w.a, hi = b.chan ? w.d : w.c;
        if constexpr (FMT == kF32) {
          v[t] = mk(__builtin_bit_cast(float, lo), __builtin_bit_cast(float, hi));
        } else {
          v[t] = mk(static_cast<float>(lo) * (1.0f / 2147483648.0f), static_cast<float>(hi) * (1.0f / 2147483648.0f));
        }
        continue;
      }
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
  // The vector modes in two steps, so that a first pass can request ALL of a thread's input words before it converts
  // the first one (fwd_first): Raw = the loaded bytes (MODE 2: two whole stereo frames, else one complex word).
  template <int MODE>
  struct alignas(MODE == 2 ? 16 : 8) Raw {
    int32_t w[MODE == 2 ? 4 : 2];
  };
  template <int R, int MODE, int SP>
  static MI_DEVICE void global_fetch(const BlockIo &b, int q, Raw<MODE> *r) {
    MI_UNROLL
    for (int t = 0; t < R; ++t) {
      const unsigned word = static_cast<unsigned>(q + t * (K / R));  // complex word of this transform
      const unsigned n = SP ? 4u * word + static_cast<unsigned>(b.noff) : 2u * word;
      if constexpr (MODE == 3) {
        r[t] = *reinterpret_cast<const Raw<MODE> *>(b.pin + (b.noff >> 1) * b.half_stride + word * 8u);
      } else if constexpr (MODE == 1) {
        r[t] = *reinterpret_cast<const Raw<MODE> *>(b.pin + n * 4u);
      } else {
        r[t] = *reinterpret_cast<const Raw<MODE> *>(b.pin - 4 * b.chan + n * 8u);
      }
    }
  }
  template <int FMT, int R, int MODE>
  static MI_DEVICE void global_unpack(const BlockIo &b, const Raw<MODE> *r, cf *v) {
    MI_UNROLL
    for (int t = 0; t < R; ++t) {
      const int32_t lo = MODE == 2 ? (b.chan ? r[t].w[1] : r[t].w[0]) : r[t].w[0];
      const int32_t hi = MODE == 2 ? (b.chan ? r[t].w[3] : r[t].w[2]) : r[t].w[1];
      if constexpr (FMT == kF32) {
        v[t] = mk(__builtin_bit_cast(float, lo), __builtin_bit_cast(float, hi));
      } else {
        v[t] = mk(static_cast<float>(lo) * (1.0f / 2147483648.0f), static_cast<float>(hi) * (1.0f / 2147483648.0f));
      }
    }
  }
  // ---- store of one radix-R butterfly's outputs (inverse last pass) ---------
  // overlap-discard (:566-569): compact samples n < Oc are dropped; the kept
  // ones go, still fp32 and phase-planar, to this workgroup's staging plane
  // (plane[i] = y_p[Oc + i]) with lane-contiguous 8-byte stores.
  // Even history length (every shipped geometry): output u of butterfly q is the complex word n2 = q + u*K/R of
  // y_p and is kept iff n2 >= Oc/2; it goes to byte (n2 - Oc/2)*8 of the plane. The plane is addressed through a
  // buffer descriptor of exactly its kept size and the offset is formed in UNSIGNED arithmetic: a discarded word's
  // offset wraps past the end and the hardware drops the store. No compare, no exec-mask branch, no 64-bit address
  // per store (the compare-and-branch form cost 12 instructions and a 5-cycle s_nop per store: the base pointer sat
  // in VGPRs and was re-read with v_readfirstlane at every store).
  struct PlaneDst {
#if defined(MIUPS_HOST_EMU)
    float *plane;
    unsigned nkeep;
#else
    __amdgpu_buffer_rsrc_t rsrc;
#endif
    unsigned o2x8;  // (Oc/2) * 8
    float *raw;     // odd history length: plain pointer path
    int Oc;
  };
  static MI_DEVICE PlaneDst make_plane_dst(float *plane, int Oc, int nkeep) {
    PlaneDst d;
    d.raw = plane;
    d.Oc = Oc;
    d.o2x8 = (static_cast<unsigned>(Oc) >> 1) * 8u;
#if defined(MIUPS_HOST_EMU)
    d.plane = plane;
    d.nkeep = static_cast<unsigned>(nkeep);
#else
    // the plane pointer is workgroup-uniform; tell the compiler (it keeps it in VGPRs otherwise)
    const unsigned long long a = reinterpret_cast<unsigned long long>(plane);
    const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(a));
    const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(a >> 32));
    float *uniform = reinterpret_cast<float *>((static_cast<unsigned long long>(hi) << 32) | lo);
    d.rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform, 0, __builtin_amdgcn_readfirstlane(nkeep) * 4, 0x00020000);
#endif
    return d;
  }
  // kUpper: only outputs u >= R/2 exist in v (pruned last pass: the lower half of the transform is discarded history)
  template <int R, bool kEvenOc, bool kNT = false, bool kUpper = false>
  static MI_DEVICE void plane_write(const PlaneDst &d, int q, const cf *v) {
    if constexpr (kEvenOc) {
      MI_UNROLL
      for (int u = kUpper ? R / 2 : 0; u < R; ++u) {
        const unsigned off = (static_cast<unsigned>(q) + static_cast<unsigned>(u) * (K / R)) * 8u - d.o2x8;
        const cf y = v[out_pos<R>(u)];
#if defined(MIUPS_HOST_EMU)
        if (off < d.nkeep * 4u) {  // what the buffer range check does
          *reinterpret_cast<cf *>(reinterpret_cast<char *>(d.plane) + off) = y;
        }
#else
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        const u2 bits = {__builtin_bit_cast(unsigned, y.x), __builtin_bit_cast(unsigned, y.y)};
        // kNT: planes that another kernel turns into frames are stored with the streaming policy (aux 2 = nt), so that
        // they do not push the phase spectra out of L2 (config 4 +5 %, config 5 +3 %)
        __builtin_amdgcn_raw_buffer_store_b64(bits, d.rsrc, static_cast<int>(off), 0, kNT ? 2 : 0);
#endif
      }
    } else {
      float *plane = d.raw;
      const int Oc = d.Oc;
      MI_UNROLL
      for (int u = 0; u < R; ++u) {
        const int n = 2 * (q + u * (K / R));
        const cf y = v[out_pos<R>(u)];
        if (n >= Oc) {
          plane[n - Oc] = y.x;
        }
        if (n + 1 >= Oc) {
          plane[n + 1 - Oc] = y.y;
        }
      }
    }
  }

  // ================= forward (decimation in frequency) ======================
  // pass 0 from HBM: y_u = DFT_R(x)_u * W_K^(u*q), written to q + u*K/R
  template <int FMT, bool kHist, int MODE, int SP = 0>
  static MI_DEVICE void fwd_first(const BlockIo &b, cf *lds, const cf *tw, int tid) {
    // W_K^(tid + i*T) = W_K^tid * W_(16W)^i : one table load for all butterflies
    const cf w0 = load_tw<LOG2K>(tw, tid);
    if constexpr (R0 > 1 && MODE != 0) {
      // every input word of the thread (16 W of them) is requested before the first one is converted -- the fence pins
      // that order; the scheduler otherwise interleaves loads and conversions as register pressure suggests to it and
      // the round trips queue up (same source, other builds: 12.3k .. 16.3k cycles, profiles/r02_d_*)
      constexpr int NB = 16 * W / R0;
      Raw<MODE> in[NB][R0];
      MI_UNROLL
      for (int i = 0; i < NB; ++i) {
        global_fetch<R0, MODE, SP>(b, tid + i * T, in[i]);
      }
      MI_SCHED_FENCE();
      MI_UNROLL
      for (int i = 0; i < NB; ++i) {
        const int q = tid + i * T;
        cf v[R0];
        global_unpack<FMT, R0, MODE>(b, in[i], v);
        const cf wq = i == 0 ? w0 : cmul(w0, w32(i * (2 / W)));
        if constexpr (R0 >= 16) {
          v2 t[15];
          make_twiddles<-1, 16>(V(wq), t);
          dft_put<-1, R0, S0, true>(lds, Bfly<R0, S0>(q), v, t, vmul(t[7], t[7]));
        } else {
          dftR<-1, R0>(v);
          apply_twiddles_out<-1, R0>(v, wq);
          lds_put_dft<R0, S0>(lds, Bfly<R0, S0>(q), v);
        }
      }
    } else if constexpr (R0 > 1) {
      // every input load of the thread is issued before the first butterfly, so the
      // HBM round trip is paid once, not once per butterfly
      constexpr int NB = 16 * W / R0;
      cf raw[NB][R0];
      MI_UNROLL
      for (int i = 0; i < NB; ++i) {
        global_read<FMT, R0, kHist, MODE, SP>(b, tid + i * T, raw[i]);
      }
      MI_UNROLL
      for (int i = 0; i < NB; ++i) {
        const int q = tid + i * T;
        const cf wq = i == 0 ? w0 : cmul(w0, w32(i * (2 / W)));
        if constexpr (R0 >= 16) {
          v2 t[15];
          make_twiddles<-1, 16>(V(wq), t);
          dft_put<-1, R0, S0, true>(lds, Bfly<R0, S0>(q), raw[i], t, vmul(t[7], t[7]));
        } else {
          dftR<-1, R0>(raw[i]);
          apply_twiddles_out<-1, R0>(raw[i], wq);
          lds_put_dft<R0, S0>(lds, Bfly<R0, S0>(q), raw[i]);
        }
      }
    } else if constexpr (W == 2 && MODE != 0) {
      Raw<MODE> ia[16], ib[16];
      global_fetch<16, MODE, SP>(b, tid, ia);
      global_fetch<16, MODE, SP>(b, tid + T, ib);
      MI_SCHED_FENCE();
      cf A[16], B[16];
      global_unpack<FMT, 16, MODE>(b, ia, A);
      {
        Tw16 ta;
        make_twiddles16<-1>(w0, ta);
        dft_put<-1, 16, K / 16, true>(lds, Bfly<16, K / 16>(tid), A, ta.t);
      }
      MI_SCHED_FENCE();
      global_unpack<FMT, 16, MODE>(b, ib, B);
      Tw16 tb;
      make_twiddles16<-1>(cmul(w0, w32(1)), tb);
      dft_put<-1, 16, K / 16, true>(lds, Bfly<16, K / 16>(tid + T), B, tb.t);
    } else if constexpr (W == 1) {
      cf A[16];
      global_read<FMT, 16, kHist, MODE, SP>(b, tid, A);
      Tw16 ta;
      make_twiddles16<-1>(w0, ta);
      dft_put<-1, 16, K / 16, true>(lds, Bfly<16, K / 16>(tid), A, ta.t);
    } else {
      cf A[16], B[16];
      global_read<FMT, 16, kHist, MODE, SP>(b, tid, A);
      global_read<FMT, 16, kHist, MODE, SP>(b, tid + T, B);
      {
        Tw16 ta;
        make_twiddles16<-1>(w0, ta);
        dft_put<-1, 16, K / 16, true>(lds, Bfly<16, K / 16>(tid), A, ta.t);
      }
      MI_SCHED_FENCE();
      Tw16 tb;
      make_twiddles16<-1>(cmul(w0, w32(1)), tb);
      dft_put<-1, 16, K / 16, true>(lds, Bfly<16, K / 16>(tid + T), B, tb.t);
    }
  }
  template <int FMT, int SP = 0>
  static MI_DEVICE void fwd_first_fmt(const BlockIo &b, cf *lds, const cf *tw, int tid) {
    if constexpr (FMT == kF32 && SP == 1) {
      if (b.vec_mode == 3) {
        fwd_first<FMT, false, 3, SP>(b, lds, tw, tid);
        return;
      }
    }
    if constexpr (FMT == kS32 || FMT == kF32) {
      if (b.vec_mode == 2) {
        fwd_first<FMT, false, 2, SP>(b, lds, tw, tid);
        return;
      }
      if (b.vec_mode == 1) {
        fwd_first<FMT, false, 1, SP>(b, lds, tw, tid);
        return;
      }
    }
    if (b.n_hist == 0) {
      fwd_first<FMT, false, 0, SP>(b, lds, tw, tid);
    } else {
      fwd_first<FMT, true, 0, SP>(b, lds, tw, tid);
    }
  }

  // middle radix-16 pass with stride S (16 or 256), in place
  template <int S>
  static MI_DEVICE void fwd_mid(cf *lds, const cf *tw, int tid) {
    constexpr int LOG2L = (S == 16) ? 8 : 12;
    if constexpr (W == 1) {
      const Bfly<16, S> bf(tid);
      cf X[16];
      lds_get<16, S>(lds, bf, X);
      Tw16 t16;
      make_twiddles16<-1>(load_tw<LOG2L>(tw, tid & (S - 1)), t16);
      dft_put<-1, 16, S, true>(lds, bf, X, t16.t);
      return;
    }
    const int qA = tid, qB = tid + T;
    const cf wA = load_tw<LOG2L>(tw, qA & (S - 1));
    const cf wB = (T % S == 0) ? wA : load_tw<LOG2L>(tw, qB & (S - 1));
    const Bfly<16, S> bA(qA), bB(qB);
    cf A[16], B[16];
    lds_get<16, S>(lds, bA, A);
    lds_get<16, S>(lds, bB, B);
    if constexpr (T % S == 0) {  // both butterflies use the same twiddle: one power tree (-9 % VALU in this pass)
      Tw16 tw16;
      make_twiddles16<-1>(wA, tw16);
      dft_put<-1, 16, S, true>(lds, bA, A, tw16.t);
      MI_SCHED_FENCE();
      dft_put<-1, 16, S, true>(lds, bB, B, tw16.t);
      return;
    }
    {
      Tw16 twA;
      make_twiddles16<-1>(wA, twA);
      dft_put<-1, 16, S, true>(lds, bA, A, twA.t);
    }
    MI_SCHED_FENCE();
    Tw16 twB;
    make_twiddles16<-1>(wB, twB);
    dft_put<-1, 16, S, true>(lds, bB, B, twB.t);
  }
  // last pass (stride 1, no twiddles): the thread's two sets stay in registers,
  // A[out_pos<16>(t)] = Z[a + t*J], B[out_pos<16>(t)] = Z[(J - a) + t*J]
  static MI_DEVICE void fwd_last(const cf *lds, int blkA, int blkB, cf *A, cf *B) {
    lds_get<16, 1>(lds, Bfly<16, 1>(blkA), A);
    lds_get<16, 1>(lds, Bfly<16, 1>(blkB), B);
    dft16<-1>(A);
    MI_SCHED_FENCE();
    dft16<-1>(B);
  }

  // ================= inverse (decimation in time) ===========================
  // first pass (stride 1): inputs in A/B (natural order), results into the
  // thread's own two LDS blocks
  static MI_DEVICE void inv_first(cf *lds, int blkA, int blkB, cf *A, cf *B) {
    dft_put<+1, 16, 1, false>(lds, Bfly<16, 1>(blkA), A, nullptr);
    MI_SCHED_FENCE();
    dft_put<+1, 16, 1, false>(lds, Bfly<16, 1>(blkB), B, nullptr);
  }
  template <int S>
  static MI_DEVICE void inv_mid(cf *lds, const cf *tw, int tid) {
    constexpr int LOG2L = (S == 16) ? 8 : 12;
    if constexpr (W == 1) {
      const Bfly<16, S> bf(tid);
      cf X[16];
      lds_get<16, S>(lds, bf, X);
      apply_twiddles<+1, 16>(X, load_tw<LOG2L>(tw, tid & (S - 1)));
      dft_put<+1, 16, S, false>(lds, bf, X, nullptr);
      return;
    }
    const int qA = tid, qB = tid + T;
    const cf wA = load_tw<LOG2L>(tw, qA & (S - 1));
    const cf wB = (T % S == 0) ? wA : load_tw<LOG2L>(tw, qB & (S - 1));
    const Bfly<16, S> bA(qA), bB(qB);
    cf A[16], B[16];
    lds_get<16, S>(lds, bA, A);
    lds_get<16, S>(lds, bB, B);
    if constexpr (T % S == 0) {
      Tw16 tw16;
      make_twiddles16<+1>(wA, tw16);
      mul_twiddles16(A, tw16);
      dft_put<+1, 16, S, false>(lds, bA, A, nullptr);
      MI_SCHED_FENCE();
      mul_twiddles16(B, tw16);
      dft_put<+1, 16, S, false>(lds, bB, B, nullptr);
      return;
    }
    apply_twiddles<+1, 16>(A, wA);
    dft_put<+1, 16, S, false>(lds, bA, A, nullptr);
    MI_SCHED_FENCE();
    apply_twiddles<+1, 16>(B, wB);
    dft_put<+1, 16, S, false>(lds, bB, B, nullptr);
  }
  // R32 plan: the one middle pass, radix 32 with stride 16 (sub-transform length 512), one butterfly per thread
  static MI_DEVICE void fwd_mid32(cf *lds, const cf *tw, int tid) {
    const Bfly<32, 16> bf(tid);
    cf X[32];
    lds_get<32, 16>(lds, bf, X);
    v2 t[15];
    make_twiddles<-1, 16>(V(load_tw<9>(tw, tid & 15)), t);
    dft_put<-1, 32, 16, true>(lds, bf, X, t, vmul(t[7], t[7]));
  }
  static MI_DEVICE void inv_mid32(cf *lds, const cf *tw, int tid) {
    const Bfly<32, 16> bf(tid);
    cf X[32];
    lds_get<32, 16>(lds, bf, X);
    apply_twiddles<+1, 32>(X, load_tw<9>(tw, tid & 15));
    dft_put<+1, 32, 16, false>(lds, bf, X, nullptr);
  }
  // last pass: stride K/R, natural-order results -> staging plane
  // kUpper (even history length only): the discarded history covers at least the lower half of the transform (Oc >= K,
  // i.e. O/N >= 1/2: every shipped filter), so only the upper half of every butterfly's outputs is computed and stored
  // (fft_radix.h vdftR_upper). The hardware range check still drops the discarded part of the upper half.
  template <bool kEvenOc, bool kNT = false, bool kUpper = false>
  static MI_DEVICE void inv_last(float *plane_, int Oc, int nkeep, const cf *lds, const cf *tw, int tid) {
    static_assert(!kUpper || (kEvenOc && W == 2 && !R32), "the pruned last pass exists for the wide classic form, even history");
    const PlaneDst plane = make_plane_dst(plane_, Oc, nkeep);
    const cf w0 = load_tw<LOG2K>(tw, tid);
    if constexpr (R0 > 1) {
      MI_UNROLL
      for (int i = 0; i < 16 * W / R0; ++i) {
        const int q = tid + i * T;
        cf v[R0];
        lds_get<R0, S0>(lds, Bfly<R0, S0>(q), v);
        apply_twiddles<+1, R0>(v, i == 0 ? w0 : cmul(w0, w32(i * (2 / W))));
        if constexpr (kUpper) {
          dftR_upper<+1, R0>(v);
        } else {
          dftR<+1, R0>(v);
        }
        plane_write<R0, kEvenOc, kNT, kUpper>(plane, q, v);
        if ((i & 3) == 3) {
          MI_SCHED_FENCE();  // keep at most 4 butterflies' registers in flight
        }
      }
    } else if constexpr (kUpper) {
      cf A[16], B[16];
      lds_get<16, K / 16>(lds, Bfly<16, K / 16>(tid), A);
      lds_get<16, K / 16>(lds, Bfly<16, K / 16>(tid + T), B);
      apply_twiddles<+1, 16>(A, w0);
      dftR_upper<+1, 16>(A);
      plane_write<16, kEvenOc, kNT, true>(plane, tid, A);
      MI_SCHED_FENCE();
      apply_twiddles<+1, 16>(B, cmul(w0, w32(1)));
      dftR_upper<+1, 16>(B);
      plane_write<16, kEvenOc, kNT, true>(plane, tid + T, B);
    } else if constexpr (W == 1) {
      cf A[16];
      lds_get<16, K / 16>(lds, Bfly<16, K / 16>(tid), A);
      apply_twiddles<+1, 16>(A, w0);
      dft16<+1>(A);
      plane_write<16, kEvenOc, kNT>(plane, tid, A);
    } else {
      cf A[16], B[16];
      lds_get<16, K / 16>(lds, Bfly<16, K / 16>(tid), A);
      lds_get<16, K / 16>(lds, Bfly<16, K / 16>(tid + T), B);
      apply_twiddles<+1, 16>(A, w0);
      dft16<+1>(A);
      plane_write<16, kEvenOc, kNT>(plane, tid, A);
      MI_SCHED_FENCE();
      apply_twiddles<+1, 16>(B, cmul(w0, w32(1)));
      dft16<+1>(B);
      plane_write<16, kEvenOc, kNT>(plane, tid + T, B);
    }
  }

  // ================= spectral stage ==========================================
  // A thread's sixteen mirror pairs: pair t = bins k = a + t*J and K - k = (J - a) + (15 - t)*J, with the untangle
  // twiddle W_M^k = Wa * W_32^t. On entry Xa/Xb hold the split spectrum, on exit A/B hold the inputs of the first
  // inverse pass in natural order (element t of the thread's two blocks).
  //
  // Thread 0 owns the two self-mirrored sets S_0 and S_{J/2}: 9 + 8 pairs INSIDE each set. It runs the same sixteen
  // slots as every other thread -- slots 0..8 = the pairs of S_0 (its Wa is 1), slots 9..15 = the first seven pairs of
  // S_{J/2} (twiddle base Wself = W_M^(J/2) / W_32^9 instead of Wa, so that Wself * W_32^s is the right twiddle; the
  // host stores lane 0's spectra in column 0 of the same table) -- plus ONE extra pair and a register permutation under
  // `tid == 0`. (A separate code path for thread 0 made wave 0 execute both paths at every phase: twice the spectral
  // stage on the critical path of every barrier, half of the whole workgroup at K <= 4096.)
  static MI_DEVICE cf slot_twiddle(int t, cf Wa, cf Wa2) { return cmul(t < 9 ? Wa : Wa2, w32(t)); }

  static MI_DEVICE void phase_inputs(int tid, const cf *Xa, const cf *Xb, cf Wa, cf Wa2, cf Wb, const f4 *MI_RESTRICT gt,
                                     const f4 *MI_RESTRICT g0, cf *A, cf *B) {
    // pair t: k = a + t*J  <->  K-k = (J-a) + (15-t)*J
    // The sixteen table words in four groups of four, group g+1 requested before group g is used (two groups = 32
    // registers in flight), each group's loads fenced ahead of the arithmetic: left to itself the scheduler sometimes
    // issues load, wait, use, load, wait, use -- sixteen exposed L2 round trips, 3.6k -> 6.8k cycles per phase
    // (profiles/r02_d_phase_loads.txt; which schedule came out depended on unrelated code elsewhere in the kernel).
    const f4 *pg = gt + tid;
    f4 gv[2][4];
    MI_UNROLL
    for (int j = 0; j < 4; ++j) {
      gv[0][j] = pg[j * T];
    }
    MI_UNROLL
    for (int q = 0; q < 4; ++q) {
      if (q < 3) {
        MI_UNROLL
        for (int j = 0; j < 4; ++j) {
          gv[(q + 1) & 1][j] = pg[(4 * (q + 1) + j) * T];
        }
      }
      MI_SCHED_FENCE();
      MI_UNROLL
      for (int j = 0; j < 4; ++j) {
        const int t = 4 * q + j;
        pair_phase(Xa[t], Xb[t], slot_twiddle(t, Wa, Wa2), gv[q & 1][j], A[t], B[15 - t]);
      }
      MI_SCHED_FENCE();
    }
    if (tid == 0) {
      // slot s <= 8: zk is A[s] already, zkm belongs at A[16-s] (s = 1..7; DC and the self-mirrored bin have none);
      // slot s >= 9 (pair s-9 of S_{J/2}): zk belongs at B[s-9], zkm at B[24-s]; the eighth pair is the extra one
      cf ez, em;
      pair_phase(Xa[16], Xb[16], cmul(Wb, w32(7)), g0[16], ez, em);
      cf keep[7];
      MI_UNROLL
      for (int j = 0; j < 7; ++j) {
        keep[j] = B[j];  // zkm of slots 15..9
      }
      MI_UNROLL
      for (int i = 9; i < 16; ++i) {
        const cf zk = A[i];
        A[i] = B[i - 1];  // zkm of slot 16-i
        B[i - 9] = zk;
      }
      MI_UNROLL
      for (int j = 9; j < 16; ++j) {
        B[j] = keep[j - 9];  // zkm of pair t' = 15-j of S_{J/2} (slot 9+t') sat in B[15 - (9 + t')] = B[j - 9]
      }
      B[7] = ez;
      B[8] = em;
    }
  }

  static MI_DEVICE void split_spectrum(int tid, cf *A, cf *B, cf Wa, cf Wa2, cf Wb, cf *Xa, cf *Xb) {
    Xa[16] = Xb[16] = mk(0.0f, 0.0f);
    if (tid == 0) {
      // lane 0: arrange its two sets as the sixteen (u, mirror) slots of the generic loop; the 17th pair on the side
      pair_split(B[out_pos<16>(7)], B[out_pos<16>(8)], cmul(Wb, w32(7)), Xa[16], Xb[16]);
      cf a0[16], b0[16];
      MI_UNROLL
      for (int t = 0; t < 16; ++t) {
        a0[t] = A[out_pos<16>(t)];
        b0[t] = B[out_pos<16>(t)];
      }
      MI_UNROLL
      for (int s = 9; s < 16; ++s) {
        A[out_pos<16>(s)] = b0[s - 9];           // u of slot s: element s-9 of S_{J/2}
      }
      MI_UNROLL
      for (int j = 0; j < 7; ++j) {
        B[out_pos<16>(j)] = b0[j + 9];           // mirror of slot 15-j (>= 9): element 15-(15-j-9) = j+9 of S_{J/2}
      }
      MI_UNROLL
      for (int j = 7; j < 15; ++j) {
        B[out_pos<16>(j)] = a0[j + 1];           // mirror of slot 15-j (1..8): element 16-(15-j) of S_0
      }
      B[out_pos<16>(15)] = a0[0];                // slot 0 pairs DC with itself (Nyquist rides in gc)
    }
    MI_UNROLL
    for (int t = 0; t < 16; ++t) {
      pair_split(A[out_pos<16>(t)], B[out_pos<16>(15 - t)], slot_twiddle(t, Wa, Wa2), Xa[t], Xb[t]);
    }
  }

  // ================= spectral stage, narrow form (W = 1) =====================
  // Lane l of a wave holds the sixteen bins own[t] = Z[k0 + t*J] of ONE set S_k0; its partner lane l ^ 32 holds the
  // mirror set S_(J-k0). Mirror of own[j] is the partner's own[15-j], so each lane takes the eight pairs of its own
  // bins j = 0..7 (the partner takes the other eight): it sends own[15..8], receives the partner's, untangles /
  // multiplies / re-tangles its eight pairs with twiddle W_M^(k0 + j*J) = Wl * W_32^j, keeps zk (its own bin j) and
  // sends zkm (the partner's bin 15-j) back. Both lanes run the same instructions on the same register names.
  // The two self-mirrored sets sit in lanes 0 (S_0: pairs (t, 16-t), nine of them) and 32 (S_{J/2}: pairs
  // (t, 15-t)) of wave 0; they take part in the exchanges like everyone and then replace what they received by a
  // permutation of their own values (lane 0 also computes its ninth pair) under `tid == 0` / `tid == 32`.
  static MI_DEVICE void split_spectrum_n(int tid, const cf *A, cf Wl, cf *Xa, cf *Xb) {
    cf recv[8];
    MI_UNROLL
    for (int j = 0; j < 8; ++j) {
      recv[j] = xchg32(A[out_pos<16>(15 - j)]);
    }
    Xa[8] = Xb[8] = mk(0.0f, 0.0f);
    if (tid == 0) {
      // S_0: the mirror of own[j] is own[(16-j) & 15]; own[8] is its own mirror (ninth pair)
      recv[0] = A[out_pos<16>(0)];
      MI_UNROLL
      for (int j = 1; j < 8; ++j) {
        recv[j] = A[out_pos<16>(16 - j)];
      }
      pair_split(A[out_pos<16>(8)], A[out_pos<16>(8)], w32(8), Xa[8], Xb[8]);
    } else if (tid == 32) {
      MI_UNROLL
      for (int j = 0; j < 8; ++j) {
        recv[j] = A[out_pos<16>(15 - j)];  // S_{J/2}: the mirror of own[j] is own[15-j]
      }
    }
    MI_UNROLL
    for (int j = 0; j < 8; ++j) {
      pair_split(A[out_pos<16>(j)], recv[j], cmul(Wl, w32(j)), Xa[j], Xb[j]);
    }
  }
  // gt: this phase's [8][T] table, g0: this phase's entry of lane 0's ninth pair; V: inputs of the first inverse
  // pass in natural order
  static MI_DEVICE void phase_inputs_n(int tid, const cf *Xa, const cf *Xb, cf Wl, const f4 *MI_RESTRICT gt,
                                       const f4 *MI_RESTRICT g0, cf *V) {
    const f4 *pg = gt + tid;
    cf zkm[8];
    MI_UNROLL
    for (int j = 0; j < 8; ++j) {
      pair_phase(Xa[j], Xb[j], cmul(Wl, w32(j)), pg[j * T], V[j], zkm[j]);
    }
    MI_UNROLL
    for (int j = 0; j < 8; ++j) {
      V[15 - j] = xchg32(zkm[j]);
    }
    if (tid == 0) {
      MI_UNROLL
      for (int j = 1; j < 8; ++j) {
        V[16 - j] = zkm[j];  // the mirror bins are lane 0's own (zkm[0] pairs DC with Nyquist: no bin of its own)
      }
      cf unused;
      pair_phase(Xa[8], Xb[8], w32(8), g0[0], V[8], unused);
    } else if (tid == 32) {
      MI_UNROLL
      for (int j = 0; j < 8; ++j) {
        V[15 - j] = zkm[j];
      }
    }
  }

  // ================= split form: block transform length 2K (K = this LDS length) ===========
  // The 2K-point transform Z of the block's complex words z[n] is done as two K-point
  // transforms through the same LDS buffer, E of the even words and O of the odd words,
  // joined by one radix-2 stage in registers (w = W_2K^k):
  //     Z[k] = E[k] + w O[k],   Z[k+K] = E[k] - w O[k]                       (forward, DIT)
  //     A'[k] = Z'[k] + Z'[k+K],  B'[k] = (Z'[k] - Z'[k+K]) conj(w)          (inverse, DIF)
  //     z'[2n] = IFFT_K(A')[n],   z'[2n+1] = IFFT_K(B')[n]
  // The K-point mirror pair (k, K-k) a thread owns closes under all of it: it carries the
  // two 2K-point mirror pairs (k, 2K-k) and (K-k, K+k), whose untangle twiddles are
  // W1 = W_4K^k and W2 = W_4K^(K-k) = -j conj(W1), and w = W1^2, W_2K^(K-k) = -conj(w).
  // X1*/X2* hold the untangled spectrum of pair 1 / pair 2 (16 quads per thread).
  //
  // Thread 0 owns the self-mirrored sets S_0 and S_{J/2}. Their 17 mirror pairs are quads of
  // exactly the same form (k = 0: pairs (0, 2K) and (K, K); k = K/2: both pairs coincide), so
  // instead of a second code path that would make wave 0 take twice as long as every other
  // wave at each spectral stage, lanes 0..16 of wave 0 take one of them each as a 17th quad:
  // thread 0 publishes its 64 transform outputs through 64 extra LDS words (xch), lane l
  // picks its four, and the results go straight into thread 0's two LDS blocks as the inputs
  // of its first inverse pass. Lane l <= 8: k = l*J (set S_0); lane l >= 9: k = J/2 + (l-9)*J.
  struct SelfLane {
    int srcK, srcM;  // xch words of E[k], E[K-k] (O at +32)
    int posK, posM;  // LDS words of the first-pass inputs A'[k], A'[K-k] in thread 0's blocks
  };
  static MI_DEVICE SelfLane self_lane(int tid, int blkB0) {
    SelfLane sl;
    if (tid <= 8) {
      const Bfly<16, 1> b0(Cfg::block_a(0));
      sl.srcK = tid;
      sl.srcM = (16 - tid) & 15;
      sl.posK = b0.at(tid);
      sl.posM = b0.at((16 - tid) & 15);
    } else {
      const int t = (tid - 9) & 7;
      const Bfly<16, 1> b1(blkB0);
      sl.srcK = 16 + t;
      sl.srcM = 16 + 15 - t;
      sl.posK = b1.at(t);
      sl.posM = b1.at(15 - t);
    }
    return sl;
  }
  static constexpr int kSelfLanes = 17;
  static constexpr int kXchWords = 64;
  static MI_DEVICE void quad_split(cf Ek, cf Em, cf Ok, cf Om, cf W1, cf &x1a, cf &x1b, cf &x2a, cf &x2b) {
    const cf w = cmul(W1, W1);
    const cf wo = cmul(w, Ok), co = cmulc(Om, w);  // w O[k], conj(w) O[K-k]
    const cf Zk = cadd(Ek, wo), ZkK = csub(Ek, wo);  // Z[k], Z[k+K]
    const cf Zm = csub(Em, co), ZmK = cadd(Em, co);  // Z[K-k], Z[2K-k]
    pair_split(Zk, ZmK, W1, x1a, x1b);
    pair_split(Zm, ZkK, cneg(cmulj(cconj(W1))), x2a, x2b);
  }
  // H = 0: A'[k], A'[K-k]; H = 1: B'[k], B'[K-k]
  template <int H>
  static MI_DEVICE void quad_phase(cf x1a, cf x1b, cf x2a, cf x2b, cf W1, f4 g1, f4 g2, cf &ok, cf &om) {
    cf zk1, zkm1, zk2, zkm2;
    pair_phase(x1a, x1b, W1, g1, zk1, zkm1);                        // Z'[k], Z'[2K-k]
    pair_phase(x2a, x2b, cneg(cmulj(cconj(W1))), g2, zk2, zkm2);    // Z'[K-k], Z'[K+k]
    if constexpr (H == 0) {
      ok = cadd(zk1, zkm2);
      om = cadd(zk2, zkm1);
    } else {
      const cf w = cmul(W1, W1);
      ok = cmulc(csub(zk1, zkm2), w);
      om = cneg(cmul(csub(zk2, zkm1), w));
    }
  }
  // both halves at once: the pair products are the same, only the last combination differs
  static MI_DEVICE void quad_phase_both(cf x1a, cf x1b, cf x2a, cf x2b, cf W1, f4 g1, f4 g2, cf &ok0, cf &om0, cf &ok1,
                                        cf &om1) {
    cf zk1, zkm1, zk2, zkm2;
    pair_phase(x1a, x1b, W1, g1, zk1, zkm1);                        // Z'[k], Z'[2K-k]
    pair_phase(x2a, x2b, cneg(cmulj(cconj(W1))), g2, zk2, zkm2);    // Z'[K-k], Z'[K+k]
    ok0 = cadd(zk1, zkm2);
    om0 = cadd(zk2, zkm1);
    const cf w = cmul(W1, W1);
    ok1 = cmulc(csub(zk1, zkm2), w);
    om1 = cneg(cmul(csub(zk2, zkm1), w));
  }
  static MI_DEVICE void split_spectrum2(const cf *EA, const cf *EB, const cf *OA, const cf *OB, cf Wa, cf *X1a, cf *X1b,
                                        cf *X2a, cf *X2b) {
    MI_UNROLL
    for (int t = 0; t < 16; ++t) {
      quad_split(EA[out_pos<16>(t)], EB[out_pos<16>(15 - t)], OA[out_pos<16>(t)], OB[out_pos<16>(15 - t)],
                 cmul(Wa, w64(t)), X1a[t], X1b[t], X2a[t], X2b[t]);
      if ((t & 3) == 3) {
        MI_SCHED_FENCE();  // inputs die as outputs are born: keeps the stage near 128 + 32 registers
      }
    }
  }
  // gt: this phase's [2][16][T] table, g0: this phase's [2][17] table of the self lanes.
  // The inputs of the first inverse pass go, in natural order, straight to the LDS words of
  // the thread's own two blocks (bA, bB): the blocks are dead at this point, only this thread
  // touches them before the next barrier, and the spectrum already holds 128 registers (the
  // 32 results would not fit beside it). Thread 0 has no quads of its own (its table rows
  // are zero and its stores are masked); both its blocks are filled by the self lanes.
  template <int H>
  static MI_DEVICE void phase_inputs2(int tid, const cf *X1a, const cf *X1b, const cf *X2a, const cf *X2b, const cf *Xs,
                                      cf Wa, cf Ws, const SelfLane &sl, const f4 *MI_RESTRICT gt,
                                      const f4 *MI_RESTRICT g0, cf *lds, const Bfly<16, 1> &bA, const Bfly<16, 1> &bB) {
    // Table words in eight groups of two slots (4 words), group g+1 requested before group g is used and each group's
    // loads fenced ahead of its arithmetic (8 words = 32 registers in flight beside the 128 of the spectrum): see
    // phase_inputs -- left alone the scheduler serialises load, wait, use (19.9k cycles per call instead of ~8k).
    const f4 *pg = gt + tid;
#if !defined(MIUPS_PHASE2_SLOTS)
#define MIUPS_PHASE2_SLOTS 1  // experiment switch (profiles/): slots per table group of the split form's spectral stage
#endif
#if !defined(MIUPS_PHASE2_DEPTH)
#define MIUPS_PHASE2_DEPTH 3  // ... and groups in flight (1x3: 117.9, 2x2: 116.3, 1x2: 110.6 Gsamples/s at config 4)
#endif
    constexpr int GS = MIUPS_PHASE2_SLOTS, GD = MIUPS_PHASE2_DEPTH, NG = 16 / GS;
    f4 gv[GD][2 * GS];
    auto request = [&](int q) {
      MI_UNROLL
      for (int j = 0; j < GS; ++j) {
        gv[q % GD][2 * j] = pg[(GS * q + j) * T];
        gv[q % GD][2 * j + 1] = pg[(16 + GS * q + j) * T];
      }
    };
    MI_UNROLL
    for (int q = 0; q < GD - 1; ++q) {
      request(q);
    }
    MI_UNROLL
    for (int q = 0; q < NG; ++q) {
      if (q + GD - 1 < NG) {
        request(q + GD - 1);
      }
      MI_SCHED_FENCE();
      MI_UNROLL
      for (int j = 0; j < GS; ++j) {
        const int t = GS * q + j;
        cf ok, om;
#if defined(MIUPS_EXP_NO_G)  // experiment switch (profiles/): spectral stage without its table words (wrong results)
        quad_phase<H>(X1a[t], X1b[t], X2a[t], X2b[t], cmul(Wa, w64(t)), f4{1.0f, 0.0f, 1.0f, 0.0f},
                      f4{0.5f, 0.0f, 0.5f, 0.0f}, ok, om);
#else
        quad_phase<H>(X1a[t], X1b[t], X2a[t], X2b[t], cmul(Wa, w64(t)), gv[q % GD][2 * j], gv[q % GD][2 * j + 1], ok, om);
#endif
        if (tid != 0) {
          lds[bA.at(t)] = ok;
          lds[bB.at(15 - t)] = om;
        }
      }
      MI_SCHED_FENCE();
    }
    if (tid < kSelfLanes) {
      cf ok, om;
      quad_phase<H>(Xs[0], Xs[1], Xs[2], Xs[3], Ws, g0[tid], g0[kSelfLanes + tid], ok, om);
      lds[sl.posK] = ok;
      lds[sl.posM] = om;  // lanes 0 and 8: the same word, the same value
    }
  }

  // The same stage for BOTH half transforms of a phase in one go (IoDesc::park != null): the pair products of the two
  // halves are identical, only the last combination differs, so the second half's first-pass inputs are formed here
  // too and parked in global memory ([slot][T] words of 16 bytes, lane-contiguous; 128 KB per workgroup, rewritten every
  // phase, so it lives in L2 / Infinity Cache) until the first half's inverse transform has left the LDS. Round 2 recomputed
  // the whole stage for the second half (table loads and ~1000 VALU instructions per thread again): the stage was 32 % of
  // the split kernel (profiles/r03_d_stamps_config4_split.txt).
  static MI_DEVICE void phase_inputs2_both(int tid, const cf *X1a, const cf *X1b, const cf *X2a, const cf *X2b, const cf *Xs,
                                           cf Wa, cf Ws, const SelfLane &sl, const f4 *MI_RESTRICT gt,
                                           const f4 *MI_RESTRICT g0, cf *lds, const Bfly<16, 1> &bA, const Bfly<16, 1> &bB,
                                           f4 *MI_RESTRICT park) {
    const f4 *pg = gt + tid;
    f4 *pk = park + tid;
    constexpr int GS = MIUPS_PHASE2_SLOTS, GD = MIUPS_PHASE2_DEPTH, NG = 16 / GS;
    f4 gv[GD][2 * GS];
    auto request = [&](int q) {
      MI_UNROLL
      for (int j = 0; j < GS; ++j) {
        gv[q % GD][2 * j] = pg[(GS * q + j) * T];
        gv[q % GD][2 * j + 1] = pg[(16 + GS * q + j) * T];
      }
    };
    MI_UNROLL
    for (int q = 0; q < GD - 1; ++q) {
      request(q);
    }
    MI_UNROLL
    for (int q = 0; q < NG; ++q) {
      if (q + GD - 1 < NG) {
        request(q + GD - 1);
      }
      MI_SCHED_FENCE();
      MI_UNROLL
      for (int j = 0; j < GS; ++j) {
        const int t = GS * q + j;
        cf ok0, om0, ok1, om1;
        quad_phase_both(X1a[t], X1b[t], X2a[t], X2b[t], cmul(Wa, w64(t)), gv[q % GD][2 * j], gv[q % GD][2 * j + 1], ok0, om0,
                        ok1, om1);
        if (tid != 0) {
          lds[bA.at(t)] = ok0;
          lds[bB.at(15 - t)] = om0;
        }
        pk[t * T] = f4{ok1.x, ok1.y, om1.x, om1.y};
      }
      MI_SCHED_FENCE();
    }
    if (tid < kSelfLanes) {
      cf ok0, om0, ok1, om1;
      quad_phase_both(Xs[0], Xs[1], Xs[2], Xs[3], Ws, g0[tid], g0[kSelfLanes + tid], ok0, om0, ok1, om1);
      lds[sl.posK] = ok0;
      lds[sl.posM] = om0;  // lanes 0 and 8: the same word, the same value
      park[16 * T + tid] = f4{ok1.x, ok1.y, om1.x, om1.y};
    }
  }
  // ... and the second half's stage: its parked first-pass inputs back into the thread's own two LDS blocks
  static MI_DEVICE void phase_inputs2_unpark(int tid, const SelfLane &sl, cf *lds, const Bfly<16, 1> &bA,
                                             const Bfly<16, 1> &bB, const f4 *MI_RESTRICT park) {
    const f4 *pk = park + tid;
    f4 v[16];
    MI_UNROLL
    for (int t = 0; t < 16; ++t) {
      v[t] = pk[t * T];
    }
    MI_SCHED_FENCE();
    if (tid != 0) {
      MI_UNROLL
      for (int t = 0; t < 16; ++t) {
        lds[bA.at(t)] = mk(v[t].x, v[t].y);
        lds[bB.at(15 - t)] = mk(v[t].z, v[t].w);
      }
    }
    if (tid < kSelfLanes) {
      const f4 s = park[16 * T + tid];
      lds[sl.posK] = mk(s.x, s.y);
      lds[sl.posM] = mk(s.z, s.w);
    }
  }

  // ================= epilogue: staging planes -> interleaved PCM frames ======
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
    constexpr int kDepth = (VPT >= 16 ? 4 : 8) / (W == 1 ? 2 : 1);  // the narrow form has half the registers, twice the threads
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
#if defined(MIUPS_EXP_NT_SCRATCH) && !defined(MIUPS_HOST_EMU)  // experiment switch (profiles/)
            v[d][e] = __builtin_nontemporal_load(scr + (cc * g.P + q * pg + pp) * g.Bp + i);
#else
            v[d][e] = scr[(cc * g.P + q * pg + pp) * g.Bp + i];
#endif
          }
        }
      }
      MI_UNROLL
      for (int d = 0; d < kDepth; ++d) {
        const int unit = base + d * T;
        if (unit < units) {
          // value e = (phase pp, channel cc) of frame (unit*pg + pp): the group's cg
          // channels of one frame are contiguous, frames are io.channels samples apart
          // (when cg == io.channels everything is one contiguous run)
          const long long frame0 = static_cast<long long>(unit) * pg;
          MI_UNROLL
          for (int e = 0; e < VPT; e += 4) {
            const int pp = e / cg, cc = e - pp * cg;
            char *dst = out_blk + ((frame0 + pp) * io.channels + cc) * 4 - 4 * e;
            if constexpr (FMT == kF32) {
              *reinterpret_cast<f4 *>(dst + 4 * e) = f4{v[d][e], v[d][e + 1], v[d][e + 2], v[d][e + 3]};
            } else {
              struct alignas(16) I4 {
                int32_t a, b, c, d;
              };
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
  // The same for the usual case pg == P (a unit = one i = all R = cg*P values of P consecutive frames = R consecutive
  // output samples): every plane base is wave-uniform (scalar registers), the only per-lane address words are the 32-bit
  // byte offsets 4*i (loads, shared by all R planes) and 4*R*i (stores) -- the epilogue's address arithmetic was most
  // of its instruction stream (profiles/r02_d_*: it ran 1.6x faster with twice the waves, i.e. issue-bound).
  template <int FMT, int R>
  static MI_DEVICE void epilogue_rows(const Geometry &g, const IoDesc &io, char *out_blk, const float *scr, int tid) {
    const unsigned Bc = static_cast<unsigned>(g.Bc), cg = static_cast<unsigned>(io.cg), P = static_cast<unsigned>(g.P);
    constexpr int kDepth = MIUPS_ROWS_DEPTH * (R >= 16 ? 4 : 8) / (W == 1 ? 2 : 1);
    const float *pl[R];  // value e of a unit = (phase e / cg, channel e % cg)
    MI_UNROLL
    for (int e = 0; e < R; ++e) {
      const unsigned pp = static_cast<unsigned>(e) / cg, cc = static_cast<unsigned>(e) - pp * cg;
      pl[e] = scr + static_cast<size_t>(cc * P + pp) * static_cast<unsigned>(g.Bp);
    }
    for (unsigned base = static_cast<unsigned>(tid); base < Bc; base += T * kDepth) {
      float v[kDepth][R];
      MI_UNROLL
      for (int d = 0; d < kDepth; ++d) {
        const unsigned i = base + d * T;
        if (i < Bc) {
          MI_UNROLL
          for (int e = 0; e < R; ++e) {
#if defined(MIUPS_EXP_EPI_NO_LOAD)  // timing experiment (profiles/r03_c_*): frames without reading the planes (WRONG results)
            v[d][e] = static_cast<float>(i + e) * 1.0e-9f;
#else
            v[d][e] = pl[e][i];
#endif
          }
        }
      }
      MI_UNROLL
      for (int d = 0; d < kDepth; ++d) {
        const unsigned i = base + d * T;
        if (i < Bc) {
          char *dst = out_blk + static_cast<size_t>(i * static_cast<unsigned>(4 * R));
#if defined(MIUPS_EXP_EPI_NO_STORE) && !defined(MIUPS_HOST_EMU)  // timing experiment (profiles/r03_c_*): planes read, no frame written
          MI_UNROLL
          for (int e = 0; e < R; ++e) {
            asm volatile("" ::"v"(v[d][e]));
          }
          continue;
#endif
          MI_UNROLL
          for (int e = 0; e < R; e += 4) {
            if constexpr (FMT == kF32) {
              *reinterpret_cast<f4 *>(dst + 4 * e) = f4{v[d][e], v[d][e + 1], v[d][e + 2], v[d][e + 3]};
            } else {
              struct alignas(16) I4 {
                int32_t a, b, c, d;
              };
              I4 w;
              w.a = static_cast<int32_t>(pcm_clamp(v[d][e], 0.9999999f) * 2147483648.0f);
              w.b = static_cast<int32_t>(pcm_clamp(v[d][e + 1], 0.9999999f) * 2147483648.0f);
              w.c = static_cast<int32_t>(pcm_clamp(v[d][e + 2], 0.9999999f) * 2147483648.0f);
              w.d = static_cast<int32_t>(pcm_clamp(v[d][e + 3], 0.9999999f) * 2147483648.0f);
#if defined(MIUPS_EXP_NT_FRAMES) && !defined(MIUPS_HOST_EMU)  // experiment switch (profiles/r03_c_*): streaming frame stores
              typedef int i4v __attribute__((ext_vector_type(4)));
              const i4v wv = {w.a, w.b, w.c, w.d};
              __builtin_nontemporal_store(wv, reinterpret_cast<i4v *>(dst + 4 * e));
#else
              *reinterpret_cast<I4 *>(dst + 4 * e) = w;
#endif
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
      pcm_store(out_blk, FMT, static_cast<long long>(m) * io.channels + cc, scr[(cc * g.P + p) * g.Bp + i]);
    }
  }
  // Tiled form for many planes (R = cg*P > 16 rows): a [R][64] tile of the staging planes
  // goes through LDS (free during the epilogue; rows padded to 65 words), read with
  // lane-contiguous 256-byte row segments and written as 16-byte vectors in output order
  // (for one i: p-major, channel-minor = exactly the frame layout). The loads of tile k+1
  // are in flight while tile k is stored.
  template <int FMT, int EPT>
  static MI_DEVICE void epilogue_tiled(const Geometry &g, const IoDesc &io, char *out_blk, const float *scr, float *tile,
                                       int tid) {
    constexpr int TI = 64, LD = TI + 1;
    const int P = g.P;
    const int lcg = __builtin_ctz(io.cg), lR = lcg + __builtin_ctz(P);  // both powers of two here
    const int ntiles = (g.Bc + TI - 1) / TI;
    float v[EPT];
    auto fetch = [&](int k) {
      MI_UNROLL
      for (int j = 0; j < EPT; ++j) {
        const int x = tid + j * T, row = x / TI, col = x - row * TI, i = k * TI + col;
        v[j] = (i < g.Bc) ? scr[static_cast<long long>(row) * g.Bp + i] : 0.0f;
      }
    };
    fetch(0);
    for (int k = 0; k < ntiles; ++k) {
      MI_UNROLL
      for (int j = 0; j < EPT; ++j) {
        const int x = tid + j * T, row = x / TI, col = x - row * TI;
        tile[row * LD + col] = v[j];
      }
      MI_SYNC();
      if (k + 1 < ntiles) {
        fetch(k + 1);
      }
      MI_UNROLL
      for (int j = 0; j < EPT / 4; ++j) {
        const int e4 = 4 * (tid + j * T);  // first of 4 consecutive output values of this tile
        const int il = e4 >> lR, r0 = e4 - (il << lR), i = k * TI + il;
        if (i < g.Bc) {
          float w[4];
          MI_UNROLL
          for (int e = 0; e < 4; ++e) {
            const int r = r0 + e, p = r >> lcg, cc = r - (p << lcg);
            w[e] = tile[(cc * P + p) * LD + il];
          }
          const int p0 = r0 >> lcg, cc0 = r0 - (p0 << lcg);
          char *dst = out_blk + ((static_cast<long long>(i) * P + p0) * io.channels + cc0) * 4;
          if constexpr (FMT == kF32) {
            *reinterpret_cast<f4 *>(dst) = f4{w[0], w[1], w[2], w[3]};
          } else {
            struct alignas(16) I4 {
              int32_t a, b, c, d;
            };
            I4 o;
            o.a = static_cast<int32_t>(pcm_clamp(w[0], 0.9999999f) * 2147483648.0f);
            o.b = static_cast<int32_t>(pcm_clamp(w[1], 0.9999999f) * 2147483648.0f);
            o.c = static_cast<int32_t>(pcm_clamp(w[2], 0.9999999f) * 2147483648.0f);
            o.d = static_cast<int32_t>(pcm_clamp(w[3], 0.9999999f) * 2147483648.0f);
            *reinterpret_cast<I4 *>(dst) = o;
          }
        }
      }
      MI_SYNC();
    }
  }
  // Register-transposed form for many planes when Bc % 4 == 0: a unit is 4 consecutive
  // i of the 4 planes that make one 16-byte run of a frame (4 x 16-byte loads, a 4x4
  // transpose in registers, 4 x 16-byte stores). Lanes run over the R/4 runs of a frame
  // first, so one store instruction writes whole frames (>= 128 contiguous bytes) and one
  // load instruction reads 64-byte pieces of R/4 planes; kUnits units per thread keep
  // kUnits*64 bytes per lane in flight (the planes come from L2/Infinity Cache/HBM with
  // microseconds of latency: the epilogue is latency-bound unless this much is in flight).
  template <int FMT>
  static MI_DEVICE void epilogue_quad(const Geometry &g, const IoDesc &io, char *out_blk, const float *scr, int tid) {
    constexpr int kUnits = W == 1 ? 4 : 8;  // (the narrow form has half the registers, twice the threads)
    const int P = g.P, cg = io.cg;
    const int lcg = __builtin_ctz(cg), lRq = lcg + __builtin_ctz(P) - 2;  // log2(R/4)
    const int units = (g.Bc >> 2) << lRq;
    for (int base = tid; base < units; base += T * kUnits) {
      f4 v[kUnits][4];
      MI_UNROLL
      for (int d = 0; d < kUnits; ++d) {
        const int u = base + d * T;
        if (u < units) {
          const int iq = u >> lRq, r0 = (u - (iq << lRq)) << 2;
          MI_UNROLL
          for (int e = 0; e < 4; ++e) {
            const int r = r0 + e, p = r >> lcg, cc = r - (p << lcg);
            v[d][e] = *reinterpret_cast<const f4 *>(scr + static_cast<long long>(cc * P + p) * g.Bp + 4 * iq);
          }
        }
      }
      MI_UNROLL
      for (int d = 0; d < kUnits; ++d) {
        const int u = base + d * T;
        if (u < units) {
          const int iq = u >> lRq, r0 = (u - (iq << lRq)) << 2;
          const int p0 = r0 >> lcg, cc0 = r0 - (p0 << lcg);
          char *dst = out_blk + ((static_cast<long long>(4 * iq) * P + p0) * io.channels + cc0) * 4;
          const long long frame_step = static_cast<long long>(P) * io.channels * 4;  // bytes from i to i+1
          const float m[4][4] = {{v[d][0].x, v[d][1].x, v[d][2].x, v[d][3].x},
                                 {v[d][0].y, v[d][1].y, v[d][2].y, v[d][3].y},
                                 {v[d][0].z, v[d][1].z, v[d][2].z, v[d][3].z},
                                 {v[d][0].w, v[d][1].w, v[d][2].w, v[d][3].w}};
          MI_UNROLL
          for (int e = 0; e < 4; ++e) {
            if constexpr (FMT == kF32) {
              *reinterpret_cast<f4 *>(dst + e * frame_step) = f4{m[e][0], m[e][1], m[e][2], m[e][3]};
            } else {
              struct alignas(16) I4 {
                int32_t a, b, c, d;
              };
              I4 o;
              o.a = static_cast<int32_t>(pcm_clamp(m[e][0], 0.9999999f) * 2147483648.0f);
              o.b = static_cast<int32_t>(pcm_clamp(m[e][1], 0.9999999f) * 2147483648.0f);
              o.c = static_cast<int32_t>(pcm_clamp(m[e][2], 0.9999999f) * 2147483648.0f);
              o.d = static_cast<int32_t>(pcm_clamp(m[e][3], 0.9999999f) * 2147483648.0f);
              *reinterpret_cast<I4 *>(dst + e * frame_step) = o;
            }
          }
        }
      }
    }
  }

  template <int FMT>
  static MI_DEVICE bool epilogue_tiled_dispatch(const Geometry &g, const IoDesc &io, char *out_blk, const float *scr,
                                                float *tile, int tid) {
    const int ept = io.cg * g.P * 64 / T;  // tile values per thread
    switch (ept) {
      case 4: epilogue_tiled<FMT, 4>(g, io, out_blk, scr, tile, tid); return true;
      case 8: epilogue_tiled<FMT, 8>(g, io, out_blk, scr, tile, tid); return true;
      case 16: epilogue_tiled<FMT, 16>(g, io, out_blk, scr, tile, tid); return true;
      case 32: epilogue_tiled<FMT, 32>(g, io, out_blk, scr, tile, tid); return true;
      default: return false;
    }
  }

  static MI_DEVICE void epilogue(const Geometry &g, const IoDesc &io, int s, int c0, int blk, const float *scr, cf *lds,
                                 int tid) {
    const int ob = pcm_bytes(io.out_fmt);
    char *out_blk = static_cast<char *>(io.out) + s * io.out_stream_stride +
                    (static_cast<long long>(blk) * g.B * io.channels + c0) * ob;
    const int cg = io.cg;
    const bool pow2 = (cg & (cg - 1)) == 0 && (g.P & (g.P - 1)) == 0;
    {
      // many planes: LDS-tiled transpose (needs 16-byte runs, a 4-byte format, the tile in LDS)
      const int R = cg * g.P;
      const bool runs4 = (cg % 4 == 0 && io.channels % 4 == 0) || (cg == io.channels && R % 4 == 0);
      const bool fits = static_cast<long long>(R) * 65 * 4 <= static_cast<long long>(K) * 8 && (R * 64) % T == 0;
      const bool wide = pow2 && R > 16 && runs4 && io.out_vec_ok && (io.out_fmt == kF32 || io.out_fmt == kS32);
#if defined(MIUPS_EXP_QUAD8)  // experiment switch (profiles/): register-transposed form from 8 planes up
      const bool quad = (wide || (pow2 && R >= 8 && runs4 && io.out_vec_ok && (io.out_fmt == kF32 || io.out_fmt == kS32)));
#else
      const bool quad = wide;
#endif
      if (quad && g.Bc % 4 == 0 && reinterpret_cast<uintptr_t>(scr) % 16 == 0) {
        if (io.out_fmt == kF32) {
          epilogue_quad<kF32>(g, io, out_blk, scr, tid);
        } else {
          epilogue_quad<kS32>(g, io, out_blk, scr, tid);
        }
        return;
      }
      if (wide && fits) {
        float *tile = reinterpret_cast<float *>(lds);
        const bool done = io.out_fmt == kF32 ? epilogue_tiled_dispatch<kF32>(g, io, out_blk, scr, tile, tid)
                                             : epilogue_tiled_dispatch<kS32>(g, io, out_blk, scr, tile, tid);
        if (done) {
          return;
        }
      }
    }
    const int pg = pow2 ? (cg * g.P <= 16 ? g.P : (16 / cg > 0 ? 16 / cg : 1)) : 1;
    const int vpt = cg * pg;
    // 16-byte stores need every 4-value run contiguous and aligned: the group is the whole
    // frame, or groups and frames are both multiples of 4 channels
    const bool runs_ok = cg == io.channels || (cg % 4 == 0 && io.channels % 4 == 0);
    const bool vec = pow2 && io.out_vec_ok && runs_ok && cg <= 16 && vpt >= 4 &&
                     (io.out_fmt == kF32 || io.out_fmt == kS32);
    if (vec && pg == g.P && cg == io.channels && static_cast<long long>(g.B) * cg * 4 < (1ll << 32)) {
      if (io.out_fmt == kF32) {
        if (vpt == 4) epilogue_rows<kF32, 4>(g, io, out_blk, scr, tid);
        else if (vpt == 8) epilogue_rows<kF32, 8>(g, io, out_blk, scr, tid);
        else epilogue_rows<kF32, 16>(g, io, out_blk, scr, tid);
      } else {
        if (vpt == 4) epilogue_rows<kS32, 4>(g, io, out_blk, scr, tid);
        else if (vpt == 8) epilogue_rows<kS32, 8>(g, io, out_blk, scr, tid);
        else epilogue_rows<kS32, 16>(g, io, out_blk, scr, tid);
      }
      return;
    }
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

  // One channel-block: forward FFT, split, then per phase multiply + inverse FFT
  // into this channel's staging planes (scr_c = [P][Bc] floats).
  // EXT: the frames are written by interleave_*_kernel (groups narrower than a frame); the
  // staging planes are then not read back by this kernel and are stored with streaming stores.
  template <bool EXT, bool PARTS = false>
  static MI_DEVICE void channel_block(const Geometry &g, const IoDesc &io, const BlockIo &b, float *scr_c,
                                      const FusedTables &ft, cf *lds, int tid, int cc, int pLo = 0, int pHi = 0) {
    const int sb = 64 * (cc & 1);  // stamp slot base (diagnostic builds)
    (void)sb;
    MI_STAMP(sb + 0);
    // ------------------------------ forward ------------------------------
    switch (io.in_fmt) {
      case kS32: fwd_first_fmt<kS32>(b, lds, ft.tw, tid); break;
      case kF32: fwd_first_fmt<kF32>(b, lds, ft.tw, tid); break;
      case kS16: fwd_first_fmt<kS16>(b, lds, ft.tw, tid); break;
      default: fwd_first_fmt<kS24_3LE>(b, lds, ft.tw, tid); break;
    }
    MI_STAMP(sb + 1);
    MI_SYNC_PAIR();
    MI_STAMP(sb + 2);
    // radix-16 passes with strides 256 and 16 exist when K/R0 >= 4096 resp. >= 256
    // (when R0 == 1 the first pass already was the stride-K/16 radix-16 pass); R32 plan: one radix-32 pass
    constexpr int kFirstMidStride = R32 ? 0 : ((R0 > 1) ? S0 / 16 : S0 / 256);
    if constexpr (R32) {
      fwd_mid32(lds, ft.tw, tid);
      MI_STAMP(sb + 5);
      MI_SYNC_PAIR();
      MI_STAMP(sb + 6);
    }
    if constexpr (kFirstMidStride >= 256) {
      fwd_mid<256>(lds, ft.tw, tid);
      MI_STAMP(sb + 3);
      MI_SYNC();
      MI_STAMP(sb + 4);
    }
    if constexpr (kFirstMidStride >= 16) {
      fwd_mid<16>(lds, ft.tw, tid);
      MI_STAMP(sb + 5);
      MI_SYNC_PAIR();
      MI_STAMP(sb + 6);
    }
    const bool evenOc = (b.Oc & 1) == 0;
    // Workgroups that share an XCD run in near lockstep and would all pull the same
    // spectrum lines out of the same L2 channels at the same moment (measured: 3.5x
    // slower phase-spectrum loads). Each starts its phase loop at a different phase.
    // PARTS (fused_parts_kernel): this workgroup computes phases [pLo, pHi) only; its siblings of the same unit take the
    // others, so the rotation must not depend on the workgroup
    const int rot = PARTS ? cc : (MI_BID_X >> 3) + cc;
    if constexpr (W == 1) {
      // ---- narrow form: one set per lane, mirror set in lane ^ 32 ----
      const int blk = ft.blockB[tid];  // this lane's LDS block in the two stride-1 passes
      cf V[16];
      lds_get<16, 1>(lds, Bfly<16, 1>(blk), V);
      dft16<-1>(V);
      MI_STAMP(sb + 7);
      cf Xa[9], Xb[9];
      const cf Wl = ft.WmT[tid];
      split_spectrum_n(tid, V, Wl, Xa, Xb);
      MI_STAMP(sb + 8);
      for (int pi = 0; pi < g.P; ++pi) {
        const int p = (pi + rot) % g.P;
        const f4 *gt = ft.GT + static_cast<long long>(p) * 8 * T;
        const f4 *g0 = ft.G0 + p;
        float *plane = scr_c + static_cast<long long>(p) * g.Bp;
        int tl = tid;  // per-phase copy of the thread index (see MI_OPAQUE_VGPR)
        MI_OPAQUE_VGPR(tl);
        phase_inputs_n(tl, Xa, Xb, Wl, gt, g0, V);
        const int sp = sb + 9 + 10 * (pi & 3);
        (void)sp;
        MI_STAMP(sp + 0);
        int bk = blk;
        MI_OPAQUE_VGPR(bk);
        dft_put<+1, 16, 1, false>(lds, Bfly<16, 1>(bk), V, nullptr);
        MI_STAMP(sp + 1);
        MI_SYNC_PAIR();
        MI_STAMP(sp + 2);
        if constexpr (kFirstMidStride >= 16) {
          MI_OPAQUE_VGPR(tl);
          inv_mid<16>(lds, ft.tw, tl);
          MI_STAMP(sp + 3);
          MI_SYNC();
          MI_STAMP(sp + 4);
        }
        if constexpr (kFirstMidStride >= 256) {
          MI_OPAQUE_VGPR(tl);
          inv_mid<256>(lds, ft.tw, tl);
          MI_STAMP(sp + 5);
          MI_SYNC_PAIR();
          MI_STAMP(sp + 6);
        }
        MI_OPAQUE_VGPR(tl);
        if (evenOc) {
          inv_last<true, EXT>(plane, b.Oc, g.Bc, lds, ft.tw, tl);
        } else {
          inv_last<false>(plane, b.Oc, g.Bc, lds, ft.tw, tl);
        }
        MI_STAMP(sp + 7);
        MI_SYNC();  // every read of this phase done before the next phase's first pass writes
        MI_STAMP(sp + 8);
      }
    } else {
      channel_block_wide<EXT, PARTS>(g, b, scr_c, ft, lds, tid, sb, evenOc, rot, pLo, pHi);
    }
  }

  // the wide form's second half: last forward pass in registers, split, phase loop
  template <bool EXT, bool PARTS = false>
  static MI_DEVICE void channel_block_wide(const Geometry &g, const BlockIo &b, float *scr_c, const FusedTables &ft, cf *lds,
                                           int tid, int sb, bool evenOc, int rot, int pLo = 0, int pHi = 0) {
    (void)sb;
    constexpr int kFirstMidStride = R32 ? 0 : ((R0 > 1) ? S0 / 16 : S0 / 256);
    const int blkA = Cfg::block_a(tid);
    const int blkB = ft.blockB[tid];
    cf A[16], B[16];
    fwd_last(lds, blkA, blkB, A, B);
    MI_STAMP(sb + 7);
    // no barrier: the next LDS access is this thread writing its own two blocks

    // ------------------------- split (once per block) --------------------
    cf Xa[17], Xb[17];
    const cf Wa = ft.WmT[tid];
    const cf Wb = ft.Wb;
    const cf Wa2 = tid == 0 ? ft.Wself : Wa;
    split_spectrum(tid, A, B, Wa, Wa2, Wb, Xa, Xb);
    MI_STAMP(sb + 8);

    // --------------------------- per output phase ------------------------
    const int piLo = PARTS ? pLo : 0, piHi = PARTS ? pHi : g.P;
    for (int pi = piLo; pi < piHi; ++pi) {
      const int p = (pi + rot) % g.P;
      const f4 *gt = ft.GT + static_cast<long long>(p) * 16 * T;
      const f4 *g0 = ft.G0 + p * 17;
      float *plane = scr_c + static_cast<long long>(p) * g.Bp;
      int tl = tid;  // per-phase copy of the thread index (see MI_OPAQUE_VGPR)
      MI_OPAQUE_VGPR(tl);
      phase_inputs(tl, Xa, Xb, Wa, Wa2, Wb, gt, g0, A, B);
      const int sp = sb + 9 + 10 * (pi & 3);
      (void)sp;
      MI_STAMP(sp + 0);
      int ba = blkA, bb = blkB;
      MI_OPAQUE_VGPR(ba);
      MI_OPAQUE_VGPR(bb);
      inv_first(lds, ba, bb, A, B);
      MI_STAMP(sp + 1);
      MI_SYNC_PAIR();
      MI_STAMP(sp + 2);
      if constexpr (R32) {
        MI_OPAQUE_VGPR(tl);
        inv_mid32(lds, ft.tw, tl);
        MI_STAMP(sp + 3);
        MI_SYNC();
        MI_STAMP(sp + 4);
      }
      if constexpr (kFirstMidStride >= 16) {
        MI_OPAQUE_VGPR(tl);
        inv_mid<16>(lds, ft.tw, tl);
        MI_STAMP(sp + 3);
        MI_SYNC();
        MI_STAMP(sp + 4);
      }
      if constexpr (kFirstMidStride >= 256) {
        MI_OPAQUE_VGPR(tl);
        inv_mid<256>(lds, ft.tw, tl);
        MI_STAMP(sp + 5);
        MI_SYNC_PAIR();
        MI_STAMP(sp + 6);
      }
      MI_OPAQUE_VGPR(tl);
      if (evenOc) {
#if defined(MIUPS_EXP_NT_PLANES_ALWAYS)  // experiment switch (profiles/): streaming plane stores with the in-kernel epilogue too
        inv_last<true, true>(plane, b.Oc, g.Bc, lds, ft.tw, tl);
#elif defined(MIUPS_EXP_NO_NT_PLANES)  // experiment switch (profiles/): cached plane stores for the interleave kernels too
        inv_last<true, false>(plane, b.Oc, g.Bc, lds, ft.tw, tl);
#elif defined(MIUPS_EXP_NO_PRUNE)  // experiment switch (profiles/r03_b_*): full last pass whatever the history length
        inv_last<true, EXT>(plane, b.Oc, g.Bc, lds, ft.tw, tl);
#else
        if constexpr (R32) {
          inv_last<true, EXT>(plane, b.Oc, g.Bc, lds, ft.tw, tl);
        } else if (R0 > 1 && b.Oc >= K) {  // workgroup-uniform: the lower half of every transform is discarded history
          // (radix-16 last passes, K = 4096 / 256, keep the full form: the pruned one measured 1.3 % slower at config 3,
          // profiles/r03_b_prune_trickle.txt)
          inv_last<true, EXT, (R0 > 1)>(plane, b.Oc, g.Bc, lds, ft.tw, tl);
        } else {
          inv_last<true, EXT>(plane, b.Oc, g.Bc, lds, ft.tw, tl);
        }
#endif
      } else {
        inv_last<false>(plane, b.Oc, g.Bc, lds, ft.tw, tl);
      }
      MI_STAMP(sp + 7);
      MI_SYNC();  // every read of this phase done before the next phase's first pass writes
      MI_STAMP(sp + 8);
    }
  }

  // Split form of channel_block (block transform length 2K): two forward transforms (even
  // and odd complex words), then per phase two inverse transforms whose results are the
  // even and odd complex words of y_p. Half h of phase p goes to plane + h*Bc/2 as
  // (y[4m + 2h], y[4m + 2h + 1]) pairs, m >= Oc/4 (the host only takes this path when
  // Oc % 4 == 0); the interleave kernels read that layout (IoDesc::split_planes).
  static MI_DEVICE void forward_half(const IoDesc &io, const BlockIo &b, const FusedTables &ft, cf *lds, int tid, cf *A,
                                     cf *B) {
    static_assert(!R32, "the split form runs the classic pass plan (its 128 registers of spectrum leave no room for a radix-32 butterfly)");
    switch (io.in_fmt) {
      case kS32: fwd_first_fmt<kS32, 1>(b, lds, ft.tw, tid); break;
      case kF32: fwd_first_fmt<kF32, 1>(b, lds, ft.tw, tid); break;
      case kS16: fwd_first_fmt<kS16, 1>(b, lds, ft.tw, tid); break;
      default: fwd_first_fmt<kS24_3LE, 1>(b, lds, ft.tw, tid); break;
    }
    MI_SYNC();
    constexpr int kFirstMidStride = (R0 > 1) ? S0 / 16 : S0 / 256;
    if constexpr (kFirstMidStride >= 256) {
      fwd_mid<256>(lds, ft.tw, tid);
      MI_SYNC();
    }
    if constexpr (kFirstMidStride >= 16) {
      fwd_mid<16>(lds, ft.tw, tid);
      MI_SYNC();
    }
    fwd_last(lds, Cfg::block_a(tid), ft.blockB[tid], A, B);
  }
  // One butterfly in flight instead of two: the split form keeps 64 (thread 0: 66) complex
  // values of untangled spectrum per thread across the inverse passes (twice the plain
  // kernel's), which leaves room for one radix-16 butterfly's registers, not two.
  // (Measured alternative: parking half of the spectrum in global memory between uses
  // made the spectral stage 10x slower -- the parked words fall out of L2.)
  template <int S>
  static MI_DEVICE void inv_mid_seq(cf *lds, const cf *tw, int tid) {
    constexpr int LOG2L = (S == 16) ? 8 : 12;
    MI_UNROLL
    for (int i = 0; i < 2; ++i) {
      const int q = tid + i * T;
      const cf w = load_tw<LOG2L>(tw, q & (S - 1));
      const Bfly<16, S> bf(q);
      cf X[16];
      lds_get<16, S>(lds, bf, X);
      apply_twiddles<+1, 16>(X, w);
      dft_put<+1, 16, S, false>(lds, bf, X, nullptr);
      MI_SCHED_FENCE();
    }
  }
  // PARTS: only half transforms [itLo, itHi) of the 2P (phase, half) pairs below (fused_split_parts_kernel)
  template <bool PARTS = false>
  static MI_DEVICE void channel_block_split(const Geometry &g, const IoDesc &io, BlockIo b, float *scr_c,
                                            const FusedTables &ft, cf *lds, int tid, int cc, f4 *park, int itLo = 0,
                                            int itHi = 0) {
    constexpr int kFirstMidStride = (R0 > 1) ? S0 / 16 : S0 / 256;
    const int sb = 64 * (cc & 1);  // stamp slot base (diagnostic builds)
    (void)sb;
    const int blkA = Cfg::block_a(tid);
    const int blkB = ft.blockB[tid];
    cf X1a[16], X1b[16], X2a[16], X2b[16], Xs[4];
    const cf Wa = ft.WmT[tid];
    const SelfLane sl = self_lane(tid, ft.blockB[0]);
    const cf Ws = tid < kSelfLanes ? ft.selfW[tid] : mk(1.0f, 0.0f);
    cf *xch = lds + K;  // kXchWords extra LDS words behind the transform buffer
    {
      cf EA[16], EB[16], OA[16], OB[16];
      b.noff = 0;
      MI_STAMP(sb + 0);
      forward_half(io, b, ft, lds, tid, EA, EB);
      if (tid == 0) {
        MI_UNROLL
        for (int u = 0; u < 16; ++u) {
          xch[u] = EA[out_pos<16>(u)];
          xch[16 + u] = EB[out_pos<16>(u)];
        }
      }
      MI_STAMP(sb + 1);
      MI_SYNC();  // every thread's last-pass reads done before the next transform's first pass writes
      MI_STAMP(sb + 2);
      b.noff = 2;
      int t2 = tid;
      MI_OPAQUE_VGPR(t2);
      forward_half(io, b, ft, lds, t2, OA, OB);
      if (tid == 0) {
        MI_UNROLL
        for (int u = 0; u < 16; ++u) {
          xch[32 + u] = OA[out_pos<16>(u)];
          xch[48 + u] = OB[out_pos<16>(u)];
        }
      }
      MI_STAMP(sb + 7);
      split_spectrum2(EA, EB, OA, OB, Wa, X1a, X1b, X2a, X2b);
    }
    MI_SYNC();  // thread 0's transform outputs are in xch
    Xs[0] = Xs[1] = Xs[2] = Xs[3] = mk(0.0f, 0.0f);
    if (tid < kSelfLanes) {
      quad_split(xch[sl.srcK], xch[sl.srcM], xch[32 + sl.srcK], xch[32 + sl.srcM], Ws, Xs[0], Xs[1], Xs[2], Xs[3]);
    }
    MI_STAMP(sb + 8);
    const int rot = PARTS ? cc : (MI_BID_X >> 3) + cc;
    const int itFirst = PARTS ? itLo : 0, itEnd = PARTS ? itHi : 2 * g.P;
    for (int it = itFirst; it < itEnd; ++it) {
      const int p = ((it >> 1) + rot) % g.P, h = it & 1;
      const f4 *gt = ft.GT + static_cast<long long>(p) * 32 * T;
      const f4 *g0 = ft.G0 + p * (2 * kSelfLanes);
      float *half = scr_c + static_cast<long long>(p) * g.Bp + h * (g.Bc >> 1);
      int tl = tid;
      MI_OPAQUE_VGPR(tl);
      int ba = blkA, bb = blkB;
      MI_OPAQUE_VGPR(ba);
      MI_OPAQUE_VGPR(bb);
      const Bfly<16, 1> bfA(ba), bfB(bb);
      if (park != nullptr) {
        if (h == 0) {
          phase_inputs2_both(tl, X1a, X1b, X2a, X2b, Xs, Wa, Ws, sl, gt, g0, lds, bfA, bfB, park);
        } else {
          phase_inputs2_unpark(tl, sl, lds, bfA, bfB, park);
        }
      } else if (h == 0) {
        phase_inputs2<0>(tl, X1a, X1b, X2a, X2b, Xs, Wa, Ws, sl, gt, g0, lds, bfA, bfB);
      } else {
        phase_inputs2<1>(tl, X1a, X1b, X2a, X2b, Xs, Wa, Ws, sl, gt, g0, lds, bfA, bfB);
      }
      const int sp = sb + 9 + 10 * (it & 3);
      (void)sp;
      MI_STAMP(sp + 0);
      MI_SYNC();  // the self lanes' results are in thread 0's blocks
      // first inverse pass (stride 1), one block at a time, in place
      {
        cf X[16];
        lds_get<16, 1>(lds, bfA, X);
        dft_put<+1, 16, 1, false>(lds, bfA, X, nullptr);
        MI_SCHED_FENCE();
        lds_get<16, 1>(lds, bfB, X);
        dft_put<+1, 16, 1, false>(lds, bfB, X, nullptr);
      }
      MI_STAMP(sp + 1);
      MI_SYNC();
      MI_STAMP(sp + 2);
      if constexpr (kFirstMidStride >= 16) {
        MI_OPAQUE_VGPR(tl);
        inv_mid_seq<16>(lds, ft.tw, tl);
        MI_STAMP(sp + 3);
        MI_SYNC();
        MI_STAMP(sp + 4);
      }
      if constexpr (kFirstMidStride >= 256) {
        MI_OPAQUE_VGPR(tl);
        inv_mid_seq<256>(lds, ft.tw, tl);
        MI_STAMP(sp + 5);
        MI_SYNC();
        MI_STAMP(sp + 6);
      }
      MI_OPAQUE_VGPR(tl);
#if defined(MIUPS_EXP_NO_PRUNE)
      inv_last<true, true>(half, g.Oc >> 1, g.Bc >> 1, lds, ft.tw, tl);
#else
      if (R0 > 1 && (g.Oc >> 1) >= K) {  // the half transform's discarded history covers its lower half: pruned last pass
        inv_last<true, true, (R0 > 1)>(half, g.Oc >> 1, g.Bc >> 1, lds, ft.tw, tl);
      } else {
        inv_last<true, true>(half, g.Oc >> 1, g.Bc >> 1, lds, ft.tw, tl);
      }
#endif
      MI_STAMP(sp + 7);
      MI_SYNC();
      MI_STAMP(sp + 8);
    }
  }

  // work item = (block, stream, channel group); it = (blk*streams + s)*groups + grp
  // PARTS: io.phase_parts consecutive workgroups share one work item and take P / phase_parts phases of it each (each
  // repeats the forward transform): a call too small to fill the chip finishes in about 1/phase_parts of the phase loop.
  // Wide form, external epilogue only.
  template <bool SPLIT, bool EXT, bool PARTS = false>
  static MI_DEVICE void run(const Geometry &g, const IoDesc &io, const FusedTables &ft, cf *lds) {
    const int tid = MI_TID_X;
    // Workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so
    // hardware ids b and b+8 share an L2. Map them to CONSECUTIVE work items: with
    // whole-frame groups those are consecutive blocks of one stream, which overlap in
    // (taps-1)/L input frames, and an XCD then finds that history (and its neighbours'
    // new input) in its own L2.
    // Placement only affects speed; any mapping is a bijection onto the items.
    const int nwg = MI_GDIM_X, hw = MI_BID_X;
    const int xq = nwg / 8, xr = nwg % 8, xk = hw % 8;
#if defined(MIUPS_EXP_NO_XCD_MAP)  // experiment switch (profiles/): identity mapping
    const int local = hw + 0 * (xq + xr + xk);
#else
    const int local = xk * xq + (xk < xr ? xk : xr) + hw / 8;
#endif
    int unit = local, pLo = 0, pHi = 0;
    if constexpr (PARTS) {
      // the split form's unit of division is the half transform: 2P of them per channel-block
      const int pieces = (SPLIT ? 2 * g.P : g.P) / io.phase_parts;
      unit = local / io.phase_parts;
      pLo = (local - unit * io.phase_parts) * pieces;
      pHi = pLo + pieces;
    }
    // item = (stream*groups + group) * blocks + block   (block fastest)
    const int item = io.item0 + unit;
    // item = (stream*blocks + block)*groups + group   (group fastest: a chunk of items
    // holds whole frames, which the external epilogue needs)
    const int sb = item / io.groups;
    const int c0 = (item - sb * io.groups) * io.cg;
    const int s = sb / io.blocks;
    const int blk = sb - s * io.blocks;
    float *scr = io.scratch + static_cast<long long>(unit) * io.cg * g.P * g.Bp;
    for (int cc = 0; cc < io.cg; ++cc) {
      const BlockIo b = make_block_io(g, io, s, c0 + cc, blk);
      int tc = tid;  // fresh copy per channel: keeps address arithmetic inside the loop body
      MI_OPAQUE_VGPR(tc);
      if constexpr (SPLIT) {
        f4 *park = (io.park && !PARTS) ? io.park + static_cast<long long>(local) * split_park_words(T) : nullptr;
        channel_block_split<PARTS>(g, io, b, scr + static_cast<long long>(cc) * g.P * g.Bp, ft, lds, tc, cc, park, pLo, pHi);
      } else {
        channel_block<EXT, PARTS>(g, io, b, scr + static_cast<long long>(cc) * g.P * g.Bp, ft, lds, tc, cc, pLo, pHi);
      }
    }
    MI_STAMP(128);
    // every plane store of this workgroup is complete and visible to it
    // (the loop ends in a workgroup barrier, which carries the release/acquire)
    if constexpr (!SPLIT && !EXT) {  // the split form always leaves the frames to interleave_*_kernel (its own epilogue
                                     // measured no faster than the separate pass: profiles/r02_d_*)
      epilogue(g, io, s, c0, blk, scr, lds, tid);
    }
    if constexpr (!SPLIT && EXT && !PARTS && W == 2) {
      // cooperative frames: this workgroup's planes are stored; say so, then assemble tiles of pairs that are complete
      // (device/frame_tile.h). Workgroup-uniform branch; nothing here waits for another workgroup.
      if (io.fsync != nullptr) {
        coop_frames<T>(g, io, unit, reinterpret_cast<float *>(lds), tid);
      }
    }
    MI_STAMP(129);
  }
};

// EXT = false: whole-frame groups, frames written by the kernel's own epilogue;
// EXT = true: io.ext_epilogue, the kernel stops at the staging planes (see channel_block).
// Launch bounds: the wide form at least two waves per SIMD, i.e. at most 256 registers per lane, for every size (with
// "1" hipcc parks 9-18 values of the K <= 8192 kernels in AGPRs instead of spilling them: 272 registers, ONE wave per
// SIMD, and a lone wave issues one VALU instruction per ~5 cycles instead of one per ~2.5:
// profiles/r02_a_ubench_valu_lds_rates.txt); the narrow form four waves per SIMD = at most 128 registers.
// R32: the experimental pass plan (FusedCfg); the host lays the tables out for the same plan (FilterTables::fusedR32).
// MIUPS_EXP_VGPR_CAP (experiment, profiles/r03_j_coresident.txt): cap the kernel's registers below 256 so that one low-register
// wave of another kernel (the frame pass) fits beside two transform waves on every SIMD
#if defined(MIUPS_EXP_VGPR_CAP) && !defined(MIUPS_HOST_EMU)
#define MI_VGPR_CAP __attribute__((amdgpu_num_vgpr(MIUPS_EXP_VGPR_CAP)))
#else
#define MI_VGPR_CAP
#endif
template <int LOG2K, bool EXT, int W = 2, bool R32 = false>
MI_GLOBAL MI_LAUNCH_BOUNDS((FusedCfg<LOG2K, W>::T < 64 ? 64 : FusedCfg<LOG2K, W>::T), (W == 1 ? 4 : MIUPS_WIDE_WAVES)) MI_VGPR_CAP void fused_kernel(
    Geometry g, IoDesc io, FusedTables ft) {
  MI_DYN_SHARED(cf, lds);
  FusedKernel<LOG2K, W, R32>::template run<false, EXT>(g, io, ft, lds);
}

// Small calls (fewer work items than half the CUs): every work item is shared by io.phase_parts workgroups, each taking
// P / phase_parts of its output phases (FusedKernel::run<..., PARTS>); the frames come from interleave_*_kernel.
template <int LOG2K>
MI_GLOBAL MI_LAUNCH_BOUNDS((FusedCfg<LOG2K, 2>::T < 64 ? 64 : FusedCfg<LOG2K, 2>::T), MIUPS_WIDE_WAVES) void fused_parts_kernel(
    Geometry g, IoDesc io, FusedTables ft) {
  MI_DYN_SHARED(cf, lds);
  FusedKernel<LOG2K, 2, false>::template run<false, true, true>(g, io, ft, lds);
}

// Block transform length 2 * 2^LOG2K (K = 32768 for the 2x filters at N = 131072): see
// FusedKernel::channel_block_split. Same launch shape and LDS as fused_kernel<LOG2K>.
template <int LOG2K>
MI_GLOBAL MI_LAUNCH_BOUNDS((FusedCfg<LOG2K>::T < 64 ? 64 : FusedCfg<LOG2K>::T), 2) void fused_split_kernel(Geometry g,
                                                                                                            IoDesc io,
                                                                                                            FusedTables ft) {
  MI_DYN_SHARED(cf, lds);
  FusedKernel<LOG2K, 2, false>::template run<true, true>(g, io, ft, lds);
}

// ... and its small-call form: io.phase_parts workgroups per channel-block, 2P / phase_parts half transforms each (each
// repeats both forward halves)
template <int LOG2K>
MI_GLOBAL MI_LAUNCH_BOUNDS((FusedCfg<LOG2K>::T < 64 ? 64 : FusedCfg<LOG2K>::T), 2) void fused_split_parts_kernel(Geometry g,
                                                                                                                  IoDesc io,
                                                                                                                  FusedTables ft) {
  MI_DYN_SHARED(cf, lds);
  FusedKernel<LOG2K, 2, false>::template run<true, true, true>(g, io, ft, lds);
}

}  // namespace miups
