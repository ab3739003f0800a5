// Register-resident radix-2/4/8/16 DFT kernels for the Stockham passes.
// DIR = -1 forward (e^{-j...}), +1 inverse (unnormalised).
//
// Each dftR works in place on R registers and leaves X[u] at register index
// out_pos<R>(u) (a compile-time permutation: every caller is fully unrolled,
// so the permutation costs no instruction).
#pragma once

#include "common.h"

namespace miups {

template <int R>
MI_DEVICE constexpr int out_pos(int u) {
  return R == 16 ? (4 * (u & 3) + (u >> 2)) : (R == 8 ? (2 * (u & 3) + (u >> 2)) : u);
}

// multiply by -j (forward) / +j (inverse): the W4^1 rotation
template <int DIR>
MI_DEVICE cf rot4(cf a) {
  return DIR < 0 ? cmulnj(a) : cmulj(a);
}

template <int DIR>
MI_DEVICE void dft2(cf &a0, cf &a1) {
  const cf t = a0;
  a0 = cadd(t, a1);
  a1 = csub(t, a1);
}

template <int DIR>
MI_DEVICE void dft4(cf &a0, cf &a1, cf &a2, cf &a3) {
  const cf t0 = cadd(a0, a2), t1 = csub(a0, a2);
  const cf t2 = cadd(a1, a3), t3 = rot4<DIR>(csub(a1, a3));
  a0 = cadd(t0, t2);
  a2 = csub(t0, t2);
  a1 = cadd(t1, t3);
  a3 = csub(t1, t3);
}

// a * W8^1 : forward (r, -r), inverse (r, +r)
template <int DIR>
MI_DEVICE cf mul_w8_1(cf a) {
  const float r = 0.70710678118654752440f;
  return DIR < 0 ? mk(r * (a.x + a.y), r * (a.y - a.x)) : mk(r * (a.x - a.y), r * (a.y + a.x));
}
// a * W8^3 : forward (-r, -r), inverse (-r, +r)
template <int DIR>
MI_DEVICE cf mul_w8_3(cf a) {
  const float r = 0.70710678118654752440f;
  return DIR < 0 ? mk(r * (a.y - a.x), -r * (a.x + a.y)) : mk(-r * (a.x + a.y), r * (a.x - a.y));
}
// a * W16^q for q in {1, 3, 9}; forward W16^q = (cos(q*pi/8), -sin(q*pi/8))
template <int DIR, int Q>
MI_DEVICE cf mul_w16(cf a) {
  const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f;
  const float wr = Q == 1 ? c1 : (Q == 3 ? s1 : -c1);
  const float wi_f = Q == 1 ? -s1 : (Q == 3 ? -c1 : s1);
  const float wi = DIR < 0 ? wi_f : -wi_f;
  return mk(a.x * wr - a.y * wi, a.x * wi + a.y * wr);
}

template <int DIR>
MI_DEVICE void dft8(cf *v) {
  dft4<DIR>(v[0], v[2], v[4], v[6]);
  dft4<DIR>(v[1], v[3], v[5], v[7]);
  // odd half times W8^{k1}; result of the 4-point DFTs sits at v[2*k1 (+1)]
  v[3] = mul_w8_1<DIR>(v[3]);
  v[5] = rot4<DIR>(v[5]);
  v[7] = mul_w8_3<DIR>(v[7]);
  MI_UNROLL
  for (int k = 0; k < 4; ++k) {
    dft2<DIR>(v[2 * k], v[2 * k + 1]);
  }
}

template <int DIR>
MI_DEVICE void dft16(cf *v) {
  MI_UNROLL
  for (int n2 = 0; n2 < 4; ++n2) {
    dft4<DIR>(v[n2], v[n2 + 4], v[n2 + 8], v[n2 + 12]);
  }
  // A[n2][k1] lives at v[n2 + 4*k1]; multiply by W16^{n2*k1}
  v[1 + 4] = mul_w16<DIR, 1>(v[1 + 4]);
  v[1 + 8] = mul_w8_1<DIR>(v[1 + 8]);
  v[1 + 12] = mul_w16<DIR, 3>(v[1 + 12]);
  v[2 + 4] = mul_w8_1<DIR>(v[2 + 4]);
  v[2 + 8] = rot4<DIR>(v[2 + 8]);
  v[2 + 12] = mul_w8_3<DIR>(v[2 + 12]);
  v[3 + 4] = mul_w16<DIR, 3>(v[3 + 4]);
  v[3 + 8] = mul_w8_3<DIR>(v[3 + 8]);
  v[3 + 12] = mul_w16<DIR, 9>(v[3 + 12]);
  MI_UNROLL
  for (int k1 = 0; k1 < 4; ++k1) {
    dft4<DIR>(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
  }
}

template <int DIR, int R>
MI_DEVICE void dftR(cf *v) {
  if constexpr (R == 2) {
    dft2<DIR>(v[0], v[1]);
  } else if constexpr (R == 4) {
    dft4<DIR>(v[0], v[1], v[2], v[3]);
  } else if constexpr (R == 8) {
    dft8<DIR>(v);
  } else {
    dft16<DIR>(v);
  }
}

// v[t] *= w^t for t = 1..R-1, powers built as a depth-log2(R) product tree
// from the table value w (|error| of w^t <= ~4 ulp). DIR > 0 conjugates.
template <int DIR, int R>
MI_DEVICE void apply_twiddles(cf *v, cf w) {
  if (DIR > 0) {
    w = cconj(w);
  }
  v[1] = cmul(v[1], w);
  if constexpr (R >= 4) {
    const cf w2 = cmul(w, w);
    const cf w3 = cmul(w2, w);
    v[2] = cmul(v[2], w2);
    v[3] = cmul(v[3], w3);
    if constexpr (R >= 8) {
      const cf w4 = cmul(w2, w2);
      const cf w5 = cmul(w4, w), w6 = cmul(w4, w2), w7 = cmul(w4, w3);
      v[4] = cmul(v[4], w4);
      v[5] = cmul(v[5], w5);
      v[6] = cmul(v[6], w6);
      v[7] = cmul(v[7], w7);
      if constexpr (R >= 16) {
        const cf w8 = cmul(w4, w4);
        v[8] = cmul(v[8], w8);
        v[9] = cmul(v[9], cmul(w8, w));
        v[10] = cmul(v[10], cmul(w8, w2));
        v[11] = cmul(v[11], cmul(w8, w3));
        v[12] = cmul(v[12], cmul(w8, w4));
        v[13] = cmul(v[13], cmul(w8, w5));
        v[14] = cmul(v[14], cmul(w8, w6));
        v[15] = cmul(v[15], cmul(w8, w7));
      }
    }
  }
}

// Same powers applied to the OUTPUTS of dftR (decimation in frequency): the
// u-th output lives at v[out_pos<R>(u)] and is multiplied by w^u.
template <int DIR, int R>
MI_DEVICE void apply_twiddles_out(cf *v, cf w) {
  if (DIR > 0) {
    w = cconj(w);
  }
  v[out_pos<R>(1)] = cmul(v[out_pos<R>(1)], w);
  if constexpr (R >= 4) {
    const cf w2 = cmul(w, w);
    const cf w3 = cmul(w2, w);
    v[out_pos<R>(2)] = cmul(v[out_pos<R>(2)], w2);
    v[out_pos<R>(3)] = cmul(v[out_pos<R>(3)], w3);
    if constexpr (R >= 8) {
      const cf w4 = cmul(w2, w2);
      const cf w5 = cmul(w4, w), w6 = cmul(w4, w2), w7 = cmul(w4, w3);
      v[out_pos<R>(4)] = cmul(v[out_pos<R>(4)], w4);
      v[out_pos<R>(5)] = cmul(v[out_pos<R>(5)], w5);
      v[out_pos<R>(6)] = cmul(v[out_pos<R>(6)], w6);
      v[out_pos<R>(7)] = cmul(v[out_pos<R>(7)], w7);
      if constexpr (R >= 16) {
        const cf w8 = cmul(w4, w4);
        v[out_pos<R>(8)] = cmul(v[out_pos<R>(8)], w8);
        v[out_pos<R>(9)] = cmul(v[out_pos<R>(9)], cmul(w8, w));
        v[out_pos<R>(10)] = cmul(v[out_pos<R>(10)], cmul(w8, w2));
        v[out_pos<R>(11)] = cmul(v[out_pos<R>(11)], cmul(w8, w3));
        v[out_pos<R>(12)] = cmul(v[out_pos<R>(12)], cmul(w8, w4));
        v[out_pos<R>(13)] = cmul(v[out_pos<R>(13)], cmul(w8, w5));
        v[out_pos<R>(14)] = cmul(v[out_pos<R>(14)], cmul(w8, w6));
        v[out_pos<R>(15)] = cmul(v[out_pos<R>(15)], cmul(w8, w7));
      }
    }
  }
}

// The fifteen powers w^1..w^15 of a radix-16 butterfly's twiddle (same product tree as
// apply_twiddles), for two butterflies that share them: t[u-1] = w^u. DIR > 0 conjugates.
template <int DIR>
MI_DEVICE void make_twiddles16(cf w, cf *t) {
  if (DIR > 0) {
    w = cconj(w);
  }
  t[0] = w;
  t[1] = cmul(w, w);
  t[2] = cmul(t[1], w);
  t[3] = cmul(t[1], t[1]);
  t[4] = cmul(t[3], w);
  t[5] = cmul(t[3], t[1]);
  t[6] = cmul(t[3], t[2]);
  t[7] = cmul(t[3], t[3]);
  MI_UNROLL
  for (int u = 1; u <= 7; ++u) {
    t[7 + u] = cmul(t[7], t[u - 1]);
  }
}
// v[u] *= t[u-1] (inputs of a DIT butterfly) / v[out_pos(u)] *= t[u-1] (outputs of a DIF one)
MI_DEVICE void mul_twiddles16(cf *v, const cf *t) {
  MI_UNROLL
  for (int u = 1; u < 16; ++u) {
    v[u] = cmul(v[u], t[u - 1]);
  }
}
MI_DEVICE void mul_twiddles16_out(cf *v, const cf *t) {
  MI_UNROLL
  for (int u = 1; u < 16; ++u) {
    v[out_pos<16>(u)] = cmul(v[out_pos<16>(u)], t[u - 1]);
  }
}

// Twiddle table layout: for q = 1..log2k, entries exp(-2*pi*i*k / 2^q) for
// k in [0, 2^(q-1)) start at offset 2^(q-1) - 1. (Total 2^log2k - 1 entries.)
MI_HD constexpr int tw_offset(int q) { return (1 << (q - 1)) - 1; }

}  // namespace miups
