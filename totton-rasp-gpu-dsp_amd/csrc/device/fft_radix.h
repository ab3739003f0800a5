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

// ---- packed form (see common.h): every helper below works on v2 register pairs -------------------------------
// rotation constant of the W4^1 step: -j a = a.yx * (1,-1) forward, +j a = a.yx * (-1,1) inverse
template <int DIR>
MI_DEVICE v2 rot_sign() {
  return DIR < 0 ? v2mk(1.0f, -1.0f) : v2mk(-1.0f, 1.0f);
}
template <int DIR>
MI_DEVICE v2 vrot4(v2 a) {
  return v2swap(a) * rot_sign<DIR>();
}
// a * (wr, wi) for a compile-time constant twiddle: a*(wr,wr) + a.yx*(-wi, wi)
MI_DEVICE v2 vmul_const(v2 a, float wr, float wi) { return v2fma(v2swap(a), v2mk(-wi, wi), a * v2mk(wr, wr)); }

template <int DIR>
MI_DEVICE void vdft2(v2 &a0, v2 &a1) {
  const v2 t = a0;
  a0 = t + a1;
  a1 = t - a1;
}

// 8 packed instructions: the +-j rotation rides on the two fma's
template <int DIR>
MI_DEVICE void vdft4(v2 &a0, v2 &a1, v2 &a2, v2 &a3) {
  const v2 t0 = a0 + a2, t1 = a0 - a2;
  const v2 t2 = a1 + a3, d = v2swap(a1 - a3);
  a0 = t0 + t2;
  a2 = t0 - t2;
  a1 = v2fma(d, rot_sign<DIR>(), t1);
  a3 = v2fma(d, -rot_sign<DIR>(), t1);
}

// a * W8^1 : forward (r, -r), inverse (r, +r);  a * W8^3 : forward (-r, -r), inverse (-r, +r)
template <int DIR>
MI_DEVICE v2 vmul_w8_1(v2 a) {
  const float r = 0.70710678118654752440f;
  return vmul_const(a, r, DIR < 0 ? -r : r);
}
template <int DIR>
MI_DEVICE v2 vmul_w8_3(v2 a) {
  const float r = 0.70710678118654752440f;
  return vmul_const(a, -r, DIR < 0 ? -r : r);
}
// a * W16^q for q in {1, 3, 9}; forward W16^q = (cos(q*pi/8), -sin(q*pi/8))
template <int DIR, int Q>
MI_DEVICE v2 vmul_w16(v2 a) {
  const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f;
  const float wr = Q == 1 ? c1 : (Q == 3 ? s1 : -c1);
  const float wi_f = Q == 1 ? -s1 : (Q == 3 ? -c1 : s1);
  return vmul_const(a, wr, DIR < 0 ? wi_f : -wi_f);
}

template <int DIR>
MI_DEVICE void vdft8(v2 *v) {
  vdft4<DIR>(v[0], v[2], v[4], v[6]);
  vdft4<DIR>(v[1], v[3], v[5], v[7]);
  // odd half times W8^{k1}; result of the 4-point DFTs sits at v[2*k1 (+1)]
  v[3] = vmul_w8_1<DIR>(v[3]);
  v[5] = vrot4<DIR>(v[5]);
  v[7] = vmul_w8_3<DIR>(v[7]);
  MI_UNROLL
  for (int k = 0; k < 4; ++k) {
    vdft2<DIR>(v[2 * k], v[2 * k + 1]);
  }
}

template <int DIR>
MI_DEVICE void vdft16(v2 *v) {
  MI_UNROLL
  for (int n2 = 0; n2 < 4; ++n2) {
    vdft4<DIR>(v[n2], v[n2 + 4], v[n2 + 8], v[n2 + 12]);
  }
  // A[n2][k1] lives at v[n2 + 4*k1]; multiply by W16^{n2*k1}
  v[1 + 4] = vmul_w16<DIR, 1>(v[1 + 4]);
  v[1 + 8] = vmul_w8_1<DIR>(v[1 + 8]);
  v[1 + 12] = vmul_w16<DIR, 3>(v[1 + 12]);
  v[2 + 4] = vmul_w8_1<DIR>(v[2 + 4]);
  v[2 + 8] = vrot4<DIR>(v[2 + 8]);
  v[2 + 12] = vmul_w8_3<DIR>(v[2 + 12]);
  v[3 + 4] = vmul_w16<DIR, 3>(v[3 + 4]);
  v[3 + 8] = vmul_w8_3<DIR>(v[3 + 8]);
  v[3 + 12] = vmul_w16<DIR, 9>(v[3 + 12]);
  MI_UNROLL
  for (int k1 = 0; k1 < 4; ++k1) {
    vdft4<DIR>(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
  }
}

template <int DIR, int R>
MI_DEVICE void vdftR(v2 *v) {
  if constexpr (R == 2) {
    vdft2<DIR>(v[0], v[1]);
  } else if constexpr (R == 4) {
    vdft4<DIR>(v[0], v[1], v[2], v[3]);
  } else if constexpr (R == 8) {
    vdft8<DIR>(v);
  } else {
    vdft16<DIR>(v);
  }
}

// The R-1 powers w^1..w^(R-1) of a butterfly's twiddle as a depth-log2(R) product tree from the table value w
// (|error| of w^t <= ~4 ulp): t[u-1] = w^u. DIR > 0 conjugates. Two packed instructions per power.
template <int DIR, int R>
MI_DEVICE void make_twiddles(v2 w, v2 *t) {
  if (DIR > 0) {
    w = vconj(w);
  }
  t[0] = w;
  if constexpr (R >= 4) {
    t[1] = vmul(w, w);
    t[2] = vmul(t[1], w);
    if constexpr (R >= 8) {
      t[3] = vmul(t[1], t[1]);
      MI_UNROLL
      for (int u = 1; u <= 3; ++u) {
        t[3 + u] = vmul(t[3], t[u - 1]);
      }
      if constexpr (R >= 16) {
        t[7] = vmul(t[3], t[3]);
        MI_UNROLL
        for (int u = 1; u <= 7; ++u) {
          t[7 + u] = vmul(t[7], t[u - 1]);
        }
      }
    }
  }
}

// ---- cf-array faces used by the kernels (registers: the conversions are free) -----------------------------------
template <int DIR>
MI_DEVICE void dft16(cf *c) {
  v2 v[16];
  MI_UNROLL
  for (int i = 0; i < 16; ++i) {
    v[i] = V(c[i]);
  }
  vdft16<DIR>(v);
  MI_UNROLL
  for (int i = 0; i < 16; ++i) {
    c[i] = C(v[i]);
  }
}

template <int DIR, int R>
MI_DEVICE void dftR(cf *c) {
  v2 v[R];
  MI_UNROLL
  for (int i = 0; i < R; ++i) {
    v[i] = V(c[i]);
  }
  vdftR<DIR, R>(v);
  MI_UNROLL
  for (int i = 0; i < R; ++i) {
    c[i] = C(v[i]);
  }
}

// v[t] *= w^t for t = 1..R-1 (inputs of a decimation-in-time butterfly). DIR > 0 conjugates.
template <int DIR, int R>
MI_DEVICE void apply_twiddles(cf *v, cf w) {
  v2 t[R - 1];
  make_twiddles<DIR, R>(V(w), t);
  MI_UNROLL
  for (int u = 1; u < R; ++u) {
    v[u] = C(vmul(V(v[u]), t[u - 1]));
  }
}

// Same powers applied to the OUTPUTS of dftR (decimation in frequency): the
// u-th output lives at v[out_pos<R>(u)] and is multiplied by w^u.
template <int DIR, int R>
MI_DEVICE void apply_twiddles_out(cf *v, cf w) {
  v2 t[R - 1];
  make_twiddles<DIR, R>(V(w), t);
  MI_UNROLL
  for (int u = 1; u < R; ++u) {
    v[out_pos<R>(u)] = C(vmul(V(v[out_pos<R>(u)]), t[u - 1]));
  }
}

// The fifteen powers of a radix-16 butterfly's twiddle for two butterflies that share them.
struct Tw16 {
  v2 t[15];
};
template <int DIR>
MI_DEVICE void make_twiddles16(cf w, Tw16 &tw) {
  make_twiddles<DIR, 16>(V(w), tw.t);
}
// v[u] *= w^u (inputs of a DIT butterfly) / v[out_pos(u)] *= w^u (outputs of a DIF one)
MI_DEVICE void mul_twiddles16(cf *v, const Tw16 &tw) {
  MI_UNROLL
  for (int u = 1; u < 16; ++u) {
    v[u] = C(vmul(V(v[u]), tw.t[u - 1]));
  }
}
MI_DEVICE void mul_twiddles16_out(cf *v, const Tw16 &tw) {
  MI_UNROLL
  for (int u = 1; u < 16; ++u) {
    v[out_pos<16>(u)] = C(vmul(V(v[out_pos<16>(u)]), tw.t[u - 1]));
  }
}

// Twiddle table layout: for q = 1..log2k, entries exp(-2*pi*i*k / 2^q) for
// k in [0, 2^(q-1)) start at offset 2^(q-1) - 1. (Total 2^log2k - 1 entries.)
MI_HD constexpr int tw_offset(int q) { return (1 << (q - 1)) - 1; }

}  // namespace miups
