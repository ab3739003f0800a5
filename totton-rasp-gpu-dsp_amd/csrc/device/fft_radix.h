// Register-resident radix-2/4/8/16/32 DFT kernels for the in-LDS and staged passes.
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
  // R == 32: one radix-2 step (decimation in frequency) in front of two sixteen-point DFTs: the even outputs come out
  // of the first one, the odd outputs of the second
  return R == 32 ? (16 * (u & 1) + (4 * ((u >> 1) & 3) + (u >> 3)))
                 : (R == 16 ? (4 * (u & 3) + (u >> 2)) : (R == 8 ? (2 * (u & 3) + (u >> 2)) : u));
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

// The same sixteen-point DFT with its outputs handed out as the last stage produces them: after the k1-th final
// butterfly emit(u, X[u]) is called for u = k1, k1 + 4, k1 + 8, k1 + 12, then a scheduling fence -- the caller's stores
// of those four values are issued THERE, not in one burst behind the whole butterfly. (The LDS write path is the
// slowest resource of a pass, ~80 B/clk: a thread's last sixteen stores issued together drain for ~0.8k cycles after
// the workgroup has finished computing, in front of every barrier; profiles/r03_b_*.)
template <int DIR, typename F>
MI_DEVICE void vdft16_emit(v2 *v, F &&emit) {
  MI_UNROLL
  for (int n2 = 0; n2 < 4; ++n2) {
    vdft4<DIR>(v[n2], v[n2 + 4], v[n2 + 8], v[n2 + 12]);
  }
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
    MI_UNROLL
    for (int k2 = 0; k2 < 4; ++k2) {
      emit(k1 + 4 * k2, v[4 * k1 + k2]);
    }
    MI_SCHED_FENCE();
  }
}

// a * W32^j, j = 1..15 (forward W32^j = (cos(j*pi/16), -sin(j*pi/16)))
template <int DIR, int J>
MI_DEVICE v2 vmul_w32(v2 a) {
  constexpr float c[16] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                           0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f,
                           0.19509032201612826785f, 0.0f, -0.19509032201612826785f, -0.38268343236508977173f,
                           -0.55557023301960222474f, -0.70710678118654752440f, -0.83146961230254523708f,
                           -0.92387953251128675613f, -0.98078528040323044913f};
  constexpr float s[16] = {0.0f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f,
                           0.70710678118654752440f, 0.83146961230254523708f, 0.92387953251128675613f,
                           0.98078528040323044913f, 1.0f, 0.98078528040323044913f, 0.92387953251128675613f,
                           0.83146961230254523708f, 0.70710678118654752440f, 0.55557023301960222474f,
                           0.38268343236508977173f, 0.19509032201612826785f};
  if constexpr (J == 8) {
    return vrot4<DIR>(a);
  } else {
    return vmul_const(a, c[J], DIR < 0 ? -s[J] : s[J]);
  }
}

// 32 points as one radix-2 decimation-in-frequency step and two sixteen-point DFTs:
//   X[2v] = DFT16(x[j] + x[j+16])_v,  X[2v+1] = DFT16((x[j] - x[j+16]) W32^j)_v;  X[u] ends up at v[out_pos<32>(u)]
template <int DIR>
MI_DEVICE void vdft32(v2 *v) {
  MI_UNROLL
  for (int j = 0; j < 16; ++j) {
    const v2 t = v[j];
    v[j] = t + v[j + 16];
    v[j + 16] = t - v[j + 16];
  }
  v[17] = vmul_w32<DIR, 1>(v[17]);
  v[18] = vmul_w32<DIR, 2>(v[18]);
  v[19] = vmul_w32<DIR, 3>(v[19]);
  v[20] = vmul_w32<DIR, 4>(v[20]);
  v[21] = vmul_w32<DIR, 5>(v[21]);
  v[22] = vmul_w32<DIR, 6>(v[22]);
  v[23] = vmul_w32<DIR, 7>(v[23]);
  v[24] = vmul_w32<DIR, 8>(v[24]);
  v[25] = vmul_w32<DIR, 9>(v[25]);
  v[26] = vmul_w32<DIR, 10>(v[26]);
  v[27] = vmul_w32<DIR, 11>(v[27]);
  v[28] = vmul_w32<DIR, 12>(v[28]);
  v[29] = vmul_w32<DIR, 13>(v[29]);
  v[30] = vmul_w32<DIR, 14>(v[30]);
  v[31] = vmul_w32<DIR, 15>(v[31]);
  vdft16<DIR>(v);
  vdft16<DIR>(v + 16);
}

// ... and the 32-point form: emit(u, X[u]) in eight groups of four
template <int DIR, typename F>
MI_DEVICE void vdft32_emit(v2 *v, F &&emit) {
  MI_UNROLL
  for (int j = 0; j < 16; ++j) {
    const v2 t = v[j];
    v[j] = t + v[j + 16];
    v[j + 16] = t - v[j + 16];
  }
  vdft16_emit<DIR>(v, [&](int u, v2 y) { emit(2 * u, y); });
  v[17] = vmul_w32<DIR, 1>(v[17]);
  v[18] = vmul_w32<DIR, 2>(v[18]);
  v[19] = vmul_w32<DIR, 3>(v[19]);
  v[20] = vmul_w32<DIR, 4>(v[20]);
  v[21] = vmul_w32<DIR, 5>(v[21]);
  v[22] = vmul_w32<DIR, 6>(v[22]);
  v[23] = vmul_w32<DIR, 7>(v[23]);
  v[24] = vmul_w32<DIR, 8>(v[24]);
  v[25] = vmul_w32<DIR, 9>(v[25]);
  v[26] = vmul_w32<DIR, 10>(v[26]);
  v[27] = vmul_w32<DIR, 11>(v[27]);
  v[28] = vmul_w32<DIR, 12>(v[28]);
  v[29] = vmul_w32<DIR, 13>(v[29]);
  v[30] = vmul_w32<DIR, 14>(v[30]);
  v[31] = vmul_w32<DIR, 15>(v[31]);
  vdft16_emit<DIR>(v + 16, [&](int u, v2 y) { emit(2 * u + 1, y); });
}

template <int DIR, int R>
MI_DEVICE void vdftR(v2 *v) {
  if constexpr (R == 32) {
    vdft32<DIR>(v);
  } else if constexpr (R == 2) {
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
        if constexpr (R >= 32) {
          t[15] = vmul(t[7], t[7]);
          MI_UNROLL
          for (int u = 1; u <= 15; ++u) {
            t[15 + u] = vmul(t[15], t[u - 1]);
          }
        }
      }
    }
  }
}

// ---- pruned forms: only the UPPER half of the outputs, X[u] for u >= R/2 (left at v[out_pos<R>(u)]; the other
// positions hold garbage). For the last inverse pass of an overlap-save block whose discarded history is at least half
// the transform (every shipped filter: O/N = 0.61), output u of butterfly q is sample q + u*K/R: the lower half is
// never kept. A DFT gives up little to output pruning -- only its LAST stage shrinks: radix 2 one add of
// two, radix 4 six packed operations of eight, radix 16 the last stage's four butterflies drop two outputs each.
template <int DIR>
MI_DEVICE void vdft4_upper(v2 &a0, v2 &a1, v2 &a2, v2 &a3) {
  const v2 t0 = a0 + a2, t1 = a0 - a2;
  const v2 t2 = a1 + a3, d = v2swap(a1 - a3);
  a2 = t0 - t2;
  a3 = v2fma(d, -rot_sign<DIR>(), t1);
}
template <int DIR, int R>
MI_DEVICE void vdftR_upper(v2 *v) {
  if constexpr (R == 2) {
    v[1] = v[0] - v[1];
  } else if constexpr (R == 4) {
    vdft4_upper<DIR>(v[0], v[1], v[2], v[3]);
  } else if constexpr (R == 8) {
    vdft4<DIR>(v[0], v[2], v[4], v[6]);
    vdft4<DIR>(v[1], v[3], v[5], v[7]);
    v[3] = vmul_w8_1<DIR>(v[3]);
    v[5] = vrot4<DIR>(v[5]);
    v[7] = vmul_w8_3<DIR>(v[7]);
    MI_UNROLL
    for (int k = 0; k < 4; ++k) {
      v[2 * k + 1] = v[2 * k] - v[2 * k + 1];  // X[k + 4] sits at out_pos<8>(k + 4) = 2 k + 1
    }
  } else {
    static_assert(R == 16, "pruned DFTs exist for radix 2, 4, 8, 16");
    MI_UNROLL
    for (int n2 = 0; n2 < 4; ++n2) {
      vdft4<DIR>(v[n2], v[n2 + 4], v[n2 + 8], v[n2 + 12]);
    }
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
    for (int k1 = 0; k1 < 4; ++k1) {  // X[k1 + 4 k2] at v[4 k1 + k2]: k2 = 2, 3 are the upper half
      vdft4_upper<DIR>(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
    }
  }
}
template <int DIR, int R>
MI_DEVICE void dftR_upper(cf *c) {
  v2 v[R];
  MI_UNROLL
  for (int i = 0; i < R; ++i) {
    v[i] = V(c[i]);
  }
  vdftR_upper<DIR, R>(v);
  MI_UNROLL
  for (int i = 0; i < R; ++i) {
    c[i] = C(v[i]);
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
// R == 32: the powers 17..31 are formed one at a time from w^16 and the first fifteen (the same 30 products as the full
// tree, but only 16 of them live at once: the radix-32 butterfly itself holds 64 registers)
template <int DIR, int R>
MI_DEVICE void apply_twiddles(cf *v, cf w) {
  if constexpr (R == 32) {
    v2 t[15];
    make_twiddles<DIR, 16>(V(w), t);
    MI_UNROLL
    for (int u = 1; u < 16; ++u) {
      v[u] = C(vmul(V(v[u]), t[u - 1]));
    }
    const v2 w16 = vmul(t[7], t[7]);
    v[16] = C(vmul(V(v[16]), w16));
    MI_UNROLL
    for (int u = 1; u < 16; ++u) {
      v[16 + u] = C(vmul(V(v[16 + u]), vmul(w16, t[u - 1])));
    }
    return;
  }
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
  if constexpr (R == 32) {
    v2 t[15];
    make_twiddles<DIR, 16>(V(w), t);
    MI_UNROLL
    for (int u = 1; u < 16; ++u) {
      v[out_pos<32>(u)] = C(vmul(V(v[out_pos<32>(u)]), t[u - 1]));
    }
    const v2 w16 = vmul(t[7], t[7]);
    v[out_pos<32>(16)] = C(vmul(V(v[out_pos<32>(16)]), w16));
    MI_UNROLL
    for (int u = 1; u < 16; ++u) {
      v[out_pos<32>(16 + u)] = C(vmul(V(v[out_pos<32>(16 + u)]), vmul(w16, t[u - 1])));
    }
    return;
  }
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
