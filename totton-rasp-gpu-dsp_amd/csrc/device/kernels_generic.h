// Any-size path: the same algorithm as the fused kernel (kernel_fused.h) cut
// into one launch per stage with the work arrays in HBM. Used when the
// per-channel-block transform does not fit one CU's LDS (K > 16384, e.g. the
// 2x filters at N = 131072 and the "2m" filters) and for degenerate tiny
// geometries (K < 32). Same tables, same arithmetic, same results.
//
// Work item `it` = one channel-block: it = blk * (streams*channels) + sc.
#pragma once

#include "common.h"
#include "fft_radix.h"
#include "pcm.h"

namespace miups {

// ---- sample fetch shared with the fused kernel ---------------------------
// frame f is relative to the first new input frame of this call; f < 0 reads
// the carried history (the last hist_frames frames of earlier calls).
MI_DEVICE float fetch_frame(const Geometry &g, const IoDesc &io, int stream, int ch, long long f) {
  if (f >= 0) {
    const char *base = static_cast<const char *>(io.in) + stream * io.in_stream_stride;
    return pcm_load(base, io.in_fmt, f * io.channels + ch);
  }
  const char *base = static_cast<const char *>(io.hist) + stream * io.hist_stream_stride;
  return pcm_load(base, io.in_fmt, (g.hist_frames + f) * io.channels + ch);
}

// compact-domain sample n in [0, M) of block blk (reference: timeBuffer
// assembly, src/vulkan/vulkan_streaming_upsampler.cpp:528-534, with the
// overlap kept as raw input frames instead of zero-stuffed floats).
MI_DEVICE float compact_sample(const Geometry &g, const IoDesc &io, int stream, int ch, int blk, int n) {
  const long long pos = static_cast<long long>(blk) * g.Bc + n - g.Oc;
  if (g.S == 1) {
    return fetch_frame(g, io, stream, ch, pos);
  }
  long long q = pos / g.S;
  long long r = pos % g.S;
  if (r < 0) {
    r += g.S;
    q -= 1;
  }
  return r == 0 ? fetch_frame(g, io, stream, ch, q) : 0.0f;
}

// output sample: compact index n (>= Oc) of phase p -> frame blk*B + (n-Oc)*P + p
MI_DEVICE void store_output(const Geometry &g, const IoDesc &io, int stream, int ch, int blk, int p, int n, float v) {
  if (n < g.Oc) {
    return;
  }
  const long long frame = static_cast<long long>(blk) * g.B + static_cast<long long>(n - g.Oc) * g.P + p;
  char *base = static_cast<char *>(io.out) + stream * io.out_stream_stride;
  pcm_store(base, io.out_fmt, frame * io.channels + ch, v);
}

// ---- stage 1: z[n'] = x[2n'] + j x[2n'+1] --------------------------------
MI_GLOBAL void gen_load_kernel(Geometry g, IoDesc io, cf *MI_RESTRICT work, int item0, int nitems) {
  const long long gid = static_cast<long long>(MI_BID_X) * MI_BDIM_X + MI_TID_X;
  const long long total = static_cast<long long>(nitems) * g.K;
  if (gid >= total) {
    return;
  }
  const int it = static_cast<int>(gid / g.K);
  const int n = static_cast<int>(gid % g.K);
  const int sc_count = io.streams * io.channels;
  const int blk = (item0 + it) / sc_count;
  const int sc = (item0 + it) % sc_count;
  const int s = sc / io.channels, c = sc % io.channels;
  work[gid] = mk(compact_sample(g, io, s, c, blk, 2 * n), compact_sample(g, io, s, c, blk, 2 * n + 1));
}

// ---- stage 2/4: one Stockham radix-R pass over a batch of length-K rows ---
//   out[(j-k)*R + k + u*Ns] = DFT_R( in[j + t*K/R] * W_{Ns*R}^{t*k} ), k = j mod Ns
template <int DIR, int R>
MI_GLOBAL void gen_pass_kernel(const cf *MI_RESTRICT in, cf *MI_RESTRICT out, const cf *MI_RESTRICT tw, int K, int Ns,
                               int log2NsR, long long rows) {
  const long long gid = static_cast<long long>(MI_BID_X) * MI_BDIM_X + MI_TID_X;
  const int per_row = K / R;
  if (gid >= rows * per_row) {
    return;
  }
  const long long row = gid / per_row;
  const int j = static_cast<int>(gid % per_row);
  const cf *src = in + row * K;
  cf *dst = out + row * K;
  cf v[R];
  MI_UNROLL
  for (int t = 0; t < R; ++t) {
    v[t] = src[j + t * per_row];
  }
  const int k = j & (Ns - 1);
  if (Ns > 1) {
    apply_twiddles<DIR, R>(v, tw[tw_offset(log2NsR) + k]);
  }
  dftR<DIR, R>(v);
  const int base = (j - k) * R + k;
  MI_UNROLL
  for (int u = 0; u < R; ++u) {
    dst[base + u * Ns] = v[out_pos<R>(u)];
  }
}

// ---- stage 3: untangle (real-FFT split), multiply by the phase spectra,
// re-tangle for the packed inverse. Per bin k of phase p:
//   u = Z[k], v = conj Z[(K-k) mod K], W = W_M^k
//   Xa = (u+v) - jW(u-v)          (= 2 X[k])
//   Xb = (u+v) + jW(u-v)          (= 2 conj X[K-k])
//   P = Xa * Gs_p[k],  Q = Xb * Gc_p[k]          Gs = G/(2M), Gc[k] = conj Gs[K-k]
//   Z'_p[k] = (P+Q) + j conj(W) (P-Q)
// This is the reference's `freq[i] *= filterSpectrum_[i]`
// (vulkan_streaming_upsampler.cpp:553-558,583-585) in the polyphase basis.
MI_DEVICE cf spectral_bin(cf u, cf zm, cf W, cf gs, cf gc) {
  const cf v = cconj(zm);
  const cf s = cadd(u, v);
  const cf d = cmulj(cmul(W, csub(u, v)));
  const cf P = cmul(csub(s, d), gs);
  const cf Q = cmul(cadd(s, d), gc);
  return cadd(cadd(P, Q), cmulj(cmulc(csub(P, Q), W)));
}

MI_GLOBAL void gen_multiply_kernel(Geometry g, const cf *MI_RESTRICT Z, cf *MI_RESTRICT Zp, const cf *MI_RESTRICT Gs,
                                   const cf *MI_RESTRICT Gc, const cf *MI_RESTRICT Wm, int nitems) {
  const long long gid = static_cast<long long>(MI_BID_X) * MI_BDIM_X + MI_TID_X;
  const long long total = static_cast<long long>(nitems) * g.K;
  if (gid >= total) {
    return;
  }
  const long long it = gid / g.K;
  const int k = static_cast<int>(gid % g.K);
  const cf u = Z[it * g.K + k];
  const cf zm = Z[it * g.K + ((g.K - k) & (g.K - 1))];
  const cf W = Wm[k];
  for (int p = 0; p < g.P; ++p) {
    Zp[(it * g.P + p) * g.K + k] =
        spectral_bin(u, zm, W, Gs[static_cast<long long>(p) * g.K + k], Gc[static_cast<long long>(p) * g.K + k]);
  }
}

// ---- stage 5: overlap-discard + phase interleave + PCM store --------------
// (reference: output[i] = Re freq[overlap + i], :566-569 / :588-591)
MI_GLOBAL void gen_store_kernel(Geometry g, IoDesc io, const cf *MI_RESTRICT y, int item0, int nitems) {
  const long long gid = static_cast<long long>(MI_BID_X) * MI_BDIM_X + MI_TID_X;
  const long long total = static_cast<long long>(nitems) * g.P * g.K;
  if (gid >= total) {
    return;
  }
  const int n = static_cast<int>(gid % g.K);
  const long long ip = gid / g.K;
  const int p = static_cast<int>(ip % g.P);
  const int it = static_cast<int>(ip / g.P);
  if (2 * n + 1 < g.Oc) {
    return;
  }
  const int sc_count = io.streams * io.channels;
  const int blk = (item0 + it) / sc_count;
  const int sc = (item0 + it) % sc_count;
  const int s = sc / io.channels, c = sc % io.channels;
  const cf v = y[gid];
  store_output(g, io, s, c, blk, p, 2 * n, v.x);
  store_output(g, io, s, c, blk, p, 2 * n + 1, v.y);
}

// ---- interleaved PCM (history ++ new frames) -> one fp32 timeline per channel ----
// Pre-pass of the fused path for streams with more than two channels: a frame is then
// wider than one vector load and a per-channel gather would touch one cache line per
// sample. A workgroup converts kPlanarTile frames: coalesced reads in memory order, a
// transpose through LDS ([channel][kPlanarTile + 1] floats), coalesced 256-byte writes
// per channel. planar[(s*channels + c)*plane_floats + t], t = hist_frames + frame.
// (reference: the per-channel de-interleave loop, alsa_streamer_main.cpp:315-321)
constexpr int kPlanarTile = 64;  // smallest tile; planar_tile_frames() picks the size for a channel count
// frames per workgroup (a power of two): a 64-frame tile of 8 channels is 2 KiB in, 2 KiB out -- the launch ran at 2 TB/s
MI_HD int planar_tile_frames(int channels) {
  int pow2 = 1;  // channels rounded up to a power of two
  while (pow2 < channels) {
    pow2 <<= 1;
  }
  const int tf = 8192 / pow2;  // about 8192 samples (32 KiB in, 32 KiB out, 32 KiB of LDS) per workgroup
  return tf < 64 ? 64 : (tf > 4096 ? 4096 : tf);
}
MI_GLOBAL void planarize_kernel(Geometry g, IoDesc io, float *MI_RESTRICT planar, long long plane_floats,
                                long long total_frames, int tiles_per_stream, int tile_frames) {
  MI_DYN_SHARED(float, tile);
  const int TF = tile_frames;
  const int s = static_cast<int>(MI_BID_X) / tiles_per_stream;
  const long long t0 = static_cast<long long>(static_cast<int>(MI_BID_X) - s * tiles_per_stream) * TF;
  const int C = io.channels, n = TF * C;
  const char *hist = static_cast<const char *>(io.hist) + s * io.hist_stream_stride;
  const char *in = static_cast<const char *>(io.in) + s * io.in_stream_stride;
  const int bd = MI_BDIM_X;
  // Fast path (whole tile inside the history or inside the new input, 4-byte samples, 16-byte aligned): the tile is
  // TF*C consecutive samples in memory -- a flat copy in 16-byte words, up to 8 per thread, all requested before the
  // first LDS write; then every row leaves as lane-contiguous 16-byte words. (The general loop below pays one
  // round trip per element and thread: 73 us for 32 channels x 837k frames, 2.9 TB/s.)
  const bool in_hist = t0 + TF <= g.hist_frames, in_new = t0 >= g.hist_frames && t0 + TF <= total_frames;
  const char *flat = in_hist ? hist + t0 * C * 4 : in + (t0 - g.hist_frames) * C * 4;
  if ((in_hist || in_new) && (io.in_fmt == kS32 || io.in_fmt == kF32) && (reinterpret_cast<uintptr_t>(flat) & 15) == 0 &&
      n <= 8 * 4 * bd && (plane_floats & 3) == 0 && (reinterpret_cast<uintptr_t>(planar) & 15) == 0) {
    struct alignas(16) W4 {
      int32_t w[4];
    };
    W4 v[8];
    MI_UNROLL
    for (int k = 0; k < 8; ++k) {
      const int e = 4 * (static_cast<int>(MI_TID_X) + k * bd);
      if (e < n) {
        v[k] = *reinterpret_cast<const W4 *>(flat + static_cast<size_t>(e) * 4);
      }
    }
    MI_UNROLL
    for (int k = 0; k < 8; ++k) {
      const int e = 4 * (static_cast<int>(MI_TID_X) + k * bd);
      if (e < n) {
        int f = e / C, c = e - f * C;
        MI_UNROLL
        for (int j = 0; j < 4; ++j) {
          const float x = io.in_fmt == kF32 ? __builtin_bit_cast(float, v[k].w[j])
                                            : static_cast<float>(v[k].w[j]) * (1.0f / 2147483648.0f);
          tile[c * (TF + 1) + f] = x;
          if (++c == C) {
            c = 0;
            ++f;
          }
        }
      }
    }
    MI_SYNC();
    const int lq = 31 - __builtin_clz(static_cast<unsigned>(TF)) - 2;  // log2(TF / 4)
    for (int e = MI_TID_X; e < (n >> 2); e += bd) {
      const int c2 = e >> lq, f2 = (e & ((TF >> 2) - 1)) << 2;
      const float *row = tile + c2 * (TF + 1) + f2;
      float *plane = planar + (static_cast<long long>(s) * C + c2) * plane_floats;
      const long long t = t0 + f2;  // a multiple of 4
      if (io.split_planes) {
        *reinterpret_cast<cf *>(plane + (t >> 1)) = mk(row[0], row[1]);
        *reinterpret_cast<cf *>(plane + (plane_floats >> 1) + (t >> 1)) = mk(row[2], row[3]);
      } else {
        *reinterpret_cast<f4 *>(plane + t) = f4{row[0], row[1], row[2], row[3]};
      }
    }
    return;
  }
  // (frame, channel) of element e = tid + k * threads without a division per element: the kernel is a pure copy and its
  // index arithmetic was most of its instructions
  const int dq = bd / C, dr = bd - dq * C;
  int f = static_cast<int>(MI_TID_X) / C, c = static_cast<int>(MI_TID_X) - f * C;
  for (int e = MI_TID_X; e < n; e += bd) {
    const long long t = t0 + f;
    if (t < total_frames) {
      const float v = t < g.hist_frames ? pcm_load(hist, io.in_fmt, t * C + c)
                                        : pcm_load(in, io.in_fmt, (t - g.hist_frames) * C + c);
      tile[c * (TF + 1) + f] = v;
    }
    f += dq;
    c += dr;
    if (c >= C) {
      c -= C;
      ++f;
    }
  }
  MI_SYNC();
  const int lt = 31 - __builtin_clz(static_cast<unsigned>(TF));  // TF is a power of two
  for (int e = MI_TID_X; e < n; e += bd) {
    const int c2 = e >> lt, f2 = e & (TF - 1);
    const long long t = t0 + f2;
    if (t < total_frames) {
      // io.split_planes: split-planar order (see make_block_io) -- even complex words in the first half of the plane,
      // odd ones in the second
      const long long at = io.split_planes ? ((t & 2) ? (plane_floats >> 1) : 0) + ((t >> 2) << 1) + (t & 1) : t;
      planar[(static_cast<long long>(s) * C + c2) * plane_floats + at] = tile[c2 * (TF + 1) + f2];
    }
  }
}

// ---- staging planes of a chunk of (stream, block) pairs -> interleaved PCM frames ----
// planes[((jb*C + c)*P + p)*Bc + i] = y_p[Oc + i] of channel c of the chunk's jb-th
// (stream, block) pair sb0 + jb; output frame blk*B + i*P + p, channel c (reference:
// interleave + ConvertFloatToPcm, alsa_streamer_main.cpp:327-329,550-552).
// For one i the P*C values (p-major, channel-minor) are R = P*C consecutive output samples.
//
// Vector form (4-byte output, R % 4 == 0, Bc % 4 == 0, 16-byte aligned rows): a unit is 4
// consecutive i of the 4 planes that make one 16-byte run; 4 x 16-byte loads, a 4x4
// transpose in registers, 4 x 16-byte stores. Lanes run over the R/4 runs of an i first:
// every store instruction writes R*4 contiguous bytes per i.
template <int FMT>
MI_GLOBAL void interleave_quad_kernel(Geometry g, IoDesc io, const float *MI_RESTRICT planes, int sb0, int nb,
                                      int wgs_per_pair) {
  constexpr int kUnits = 4;
  const int jb = static_cast<int>(MI_BID_X) / wgs_per_pair;
  if (jb >= nb) {
    return;
  }
  const int wg = static_cast<int>(MI_BID_X) - jb * wgs_per_pair;
  const int C = io.channels, P = g.P, Rq = (P * C) >> 2;
  const int units = (g.Bc >> 2) * Rq;
  const int sb = sb0 + jb, s = sb / io.blocks, blk = sb - s * io.blocks;
  const float *src = planes + static_cast<long long>(jb) * C * g.P * g.Bp;
  char *out_blk = static_cast<char *>(io.out) + s * io.out_stream_stride + static_cast<long long>(blk) * g.B * C * 4;
  const long long i_step = static_cast<long long>(P) * C * 4;  // bytes from i to i + 1
  const int base = wg * static_cast<int>(MI_BDIM_X) * kUnits + static_cast<int>(MI_TID_X);
  f4 v[kUnits][4];
  MI_UNROLL
  for (int d = 0; d < kUnits; ++d) {
    const int u = base + d * static_cast<int>(MI_BDIM_X);
    if (u < units) {
      const int iq = u / Rq, r0 = (u - iq * Rq) << 2;
      MI_UNROLL
      for (int e = 0; e < 4; ++e) {
        const int r = r0 + e, p = r / C, c = r - p * C;
        const float *pl = src + (static_cast<long long>(c) * P + p) * g.Bp;
        if (io.split_planes) {
          // halves of the split form: (y[4m], y[4m+1]) pairs, then (y[4m+2], y[4m+3]) pairs
          const cf lo = *reinterpret_cast<const cf *>(pl + 2 * iq);
          const cf hi = *reinterpret_cast<const cf *>(pl + (g.Bc >> 1) + 2 * iq);
          v[d][e] = f4{lo.x, lo.y, hi.x, hi.y};
        } else {
          v[d][e] = *reinterpret_cast<const f4 *>(pl + 4 * iq);
        }
      }
    }
  }
  MI_UNROLL
  for (int d = 0; d < kUnits; ++d) {
    const int u = base + d * static_cast<int>(MI_BDIM_X);
    if (u < units) {
      const int iq = u / Rq, r0 = (u - iq * Rq) << 2;
      char *dst = out_blk + static_cast<long long>(4 * iq) * i_step + r0 * 4;
      const float m[4][4] = {{v[d][0].x, v[d][1].x, v[d][2].x, v[d][3].x},
                             {v[d][0].y, v[d][1].y, v[d][2].y, v[d][3].y},
                             {v[d][0].z, v[d][1].z, v[d][2].z, v[d][3].z},
                             {v[d][0].w, v[d][1].w, v[d][2].w, v[d][3].w}};
      MI_UNROLL
      for (int e = 0; e < 4; ++e) {
        if constexpr (FMT == kF32) {
          *reinterpret_cast<f4 *>(dst + e * i_step) = f4{m[e][0], m[e][1], m[e][2], m[e][3]};
        } else {
          struct alignas(16) I4 {
            int32_t a, b, c, d;
          };
          I4 o;
          o.a = static_cast<int32_t>(pcm_clamp(m[e][0], 0.9999999f) * 2147483648.0f);
          o.b = static_cast<int32_t>(pcm_clamp(m[e][1], 0.9999999f) * 2147483648.0f);
          o.c = static_cast<int32_t>(pcm_clamp(m[e][2], 0.9999999f) * 2147483648.0f);
          o.d = static_cast<int32_t>(pcm_clamp(m[e][3], 0.9999999f) * 2147483648.0f);
          *reinterpret_cast<I4 *>(dst + e * i_step) = o;
        }
      }
    }
  }
}
// Rows form for few planes (R = P*C = 4 or 8, either plane layout): a lane takes ONE i -- R four-byte loads, one per plane
// (consecutive lanes read consecutive floats of each plane; in the split layout two 128-byte runs per instruction), and
// R/4 sixteen-byte stores of whole frame groups (consecutive lanes write consecutive memory: 1 KiB per instruction at
// R = 4). The quad form's 16-byte loads are lane-contiguous too, but its stores land 64 (128) bytes apart per lane:
// config 4 (R = 4) 217 us; a variant with eight i per lane 250 us.
template <int FMT, int R>
MI_GLOBAL void interleave_rows_kernel(Geometry g, IoDesc io, const float *MI_RESTRICT planes, int sb0, int nb,
                                      int wgs_per_pair) {
  constexpr int kDepth = 32 / R;  // 32 loads in flight per lane
  const int jb = static_cast<int>(MI_BID_X) / wgs_per_pair;
  if (jb >= nb) {
    return;
  }
  const int wg = static_cast<int>(MI_BID_X) - jb * wgs_per_pair;
  const unsigned C = static_cast<unsigned>(io.channels), P = static_cast<unsigned>(g.P), Bc = static_cast<unsigned>(g.Bc);
  const int sb = sb0 + jb, s = sb / io.blocks, blk = sb - s * io.blocks;
  const float *src = planes + static_cast<long long>(jb) * C * g.P * g.Bp;
  char *out_blk = static_cast<char *>(io.out) + s * io.out_stream_stride + static_cast<long long>(blk) * g.B * C * 4;
  const float *pl[R];  // value e of a frame group = (phase e / C, channel e % C)
  MI_UNROLL
  for (int e = 0; e < R; ++e) {
    const unsigned pp = static_cast<unsigned>(e) / C, c = static_cast<unsigned>(e) - pp * C;
    pl[e] = src + static_cast<size_t>(c * P + pp) * static_cast<unsigned>(g.Bp);
  }
  const unsigned base = static_cast<unsigned>(wg) * MI_BDIM_X * kDepth + MI_TID_X;
  float v[kDepth][R];
  MI_UNROLL
  for (int d = 0; d < kDepth; ++d) {
    const unsigned i = base + d * MI_BDIM_X;
    if (i < Bc) {
      const unsigned at = io.split_planes ? ((i & 2u) ? (Bc >> 1) : 0u) + ((i >> 2) << 1) + (i & 1u) : i;
      MI_UNROLL
      for (int e = 0; e < R; ++e) {
        v[d][e] = pl[e][at];
      }
    }
  }
  MI_UNROLL
  for (int d = 0; d < kDepth; ++d) {
    const unsigned i = base + d * MI_BDIM_X;
    if (i < Bc) {
      char *dst = out_blk + static_cast<size_t>(i) * (4 * R);
      MI_UNROLL
      for (int e = 0; e < R; e += 4) {
        if constexpr (FMT == kF32) {
          *reinterpret_cast<f4 *>(dst + 4 * e) = f4{v[d][e], v[d][e + 1], v[d][e + 2], v[d][e + 3]};
        } else {
          struct alignas(16) I4 {
            int32_t a, b, c, d;
          };
          I4 o;
          o.a = static_cast<int32_t>(pcm_clamp(v[d][e], 0.9999999f) * 2147483648.0f);
          o.b = static_cast<int32_t>(pcm_clamp(v[d][e + 1], 0.9999999f) * 2147483648.0f);
          o.c = static_cast<int32_t>(pcm_clamp(v[d][e + 2], 0.9999999f) * 2147483648.0f);
          o.d = static_cast<int32_t>(pcm_clamp(v[d][e + 3], 0.9999999f) * 2147483648.0f);
          *reinterpret_cast<I4 *>(dst + 4 * e) = o;
        }
      }
    }
  }
}
// Tiled form for many planes (R = P*C >= 16 rows, plain plane layout): the quad form reads 16 bytes from each of R
// different planes per wave instruction -- R different cache lines, each revisited by seven later instructions; at
// R = 128 (config 3) it moved 2.4 TB/s. Here a workgroup takes a tile of TI consecutive i of ALL R rows of one (stream,
// block) pair: every wave load instruction reads 256-byte row segments (16 lanes x 16 bytes per row, whole lines), the
// tile crosses through LDS ([R][TI + 1] floats, rows in output order r = p*C + c), and every store instruction writes
// whole frames (R*4 contiguous bytes per i) as 16-byte lane-contiguous vectors. All of a thread's loads are in flight
// before its first LDS write.
#if defined(MIUPS_EXP_ILV_VGPR_CAP) && !defined(MIUPS_HOST_EMU)  // experiment (profiles/r03_j_coresident.txt)
#define MI_ILV_VGPR_CAP __attribute__((amdgpu_num_vgpr(MIUPS_EXP_ILV_VGPR_CAP)))
#else
#define MI_ILV_VGPR_CAP
#endif
template <int FMT, int TI, int EPT>
MI_GLOBAL MI_ILV_VGPR_CAP void interleave_tiled_kernel(Geometry g, IoDesc io, const float *MI_RESTRICT planes, int sb0, int nb,
                                       int tiles_per_pair) {
  MI_DYN_SHARED(float, tile);
  constexpr int LD = TI + 1, Q = TI / 4;  // row pitch, 16-byte words per row
  const int jb = static_cast<int>(MI_BID_X) / tiles_per_pair;
  if (jb >= nb) {
    return;
  }
  const int k = static_cast<int>(MI_BID_X) - jb * tiles_per_pair;
  // cooperative frames (device/frame_tile.h): the transform kernel's own workgroups have assembled the tiles below
  // FrameSync::next of this pair (same tile width); this pass takes the rest
  if (io.fsync != nullptr && static_cast<unsigned>(k) < io.fsync[jb].next) {
    return;
  }
  const int C = io.channels, P = g.P, R = P * C, Rq = R >> 2;
  const int i0 = k * TI;
  const int sb = sb0 + jb, s = sb / io.blocks, blk = sb - s * io.blocks;
  const float *src = planes + static_cast<long long>(jb) * C * g.P * g.Bp;
  char *out_blk = static_cast<char *>(io.out) + s * io.out_stream_stride + static_cast<long long>(blk) * g.B * C * 4;
  const int tid = MI_TID_X, nt = MI_BDIM_X;
  // EPT = R * Q / threads 16-byte words per thread (host: exact)
  f4 v[EPT];
  MI_UNROLL
  for (int j = 0; j < EPT; ++j) {
    const int x = tid + j * nt, r = x / Q, q4 = (x - r * Q) * 4;
    const int pp = r / C, c = r - pp * C;
    v[j] = (i0 + q4 < g.Bc) ? *reinterpret_cast<const f4 *>(src + (static_cast<long long>(c) * P + pp) * g.Bp + i0 + q4)
                            : f4{0.0f, 0.0f, 0.0f, 0.0f};  // Bc % 4 == 0: a word is inside or outside as a whole
  }
  MI_UNROLL
  for (int j = 0; j < EPT; ++j) {
    const int x = tid + j * nt, r = x / Q, q4 = (x - r * Q) * 4;
    float *row = tile + r * LD + q4;
    row[0] = v[j].x;
    row[1] = v[j].y;
    row[2] = v[j].z;
    row[3] = v[j].w;
  }
  MI_SYNC();
  MI_UNROLL
  for (int j = 0; j < EPT; ++j) {
    const int x = tid + j * nt, il = x / Rq, r0 = (x - il * Rq) * 4;  // lanes over the R/4 runs of a frame group first
    if (i0 + il < g.Bc) {
      const float *col = tile + r0 * LD + il;
      const float a = col[0], b = col[LD], c2 = col[2 * LD], d = col[3 * LD];
      char *dst = out_blk + (static_cast<long long>(i0 + il) * R + r0) * 4;
      if constexpr (FMT == kF32) {
        *reinterpret_cast<f4 *>(dst) = f4{a, b, c2, d};
      } else {
        struct alignas(16) I4 {
          int32_t a, b, c, d;
        };
        I4 o;
        o.a = static_cast<int32_t>(pcm_clamp(a, 0.9999999f) * 2147483648.0f);
        o.b = static_cast<int32_t>(pcm_clamp(b, 0.9999999f) * 2147483648.0f);
        o.c = static_cast<int32_t>(pcm_clamp(c2, 0.9999999f) * 2147483648.0f);
        o.d = static_cast<int32_t>(pcm_clamp(d, 0.9999999f) * 2147483648.0f);
        *reinterpret_cast<I4 *>(dst) = o;
      }
    }
  }
}
// General form: one output sample per thread, lanes in output order.
MI_GLOBAL void interleave_scalar_kernel(Geometry g, IoDesc io, const float *MI_RESTRICT planes, int sb0, int nb) {
  const long long gid = static_cast<long long>(MI_BID_X) * MI_BDIM_X + MI_TID_X;
  const int C = io.channels;
  const long long per_pair = static_cast<long long>(g.B) * C;
  if (gid >= per_pair * nb) {
    return;
  }
  const int jb = static_cast<int>(gid / per_pair);
  const long long rem = gid - jb * per_pair;
  const int m = static_cast<int>(rem / C), c = static_cast<int>(rem - static_cast<long long>(m) * C);
  const int i = m / g.P, p = m - i * g.P;
  const int sb = sb0 + jb, s = sb / io.blocks, blk = sb - s * io.blocks;
  const int at = io.split_planes ? ((i & 2) ? (g.Bc >> 1) : 0) + 2 * (i >> 2) + (i & 1) : i;
  const float val = planes[((static_cast<long long>(jb) * C + c) * g.P + p) * g.Bp + at];
  pcm_store(static_cast<char *>(io.out) + s * io.out_stream_stride, io.out_fmt,
            (static_cast<long long>(blk) * g.B + m) * C + c, val);
}

// ---- history carry: new_hist = last hist_frames frames of (hist ++ input) --
// (reference: overlap_.assign(timeBuffer.end() - overlap, ...), :571-572)
MI_GLOBAL void update_history_kernel(Geometry g, IoDesc io, void *MI_RESTRICT new_hist, long long total_in_frames) {
  // one 16-byte / 4-byte / 1-byte unit per thread, the widest the row length, the hist/input boundary, the strides and
  // the bases allow (the byte form alone took 28 us per call for 32 stereo streams of 40 000 frames)
  const long long gid = static_cast<long long>(MI_BID_X) * MI_BDIM_X + MI_TID_X;
  const long long frame_bytes = static_cast<long long>(io.channels) * pcm_bytes(io.in_fmt);
  const long long row_bytes = static_cast<long long>(g.hist_frames) * frame_bytes;
  // bytes of the new row that come from the old one (the rest comes from the end of the input)
  const long long from_hist = total_in_frames >= g.hist_frames ? 0 : (g.hist_frames - total_in_frames) * frame_bytes;
  const unsigned long long mix = static_cast<unsigned long long>(row_bytes) | static_cast<unsigned long long>(from_hist) |
                                 static_cast<unsigned long long>(total_in_frames * frame_bytes) |
                                 static_cast<unsigned long long>(io.in_stream_stride) |
                                 static_cast<unsigned long long>(io.hist_stream_stride) |
                                 reinterpret_cast<uintptr_t>(io.in) | reinterpret_cast<uintptr_t>(io.hist) |
                                 reinterpret_cast<uintptr_t>(new_hist);
  const int unit = (mix & 15) == 0 ? 16 : ((mix & 3) == 0 ? 4 : 1);
  const long long row_units = row_bytes / unit;
  if (gid >= row_units * io.streams) {
    return;
  }
  const long long s = gid / row_units, off = (gid - s * row_units) * unit;
  const unsigned char *src =
      off < from_hist ? static_cast<const unsigned char *>(io.hist) + s * io.hist_stream_stride + (row_bytes - from_hist) + off
                      : static_cast<const unsigned char *>(io.in) + s * io.in_stream_stride + total_in_frames * frame_bytes -
                            row_bytes + off;
  unsigned char *dst = static_cast<unsigned char *>(new_hist) + s * io.hist_stream_stride + off;
  if (unit == 16) {
    struct alignas(16) B16 {
      uint32_t w[4];
    };
    *reinterpret_cast<B16 *>(dst) = *reinterpret_cast<const B16 *>(src);
  } else if (unit == 4) {
    *reinterpret_cast<uint32_t *>(dst) = *reinterpret_cast<const uint32_t *>(src);
  } else {
    *dst = *src;
  }
}

}  // namespace miups
