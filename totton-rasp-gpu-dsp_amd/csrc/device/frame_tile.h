// One tile of frame assembly -- TI consecutive kept samples i of ALL R = P*C staging planes of one (stream, block) pair
// crossed through LDS into whole interleaved frames -- as a device function, and the cross-workgroup bookkeeping that
// lets the workgroups of the TRANSFORM kernel assemble frames themselves ("cooperative frames").
//
// Why: with frames wider than a workgroup's channel group the planes leave the transform kernel and a frame pass
// (interleave_tiled_kernel) turns them into PCM frames. That pass runs at the copy ceiling (5.4-5.7 TB/s) -- and it runs
// ALONE: 0.26 of 0.52 ms at BASELINE configs[2], 0.29 of 0.93 ms at configs[4], with every VALU idle, after a transform
// kernel during which the fabric is nearly idle. Here a transform workgroup that has stored its planes reports that
// (FrameSync::done) and then assembles a few tiles of pairs whose planes are all complete, while the other workgroups of
// the launch keep computing: the frame traffic moves under the transform's arithmetic, the plane reads hit data written
// microseconds ago, and what is left at the end of the launch (the last pairs) goes to the frame pass as before.
//
// Rules that make this safe without any workgroup ever WAITING for another (nothing spins, nothing can hang):
//  * a workgroup only ever looks for finished work; if there is none it exits. The frame pass launched behind the
//    kernel assembles every tile nobody claimed (FrameSync::next is the first unclaimed tile of a pair).
//  * visibility is only relied on INSIDE ONE XCD: the L2 caches of the eight XCDs are not coherent with each other
//    inside a kernel, so a pair is eligible only on the XCD whose id (hardware register XCC_ID) is the ONLY one its
//    producers recorded; counters of a pair that straddles two XCDs never reach the complete count in either L2 and the
//    pair falls to the frame pass. Producer: plane stores, s_waitcnt vmcnt(0) (write acknowledged by the L2), barrier,
//    then one atomic increment in the L2. Consumer: atomic loads of the counters (served by the L2), then plain loads of
//    plane lines this CU cannot hold a stale copy of (they are written once per launch, read only afterwards).
#pragma once

#include "common.h"
#include "pcm.h"

namespace miups {

#if defined(MIUPS_HOST_EMU)
// the emulation runs the workgroups of a launch one after the other
MI_DEVICE unsigned mi_atomic_add(unsigned *p, unsigned v) {
  const unsigned old = *p;
  *p = old + v;
  return old;
}
MI_DEVICE unsigned mi_atomic_or(unsigned *p, unsigned v) {
  const unsigned old = *p;
  *p = old | v;
  return old;
}
MI_DEVICE unsigned mi_atomic_load(const unsigned *p) { return *p; }
MI_DEVICE unsigned mi_xcc_id() { return 0u; }
MI_DEVICE void mi_stores_done() {}
#else
MI_DEVICE unsigned mi_atomic_add(unsigned *p, unsigned v) {
  return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
MI_DEVICE unsigned mi_atomic_or(unsigned *p, unsigned v) {
  return __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
MI_DEVICE unsigned mi_atomic_load(const unsigned *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// XCC_ID: which of the chip's XCDs (each with its own L2) this wave runs on. s_getreg_b32 hwreg(HW_REG_XCC_ID = 20), bits 3:0
MI_DEVICE unsigned mi_xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u; }
// every vector-memory operation this wave has issued has completed (stores: acknowledged by the L2)
MI_DEVICE void mi_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif

// NB tiles at once: tiles k0 .. k0 + ntile - 1 (ntile <= NB, workgroup-uniform) of one pair. Every plane word of ALL of
// them is requested before the first one is used -- NB * EPT sixteen-byte loads per thread in flight (128 registers) --
// because a workgroup that assembles one tile at a time pays the plane round trip (~2 us) per tile: 16-32 KB per 3 us per
// workgroup, an order of magnitude under what a CU can move, while it holds the LDS and registers a transform workgroup
// would use (first cut of cooperative frames: 0.77-0.79x at configs[2] / configs[4], profiles/r04_b_coop_frames.txt).
// Then tile by tile: registers -> LDS tile ([R][TI + 1] floats, rows in output order r = p*C + c) -> whole frames as
// 16-byte lane-contiguous stores. src = the pair's planes ([channel][phase][Bp] floats), out_blk = the pair's first output
// frame. TI in {16, 32, 64}; EPT = R * (TI / 4) / nt sixteen-byte words per thread and tile.
template <int FMT, int EPT, int NB>
MI_DEVICE void frame_tiles(const Geometry &g, int C, const float *MI_RESTRICT src, char *out_blk, int k0, int ntile, int TI,
                           float *tile, int tid, int nt) {
  const int P = g.P, R = P * C, Rq = R >> 2, LD = TI + 1;
  const int lq = TI == 64 ? 4 : (TI == 32 ? 3 : 2);  // log2(TI / 4): sixteen-byte words per row
  f4 v[NB][EPT];
  MI_UNROLL
  for (int b = 0; b < NB; ++b) {
    if (b < ntile) {
      const int i0 = (k0 + b) * TI;
      MI_UNROLL
      for (int j = 0; j < EPT; ++j) {
        const int x = tid + j * nt, r = x >> lq, q4 = (x - (r << lq)) * 4;
        const int pp = r / C, c = r - pp * C;
        v[b][j] = (i0 + q4 < g.Bc) ? *reinterpret_cast<const f4 *>(src + (static_cast<long long>(c) * P + pp) * g.Bp + i0 + q4)
                                   : f4{0.0f, 0.0f, 0.0f, 0.0f};  // Bc % 4 == 0: a word is inside or outside as a whole
      }
    }
  }
  MI_SCHED_FENCE();  // every load above is issued before the first LDS write below
  MI_UNROLL
  for (int b = 0; b < NB; ++b) {
    if (b < ntile) {
      const int i0 = (k0 + b) * TI;
      MI_UNROLL
      for (int j = 0; j < EPT; ++j) {
        const int x = tid + j * nt, r = x >> lq, q4 = (x - (r << lq)) * 4;
        float *row = tile + r * LD + q4;
        row[0] = v[b][j].x;
        row[1] = v[b][j].y;
        row[2] = v[b][j].z;
        row[3] = v[b][j].w;
      }
      MI_SYNC();
      MI_UNROLL
      for (int j = 0; j < EPT; ++j) {
        const int x = tid + j * nt, il = x / Rq, r0 = (x - il * Rq) * 4;  // lanes over the R/4 runs of a frame group first
        if (i0 + il < g.Bc) {
          const float *col = tile + r0 * LD + il;
          const float a = col[0], bb = col[LD], c2 = col[2 * LD], d = col[3 * LD];
          char *dst = out_blk + (static_cast<long long>(i0 + il) * R + r0) * 4;
          if constexpr (FMT == kF32) {
            *reinterpret_cast<f4 *>(dst) = f4{a, bb, c2, d};
          } else {
            struct alignas(16) I4 {
              int32_t a, b, c, d;
            };
            I4 o;
            o.a = static_cast<int32_t>(pcm_clamp(a, 0.9999999f) * 2147483648.0f);
            o.b = static_cast<int32_t>(pcm_clamp(bb, 0.9999999f) * 2147483648.0f);
            o.c = static_cast<int32_t>(pcm_clamp(c2, 0.9999999f) * 2147483648.0f);
            o.d = static_cast<int32_t>(pcm_clamp(d, 0.9999999f) * 2147483648.0f);
            *reinterpret_cast<I4 *>(dst) = o;
          }
        }
      }
      MI_SYNC();  // the tile's LDS words are free again
    }
  }
}

// After a workgroup of the transform kernel has issued the plane stores of its work item `unit` (launch-local index;
// items run group-fastest, so unit / io.groups is the launch-local pair): publish, then assemble up to io.ftile_cap tiles
// of complete pairs -- its own first, then up to three older ones (items start in order, so older pairs finish first).
// lds: at least 64 + R * (TI + 1) * 4 bytes, free at this point. Every thread of the workgroup calls this.
template <int NT, int EPT>
MI_DEVICE void coop_frames_ept(const Geometry &g, const IoDesc &io, int unit, float *lds, int tid) {
  constexpr int NB = 32 / EPT > 8 ? 8 : 32 / EPT;  // tiles per claim: NB * EPT sixteen-byte words per thread in flight
  FrameSync *fs = io.fsync;
  const int pair = unit / io.groups;
  const unsigned me = mi_xcc_id();
  mi_stores_done();
  MI_SYNC();  // ... of every wave of this workgroup
  unsigned *bcast = reinterpret_cast<unsigned *>(lds);
  float *tile = lds + 16;
  if (tid == 0) {
    (void)mi_atomic_or(&fs[pair].xcc_mask, 1u << me);  // returns: complete before the count below is issued
    (void)mi_atomic_add(&fs[pair].done, 1u);
  }
  const int C = io.channels, tiles = io.ftiles;
  const int sb0 = io.item0 / io.groups;  // first (stream, block) pair of this launch
  int budget = io.ftile_cap;
  for (int d = 0; d < 4 && budget > 0; ++d) {
    const int pr = pair - d;
    if (pr < 0) {
      break;
    }
    for (;;) {
      if (tid == 0) {
        unsigned claim = 0xffffffffu;
        if (mi_atomic_load(&fs[pr].done) == static_cast<unsigned>(io.groups) && mi_atomic_load(&fs[pr].xcc_mask) == (1u << me) &&
            mi_atomic_load(&fs[pr].next) < static_cast<unsigned>(tiles)) {
          const unsigned n = mi_atomic_add(&fs[pr].next, static_cast<unsigned>(NB));
          if (n < static_cast<unsigned>(tiles)) {
            claim = n;
          }
        }
        bcast[0] = claim;
      }
      MI_SYNC();
      const unsigned claim = bcast[0];
      MI_SYNC();  // everyone has read it before thread 0 may write the next one
      if (claim == 0xffffffffu) {
        break;
      }
      const int sb = sb0 + pr, s = sb / io.blocks, blk = sb - s * io.blocks;
      const float *src = io.scratch + static_cast<long long>(pr) * C * g.P * g.Bp;
      char *out_blk = static_cast<char *>(io.out) + s * io.out_stream_stride + static_cast<long long>(blk) * g.B * C * 4;
      const int k0 = static_cast<int>(claim), ntile = tiles - k0 < NB ? tiles - k0 : NB;
      if (io.out_fmt == kF32) {
        frame_tiles<kF32, EPT, NB>(g, C, src, out_blk, k0, ntile, io.ftile_ti, tile, tid, NT);
      } else {
        frame_tiles<kS32, EPT, NB>(g, C, src, out_blk, k0, ntile, io.ftile_ti, tile, tid, NT);
      }
      budget -= NB;
      if (budget <= 0) {
        break;
      }
    }
  }
}
template <int NT>
MI_DEVICE void coop_frames(const Geometry &g, const IoDesc &io, int unit, float *lds, int tid) {
  switch (io.ftile_ept) {  // workgroup-uniform; the host only offers these four
    case 1: coop_frames_ept<NT, 1>(g, io, unit, lds, tid); break;
    case 2: coop_frames_ept<NT, 2>(g, io, unit, lds, tid); break;
    case 4: coop_frames_ept<NT, 4>(g, io, unit, lds, tid); break;
    default: coop_frames_ept<NT, 8>(g, io, unit, lds, tid); break;
  }
}

}  // namespace miups
