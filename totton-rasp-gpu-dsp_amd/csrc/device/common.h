// Device-side common definitions for the MI355X (gfx950) upsampler kernels.
//
// The kernels are written against a tiny macro layer so that the SAME source
// can also be compiled by g++ into a thread-emulated debug build that lives
// under tests/ (tests/emu): that build exists only to bounds-check and
// cross-check kernel indexing on the CPU before a kernel is ever launched on a
// GPU box. It is never part of libmi_upsampler.so and no product entry point
// can reach it -- the product path is HIP only and fails loudly without a GPU.
#pragma once

#include <cstddef>
#include <cstdint>

#if defined(MIUPS_HOST_EMU)
#include <cmath>
namespace miups_emu {
struct Dim3 {
  unsigned x = 1, y = 1, z = 1;
};
extern thread_local Dim3 t_threadIdx;
extern thread_local Dim3 t_blockIdx;
extern Dim3 g_blockDim;
extern Dim3 g_gridDim;
void barrier();
void *dyn_shared();
float xchg_xor(float v, unsigned mask);  // value of thread (threadIdx.x ^ mask); called by every thread of the block
}  // namespace miups_emu
#define MI_DEVICE inline
#define MI_HD inline
#define MI_GLOBAL static
#define MI_SYNC() ::miups_emu::barrier()
#define MI_TID_X (::miups_emu::t_threadIdx.x)
#define MI_BID_X (::miups_emu::t_blockIdx.x)
#define MI_BID_Y (::miups_emu::t_blockIdx.y)
#define MI_BDIM_X (::miups_emu::g_blockDim.x)
#define MI_GDIM_X (::miups_emu::g_gridDim.x)
#define MI_DYN_SHARED(type, name) type *name = static_cast<type *>(::miups_emu::dyn_shared())
#define MI_UNROLL
#define MI_LAUNCH_BOUNDS(t, w)
#define MI_RESTRICT __restrict__
#else
#include <hip/hip_runtime.h>
#define MI_DEVICE __device__ __forceinline__
#define MI_HD __host__ __device__ inline
#define MI_GLOBAL __global__
#define MI_SYNC() __syncthreads()
#define MI_TID_X (threadIdx.x)
#define MI_BID_X (blockIdx.x)
#define MI_BID_Y (blockIdx.y)
#define MI_BDIM_X (blockDim.x)
#define MI_GDIM_X (gridDim.x)
#define MI_DYN_SHARED(type, name)                                  \
  extern __shared__ __attribute__((aligned(16))) char mi_dyn_smem_[]; \
  type *name = reinterpret_cast<type *>(mi_dyn_smem_)
#define MI_UNROLL _Pragma("unroll")
#define MI_LAUNCH_BOUNDS(t, w) __launch_bounds__(t, w)
#define MI_RESTRICT __restrict__
#endif

#if defined(MIUPS_HOST_EMU)
#define MI_SCHED_FENCE()
#define MI_OPAQUE_VGPR(x)
#else
// nothing moves across it in the instruction scheduler (pins the order of load groups, store groups and butterflies)
#define MI_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// Makes `x` look freshly defined: address arithmetic derived from it cannot be
// hoisted out of the phase / channel loops (hipcc otherwise precomputes every
// LDS / output offset of all passes and keeps >100 registers live -> spills).
#define MI_OPAQUE_VGPR(x) asm volatile("" : "+v"(x))
#endif

namespace miups {

// complex float, 8 bytes, layout-compatible with float2 / std::complex<float>
struct alignas(8) cf {
  float x, y;
};

MI_DEVICE cf mk(float x, float y) {
  cf r;
  r.x = x;
  r.y = y;
  return r;
}

// ---- packed arithmetic ---------------------------------------------------------------------------------------
// All butterfly arithmetic is written on `v2` = one complex value {re, im}, in a form that maps 1:1 onto gfx950's
// packed fp32 instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 on an aligned register pair; swizzles and
// whole-vector negation ride on op_sel / neg modifiers; the one thing hipcc does not fold, a swizzle with ONE negated
// half = the second step of a complex product, is a single inline-asm v_pk_fma_f32 -- hipcc pads the packed-result
// forwarding hazard around asm statements itself).
// Measured (profiles/r02_b_*): a packed instruction takes 0.72x the time of the two scalar ones it replaces when two
// waves share a SIMD, the radix-16 passes drop from 523-629 to 289-395 VALU instructions, and in isolation the middle
// passes get 7-12 % and the spectral stage 2x faster -- but the passes are then LDS-bound (2.05k cycles of LDS pipe
// per pass), the aligned pairs cost 14 more spilled registers inside the phase loop, and the whole kernel is 5-6 %
// SLOWER (config 2: 190 -> 179, config 5: 168 -> 163 Gsamples/s, same box, interleaved runs). The product therefore
// compiles this source to scalar instructions; -DMIUPS_EXP_PACKED_MATH (experiment switch) selects the packed form.
// The emulation build (g++, tests/emu) uses the scalar form too.
#if defined(MIUPS_HOST_EMU) || !defined(MIUPS_EXP_PACKED_MATH)
struct v2 {
  float x, y;
};
MI_DEVICE v2 operator+(v2 a, v2 b) { return v2{a.x + b.x, a.y + b.y}; }
MI_DEVICE v2 operator-(v2 a, v2 b) { return v2{a.x - b.x, a.y - b.y}; }
MI_DEVICE v2 operator*(v2 a, v2 b) { return v2{a.x * b.x, a.y * b.y}; }
MI_DEVICE v2 operator*(v2 a, float s) { return v2{a.x * s, a.y * s}; }
MI_DEVICE v2 operator-(v2 a) { return v2{-a.x, -a.y}; }
MI_DEVICE v2 v2swap(v2 a) { return v2{a.y, a.x}; }
MI_DEVICE v2 v2xx(v2 a) { return v2{a.x, a.x}; }
MI_DEVICE v2 v2yy(v2 a) { return v2{a.y, a.y}; }
MI_DEVICE v2 v2fma(v2 a, v2 b, v2 c) { return v2{a.x * b.x + c.x, a.y * b.y + c.y}; }
// (-a.y*w.y + t.x, a.x*w.y + t.y) and (a.y*w.y + t.x, -a.x*w.y + t.y)
MI_DEVICE v2 v2cross(v2 a, v2 w, v2 t) { return v2{-a.y * w.y + t.x, a.x * w.y + t.y}; }
MI_DEVICE v2 v2crossc(v2 a, v2 w, v2 t) { return v2{a.y * w.y + t.x, -a.x * w.y + t.y}; }
#else
typedef float v2 __attribute__((ext_vector_type(2)));
MI_DEVICE v2 v2swap(v2 a) { return a.yx; }
MI_DEVICE v2 v2xx(v2 a) { return a.xx; }
MI_DEVICE v2 v2yy(v2 a) { return a.yy; }
MI_DEVICE v2 v2fma(v2 a, v2 b, v2 c) { return __builtin_elementwise_fma(a, b, c); }
MI_DEVICE v2 v2cross(v2 a, v2 w, v2 t) {
  v2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
  return r;
}
MI_DEVICE v2 v2crossc(v2 a, v2 w, v2 t) {
  v2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
  return r;
}
#endif
MI_DEVICE v2 V(cf a) { return v2{a.x, a.y}; }
MI_DEVICE cf C(v2 a) { return mk(a.x, a.y); }
MI_DEVICE v2 v2mk(float x, float y) { return v2{x, y}; }
// a * w  and  a * conj(w): two packed instructions each
MI_DEVICE v2 vmul(v2 a, v2 w) { return v2cross(a, w, a * v2xx(w)); }
MI_DEVICE v2 vmulc(v2 a, v2 w) { return v2crossc(a, w, a * v2xx(w)); }
// j*a and -j*a
MI_DEVICE v2 vmulj(v2 a) { return v2swap(a) * v2mk(-1.0f, 1.0f); }
MI_DEVICE v2 vmulnj(v2 a) { return v2swap(a) * v2mk(1.0f, -1.0f); }
MI_DEVICE v2 vconj(v2 a) { return a * v2mk(1.0f, -1.0f); }

// The value the partner lane (lane ^ 32 of the same wave) holds in `a`. Two v_permlane32_swap_b32 move both dwords
// both ways at once: `swap x, y` exchanges x's upper half-wave with y's lower one, so after `swap x, y; swap y, x`
// y holds the partner's x and x the partner's y -- no copies, no selects, no LDS. Spelled in asm because the chained
// form of __builtin_amdgcn_permlane32_swap miscompiles on ROCm 7.2 (both results read from one register: checked on the
// GPU with scripts/ubench/permlane_check.hip, which also validates this sequence); the s_nop are the VALU-write ->
// v_permlane read wait states.
MI_DEVICE cf xchg32(cf a) {
#if defined(MIUPS_HOST_EMU)
  return mk(::miups_emu::xchg_xor(a.x, 32u), ::miups_emu::xchg_xor(a.y, 32u));
#else
  float x = a.x, y = a.y;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_permlane32_swap_b32 %1, %0\n\ts_nop 1"
               : "+v"(x), "+v"(y));
  return mk(y, x);
#endif
}

MI_DEVICE cf cadd(cf a, cf b) { return C(V(a) + V(b)); }
MI_DEVICE cf csub(cf a, cf b) { return C(V(a) - V(b)); }
MI_DEVICE cf cmul(cf a, cf b) { return C(vmul(V(a), V(b))); }
// a * conj(b)
MI_DEVICE cf cmulc(cf a, cf b) { return C(vmulc(V(a), V(b))); }
MI_DEVICE cf cconj(cf a) { return mk(a.x, -a.y); }
MI_DEVICE cf cscale(cf a, float s) { return C(V(a) * s); }
MI_DEVICE cf cneg(cf a) { return mk(-a.x, -a.y); }
// j*a and -j*a
MI_DEVICE cf cmulj(cf a) { return mk(-a.y, a.x); }
MI_DEVICE cf cmulnj(cf a) { return mk(a.y, -a.x); }

struct alignas(16) f4 {
  float x, y, z, w;
};

// LDS word (8-byte complex) index swizzle of the fused kernel. Every pass
// touches, per wave instruction, either 32 consecutive words or 16-word blocks
// whose block index varies across lanes; the XOR terms spread both patterns over
// all banks for ds_read_b64 (64 banks, 32-lane groups) and ds_write_b64
// (32 banks, 16-lane groups). Verified by tests/test_lds_layout.py.
MI_HD int lds_swz(int i) {
  return i ^ ((i >> 4) & 15) ^ ((i >> 5) & 8) ^ ((((i >> 8) ^ (i >> 9)) & 1) << 4);
}

// Pass plans of the fused kernel's in-LDS transform (kernel_fused.h FusedCfg): the classic plan (2^(log2k mod 4), 16, ..,
// 16) is the product's; the radix-32 plan (K/512, 32, 16) exists for the wide form at K = 8192 and 16384 as a measured
// experiment (profiles/r03_b_radix32.txt: one LDS round trip and one barrier fewer per transform, but +14 % VALU work and
// ~70 spilled registers -> 0.72x). The host orders the spectrum tables by the plan's digit reversal.
MI_HD constexpr bool fused_plan_r32_exists(int log2k, int w) { return w == 2 && log2k >= 13 && log2k <= 14; }

// Tables read by the fused kernel, laid out per thread (the kernel's thread ->
// frequency-bin assignment is fixed, so the host stores everything in the
// order lanes consume it: every table load is lane-contiguous).
//   T = K/32 threads; J = K/16 sixteen-bin sets S_k = {k + t*J}.
//   thread tau >= 1 owns S_a and S_{J-a}, a = set_of_thread[tau]; thread 0 owns
//   the self-mirrored S_0 and S_{J/2}.
struct FusedTables {
  const cf *tw;        // Stockham/Cooley-Tukey pass twiddles, see tw_offset()
  const cf *WmT;       // [T]            W_M^a                     (a = low index of the thread's first set)
  const int *blockB;   // [T]            LDS block that holds the thread's second set after the forward FFT
  const f4 *GT;        // [P][16][T]     {Gs[k], Gc[k]}, k = a + t*J  (pair t of thread tau); column 0 = thread 0's slots:
                       //                slot t <= 8: k = t*J, slot t >= 9: k = J/2 + (t-9)*J
  const f4 *G0;        // [P][17]        thread 0's 17 pairs: k = t*J (t = 0..8) then k = J/2 + t*J (t = 0..7); the kernel
                       //                reads entry 16 (its extra pair), the rest documents column 0 of GT
  cf Wb;               // W_M^(J/2)
  // split form (fused_split_kernel) instead: GT [P][2][16][T], G0 [P][2][17] (self lanes), and
  const cf *selfW;     // [17]           W_M^k of self lane l: k = l*J (l <= 8), J/2 + (l-9)*J
  cf Wself;            // plain form: W_M^(J/2) / W_32^9 = thread 0's twiddle base for its slots 9..15
};

// PCM sample formats at the batched boundary. Values mirror include/mi_upsampler.h.
enum PcmFormat : int { kF32 = 0, kS16 = 1, kS24_3LE = 2, kS32 = 3 };

MI_HD int pcm_bytes(int fmt) {
  return fmt == kS16 ? 2 : (fmt == kS24_3LE ? 3 : 4);
}

// Geometry shared by every kernel of one loaded filter. All sizes in samples
// of the compact (per-phase) domain unless stated.
//   N = fft_size, B = block_size, O = N - B, Lf = upsample_factor
//   P = number of output phases handled in the frequency domain (P = Lf when
//       N % Lf == 0, else 1), S = Lf / P = zero-stuff stride left in the
//       time domain (1 whenever P = Lf)
//   M = N / P real samples per channel-block, K = M / 2 complex points
//   Oc = O / P history samples, Bc = B / P new compact samples per block
struct Geometry {
  int log2k;      // K = 1 << log2k
  int K;          // complex FFT length
  int M;          // 2K
  int P;          // phases
  int S;          // residual stuffing stride
  int Oc;         // compact history length
  int Bc;         // compact new samples per block ( = n_in * S )
  int n_in;       // input frames per block (B / Lf)
  int B;          // output frames per block
  int hist_frames;  // frames of input history kept per stream = ceil(Oc / S)
  // pitch of a staging plane in floats: Bc rounded up to a whole number of 128-byte lines, so that every plane (and
  // every row segment the interleave kernels read) starts on a line boundary (8x / 16x at N = 131072: Bc = 6384 / 3192
  // floats = 199.5 / 99.75 lines). A channel's P planes take P * Bp floats.
  int Bp;
};

// Cooperative frame assembly (device/frame_tile.h): one record per (stream, block) pair of a launch, zeroed before it. A
// 128-byte line of its own: the counters of two pairs are never updated through two different L2s in one line.
struct alignas(128) FrameSync {
  unsigned done;      // workgroups of this pair whose planes are stored (complete at IoDesc::groups)
  unsigned xcc_mask;  // OR of (1 << XCC_ID) of those workgroups: the pair is eligible only where this is ONE bit
  unsigned next;      // first tile nobody has claimed; the frame pass behind the kernel assembles tiles >= next
  unsigned pad[29];
};

// Where samples live for one batched call. Frames are interleaved:
// sample(stream s, frame f, channel c) at
//   base + s * stream_stride_bytes + (f * channels + c) * bytes(fmt)
struct IoDesc {
  const void *in;        // new input frames of this call (device)
  const void *hist;      // last hist_frames frames before this call (device)
  void *out;             // output frames (device)
  long long in_stream_stride;    // bytes
  long long hist_stream_stride;  // bytes
  long long out_stream_stride;   // bytes
  int channels;          // channels per stream
  int streams;
  int in_fmt;
  int out_fmt;
  int blocks;            // blocks processed by this call
  // fused path only: one workgroup handles `cg` consecutive channels of one
  // (block, stream) and stages their phase-planar fp32 results in `scratch`
  // ([workgroup][channel-in-group][phase][kept sample]) before writing whole
  // interleaved frames. groups = channels / cg; item0 = first work item of this
  // launch (launches are chunked so that scratch stays bounded).
  float *scratch;
  int cg;
  int groups;
  int item0;
  int out_vec_ok;        // 1 when out base/strides allow 16-byte aligned vector stores
  // fused path, more than two channels: `in` is not the caller's interleaved PCM
  // but the engine's planar copy made by planarize_kernel -- one fp32 timeline
  // (history ++ new frames) per channel, in_plane_stride bytes apart, in_fmt = f32.
  int in_planar;         // 1: plain timeline; 2: split-planar (split form, block windows start at multiples of 4)
  long long in_plane_stride;
  // fused path with groups narrower than a frame (cg < channels): the kernel stops after
  // the staging planes (ext_epilogue = 1) and interleave_*_kernel turns each chunk of
  // work items (whole (stream, block) pairs: items run group-fastest) into PCM frames.
  int ext_epilogue;
  // split form (fused_split_kernel): a staging plane holds the (y[4m], y[4m+1]) pairs in its
  // first half and the (y[4m+2], y[4m+3]) pairs in its second half (m >= Oc/4, Oc % 4 == 0)
  int split_planes;
  // split form: where a workgroup parks the first-pass inputs of a phase's SECOND half transform while the first one
  // runs ([workgroup][16 slots][T] + [17] self lanes, 16 bytes each = kSplitParkWords f4 per workgroup); null = the
  // second half recomputes its spectral products (the round-2 form)
  f4 *park;
  // fused_parts_kernel: workgroups per work item (a divisor of P, >= 2); 0 elsewhere
  int phase_parts;
  // cooperative frames (ext_epilogue, plain plane layout, device/frame_tile.h): null = off. [launch-local pair]
  FrameSync *fsync;
  int ftile_ti;    // tile width in kept samples: 64 / 32 / 16 (the frame pass behind the kernel uses the same)
  int ftile_ept;   // sixteen-byte words per thread and tile inside the transform kernel: R * (ti / 4) / T, 1..8
  int ftiles;      // tiles per pair
  int ftile_cap;   // tiles one workgroup may assemble (about twice its share)
};
// smallest transform length (log2) that has a fused_parts_kernel
constexpr int kPartsMinLog2K = 10;
// f4 words per workgroup of IoDesc::park for a split kernel of T threads
MI_HD constexpr int split_park_words(int threads) { return 16 * threads + 32; }

}  // namespace miups
