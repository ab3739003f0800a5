/*
 * mi_upsampler.h -- C ABI of the MI355X-native overlap-save FIR upsampler.
 *
 * This is the drop-in boundary for ONE path of michihitoTakami/totton-rasp-gpu-dsp:
 * the per-channel streaming upsampler that the reference implements in
 *   include/vulkan/vulkan_streaming_upsampler.h:20-49   (class VulkanStreamingUpsampler)
 *   src/vulkan/vulkan_streaming_upsampler.cpp:483-600   (LoadFilter / ProcessBlock / Reset)
 * and calls from src/alsa/alsa_streamer_main.cpp:239-250,321,543.
 *
 * Two levels are exported:
 *   (1) mi_ups_*    one handle == one reference VulkanStreamingUpsampler
 *                   instance (one channel, host float buffers, blocking).
 *                   Entry points map 1:1 onto the reference's public methods.
 *   (2) mi_filter_* / mi_engine_*
 *                   the batched form the reference's caller loop collapses
 *                   into on a GPU: all channels of all streams and many
 *                   consecutive blocks per call, interleaved PCM in HBM in and
 *                   out (replaces the per-channel loops and the PCM convert /
 *                   (de)interleave around them, alsa_streamer_main.cpp:316-329,
 *                   510-553 and alsa_common.cpp:42-127).
 *
 * Conventions: plain pointers and sizes, no exceptions cross the boundary,
 * 0 == success unless stated, messages are copied into caller buffers.
 * The library is HIP-only: every entry point that needs the GPU fails with
 * MI_ERR_DEVICE (and a message) when no MI355X-class device is usable; there is
 * no CPU fallback (the reference's silent fallback, :745-750, is deliberately
 * not reproduced).
 */
#ifndef MI_UPSAMPLER_H
#define MI_UPSAMPLER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_UPS_ABI_VERSION 1

enum {
  MI_OK = 0,
  MI_ERR_ARG = 1,     /* null/invalid argument or handle state */
  MI_ERR_FILTER = 2,  /* LoadFilter failed; message = the reference's string */
  MI_ERR_DEVICE = 3,  /* HIP error / no device */
  MI_ERR_SIZE = 4     /* count or capacity does not match the geometry */
};

/* PCM formats at the batched boundary (alsa_common.cpp:12-27 + raw float) */
enum { MI_PCM_F32 = 0, MI_PCM_S16 = 1, MI_PCM_S24_3LE = 2, MI_PCM_S32 = 3 };

/* mi_filter_load flags */
enum {
  MI_LOAD_DEFAULT = 0,
  /* multiply by the spectrum the reference itself multiplies by (its fp32
   * recurrence-twiddle FFT of the taps, vulkan_streaming_upsampler.cpp:734-739)
   * instead of the exact spectrum; host-side, load-time only */
  MI_LOAD_REF_COMPAT_SPECTRUM = 1
};

/* FilterConfig, include/vulkan/vulkan_streaming_upsampler.h:12-18 */
typedef struct mi_ups_config {
  size_t taps;
  size_t fft_size;
  size_t block_size;
  size_t upsample_factor;
  char coefficients_path[1024];
} mi_ups_config;

int mi_ups_abi_version(void);
/* number of usable HIP devices (0 when there is none; never fails) */
int mi_ups_device_count(void);
/* last error text of the calling thread ("" when none) */
const char *mi_ups_last_error(void);

/* ------------------------------------------------------------------ (1) --
 * Reference-shaped single-channel handle.
 *   mi_ups_create        VulkanStreamingUpsampler()                    .h:22
 *   mi_ups_clone         copy constructor / operator= (deep copy of the
 *                        overlap state; device tables are shared)      .cpp:450-479
 *   mi_ups_load_filter   LoadFilter(jsonPath, &errorMessage)           .cpp:483-498
 *   mi_ups_get_config    GetConfig()                                   .cpp:602-604
 *   mi_ups_process_block ProcessBlock(input, count)                    .cpp:500-596
 *   mi_ups_reset         Reset()                                       .cpp:598-600
 *   mi_ups_destroy       ~VulkanStreamingUpsampler()
 */
typedef struct mi_ups mi_ups;

mi_ups *mi_ups_create(int device);
mi_ups *mi_ups_clone(const mi_ups *other);
void mi_ups_destroy(mi_ups *h);
/* returns MI_OK or MI_ERR_*; on MI_ERR_FILTER `err` holds the reference's exact
 * message. `flags`: MI_LOAD_* */
int mi_ups_load_filter(mi_ups *h, const char *json_path, int flags, char *err, size_t errcap);
int mi_ups_get_config(const mi_ups *h, mi_ups_config *out);
/* Returns the number of samples written (== block_size) or 0 in every case in
 * which the reference returns an empty vector: not initialised, null input,
 * count == 0, count != block_size / upsample_factor, device failure
 * (mi_ups_last_error() tells which). out must hold block_size floats. */
long mi_ups_process_block(mi_ups *h, const float *input, size_t count, float *out, size_t outcap);
int mi_ups_reset(mi_ups *h);
/* optional EQ (reference: eq_parser.cpp / eq_to_fir.cpp, no call site there):
 * fold the APO profile's complex response, sampled at k*fs_out/fft_size, into
 * the filter spectrum. Empty/NULL text removes the EQ. */
int mi_ups_set_eq(mi_ups *h, const char *apo_text, double fs_out);

/* ------------------------------------------------------------------ (2) --
 * Shared device-resident filter + batched multi-channel engine.
 */
typedef struct mi_filter mi_filter;
typedef struct mi_engine mi_engine;

int mi_filter_load(int device, const char *json_path, int flags, mi_filter **out, char *err, size_t errcap);
/* same, from memory (taps are float32; geometry rules as for the sidecar) */
int mi_filter_from_taps(int device, const float *taps, size_t n_taps, size_t fft_size, size_t block_size,
                        size_t upsample_factor, int flags, mi_filter **out, char *err, size_t errcap);
int mi_filter_get_config(const mi_filter *f, mi_ups_config *out);
int mi_filter_set_eq(mi_filter *f, const char *apo_text, double fs_out);
/* Evaluate the EQ cascade on the device (fp64): num_bins (re,im) pairs at
 * f_i = i*fs_out/full_fft -- computeEqResponseForFft, eq_to_fir.cpp:145-151 */
int mi_eq_response_device(int device, const char *apo_text, size_t num_bins, size_t full_fft, double fs_out,
                          double *out_reim);
void mi_filter_release(mi_filter *f);

/* Engine: `streams` independent streams of `channels` interleaved channels,
 * all through filter `f`. Keeps block_size/upsample_factor-independent state:
 * the last (taps-1)/L input frames of every stream, in the input format. */
int mi_engine_create(mi_filter *f, int streams, int channels, int in_fmt, int out_fmt, mi_engine **out);
void mi_engine_destroy(mi_engine *e);
int mi_engine_reset(mi_engine *e);
/* frames per block: input = block_size / upsample_factor, output = block_size */
size_t mi_engine_in_frames_per_block(const mi_engine *e);
size_t mi_engine_out_frames_per_block(const mi_engine *e);
/* which kernel family the geometry selected: "fused" or "staged" */
const char *mi_engine_path(const mi_engine *e);

/* Process `blocks` consecutive blocks of every stream. d_in / d_out are DEVICE
 * pointers: stream s starts at base + s*stride bytes and holds interleaved
 * frames [frame][channel]. `hip_stream` is a hipStream_t (NULL = default
 * stream); the call only enqueues work (see "Stream contract" below). */
int mi_engine_process_device(mi_engine *e, const void *d_in, size_t in_stream_stride_bytes, void *d_out,
                             size_t out_stream_stride_bytes, size_t blocks, void *hip_stream);
/* Same with HOST buffers: copies in, runs, copies out, synchronises. The call is cut into sub-batches whose H2D
 * copy, kernels and D2H copy overlap on three streams (double-buffered device staging). Buffers from mi_host_alloc
 * (pinned) are moved by DMA directly; pageable buffers work, at the runtime's staging speed. */
int mi_engine_process_host(mi_engine *e, const void *h_in, size_t in_stream_stride_bytes, void *h_out,
                           size_t out_stream_stride_bytes, size_t blocks);
/* pinned host memory for the buffers of mi_engine_process_host (hipHostMalloc / hipHostFree) */
void *mi_host_alloc(size_t bytes);
void mi_host_free(void *p);

/* Stream contract of an engine: every call is ordered after the previous call on the same engine, whatever streams
 * the two calls used (the engine chains them with events), so history and staging buffers never race. An engine is
 * not thread-safe: one host thread at a time.
 *
 * Filter changes between blocks:
 *   mi_filter_set_eq on a filter that engines are using is glitch-free: the new tables are built and uploaded beside
 *   the live ones and published atomically; calls enqueued earlier finish on the old spectrum, the next call uses the
 *   new one; a failed rebuild leaves the old spectrum in place. mi_filter_generation counts published table sets,
 *   mi_engine_last_generation tells which one the engine's latest call used.
 *   mi_engine_rebind switches an engine to ANOTHER resident filter (other ratio / phase / rate family) at a block
 *   boundary. The carried input history is kept when both filters keep the same number of history frames and
 *   reset_history is 0; otherwise it is zeroed (the state after LoadFilter, vulkan_streaming_upsampler.cpp:741). */
int mi_engine_rebind(mi_engine *e, mi_filter *f, int reset_history);
unsigned long long mi_filter_generation(const mi_filter *f);
unsigned long long mi_engine_last_generation(const mi_engine *e);
/* test hook: the next table upload of this filter fails after the host-side build (tests that a failed EQ change
 * leaves the filter usable) */
void mi_debug_fail_next_table_upload(mi_filter *f);

/* Timing of the dominant kernel(s) on the stream they run on: with slots > 0
 * every process call brackets its main kernel(s) with a hipEvent pair (ring of
 * `slots` pairs; 0 disables). last_kernel_ms waits for the latest call;
 * kernel_ms_stats waits for and averages every recorded call. */
int mi_engine_enable_kernel_timing(mi_engine *e, int slots);
double mi_engine_last_kernel_ms(mi_engine *e);
int mi_engine_kernel_ms_stats(mi_engine *e, double *avg, double *min_ms, double *max_ms, int *count);

/* ------------------------------------------------------------------ (3) --
 * Host-only helpers (no GPU needed): the caller-side logic around the path,
 * exported so the streamer and the CPU test-suite use one implementation.
 */
/* LoadFilterConfig + LoadCoefficients + taps<=fft guard without touching a
 * device (vulkan_streaming_upsampler.cpp:606-732). MI_OK or MI_ERR_FILTER. */
int mi_read_filter(const char *json_path, mi_ups_config *out, char *err, size_t errcap);
/* ResolveFilterPath (src/alsa/alsa_filter_selector.cpp:8-108). Returns 1 and
 * the path when a filter was selected, 0 otherwise (err may hold a message). */
int mi_resolve_filter_path(const char *filter_path, const char *filter_dir, const char *phase, unsigned ratio,
                           unsigned input_rate, char *out_path, size_t out_cap, char *err, size_t errcap);
/* ParseFormat (alsa_common.cpp:12-27): MI_PCM_* or -1 */
int mi_parse_format(const char *name);
size_t mi_bytes_per_sample(int fmt);
/* ConvertPcmToFloat / ConvertFloatToPcm (alsa_common.cpp:42-127), n samples */
int mi_pcm_to_float(const void *src, int fmt, size_t n, float *dst);
int mi_float_to_pcm(const float *src, size_t n, int fmt, void *dst);
/* parseEqString (eq_parser.cpp:177-259): returns the band count (or -1 when the
 * reference returns false); writes up to max_bands rows of 9 doubles:
 * enabled,type,frequency,gain,q,hasBwHz,bwHz,hasBwOct,bwOct */
long mi_eq_parse(const char *text, double *preamp_db, double *bands9, size_t max_bands);
int mi_eq_parse_filter_type(const char *s);
const char *mi_eq_filter_type_name(int type);
/* calculateBiquadCoeffs (eq_to_fir.cpp:9-75): b0,b1,b2,a1,a2 */
int mi_eq_biquad(int enabled, int type, double freq, double gain, double q, double fs, double *out5);
/* computeEqResponseForFft / computeEqMagnitudeForFft, host fp64 */
int mi_eq_response_host(const char *text, size_t num_bins, size_t full_fft, double fs_out, double *out_reim);
int mi_eq_magnitude_host(const char *text, size_t num_bins, size_t full_fft, double fs_out, double *out_mag);

/* Layout facts of the fused kernel (K = 2^log2k complex points per channel-block,
 * 5 <= log2k <= 14), exported so that the CPU tests can verify bank-conflict
 * freedom and the mirror pairing without a GPU:
 *   mi_lds_swizzle        LDS word index -> physical word index
 *   mi_fused_set_of_block sixteen-bin set {a + t*K/16} held by LDS block b after the forward FFT
 *   mi_fused_block_a      first LDS block of thread tau in the two passes next to the spectral stage */
int mi_lds_swizzle(int word_index);
int mi_fused_set_of_block(int block, int log2k);
int mi_fused_block_a(int tau, int log2k);

/* Load-time tables as the kernels will see them (host build, for inspection):
 * geometry[10] = log2k,K,M,P,S,Oc,Bc,n_in,B,hist_frames; which: 0 Gs, 1 Gc,
 * 2 Wm, 3 tw in natural bin order; 4 WmT, 5 GT, 6 G0 in the fused kernel's
 * thread order (empty outside its range); all interleaved re,im float32.
 * mi_tables_block_b: the per-thread LDS block table. apo_text may be NULL. */
typedef struct mi_tables mi_tables;
int mi_tables_build(const char *json_path, int flags, const char *apo_text, double fs_out, mi_tables **out, char *err,
                    size_t errcap);
int mi_tables_geometry(const mi_tables *t, int *geometry10);
size_t mi_tables_size(const mi_tables *t, int which); /* complex elements */
int mi_tables_copy(const mi_tables *t, int which, float *out_reim, size_t cap_complex);
int mi_tables_block_b(const mi_tables *t, int *out, size_t cap);
void mi_tables_free(mi_tables *t);

#ifdef __cplusplus
}
#endif
#endif /* MI_UPSAMPLER_H */
