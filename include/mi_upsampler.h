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
/* optional EQ (reference: eq_parser.cpp / eq_to_fir.cpp, no call site there). The APO profile's biquad cascade is folded
 * into the FIR itself: its recursion runs over the filter taps in fp64 and the result is cut back to `taps` samples
 * (closing half-Hann over the last (taps-1)/64), so the product is still ONE linear convolution with at most `taps`
 * samples, fft_size - block_size == taps - 1 still holds and nothing wraps around inside a block. What the
 * cut drops is measured (mi_eq_residual). Above the limit (default 1e-3 = -60 dB, mi_*_set_eq_limit) the call still
 * succeeds and mi_ups_last_error() holds a warning that says by how much; a handle made strict refuses instead
 * (MI_ERR_FILTER, the previous spectrum stays). Empty/NULL text removes the EQ. */
int mi_ups_set_eq(mi_ups *h, const char *apo_text, double fs_out);
/* What the latest successful EQ change dropped. h_ideal = taps (*) cascade is infinitely long, the filter uses
 * fir[n] = w[n] h_ideal[n], n < taps:
 *   tail_l1 = ||h_ideal - fir||_1 / ||h_ideal||_1   output error <= tail_l1 * ||h_ideal||_1 * max|x| for ANY input
 *   tail_l2 = the same in the 2-norm (error power for white input);   *_db = 20 log10 (-400 for 0)
 *   response_dev = max_k |FFT_N(fir)[k] - H_fir[k] EQ(f_k)| / max_k |H_fir EQ|, EQ(f_k) = the cascade evaluated per bin on
 *                  the device as computeEqResponseForFft does (eq_to_fir.cpp:145-151)
 *   tail_complete = 0: the free decay had not died after 2e8 section-steps; its rest is estimated from the pole radius */
typedef struct mi_eq_residual {
  int active;        /* 0: no EQ folded in (everything else 0) */
  int over_limit;    /* tail_l1 > limit */
  int tail_complete;
  int reserved;
  double tail_l1, tail_l2, tail_l1_db, tail_l2_db, response_dev, response_dev_db, limit;
  size_t fir_taps, taper;
} mi_eq_residual;
int mi_ups_eq_residual(const mi_ups *h, mi_eq_residual *out);
/* max_tail_l1 < 0: the default (1e-3); strict != 0: an EQ over the limit is refused */
int mi_ups_set_eq_limit(mi_ups *h, double max_tail_l1, int strict);

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
int mi_filter_eq_residual(const mi_filter *f, mi_eq_residual *out);
int mi_filter_set_eq_limit(mi_filter *f, double max_tail_l1, int strict);
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
/* Small calls (fewer channel-blocks than CUs / 2) on the fused path split every channel-block's P output phases over
 * several workgroups (each repeats the forward transform): returns that count for the latest call, 0 = not split.
 * The reference's own call shape, one channel-block per call, takes P workgroups instead of one. */
int mi_engine_last_phase_parts(const mi_engine *e);
/* 1 when the latest call of a "staged" engine (transform lengths past the fused kernels: the 640 001-tap "2m" filters at
 * 2x / 4x / 8x) ran the two-level transforms -- K = K1 x M2 with the M2-point rows in LDS, two HBM round trips per transform
 *  -- and 0 when it ran one launch per radix-16 pass (K outside 2^15 .. 2^18, or MIUPS_EXP_NO_TWO_LEVEL). */
int mi_engine_last_two_level(const mi_engine *e);
/* 1 when the latest call of a "fused" engine whose frames are wider than a workgroup's channel group (more than two
 * channels) let the transform kernel's own workgroups assemble the PCM frames of finished (stream, block) pairs while the
 * rest of the launch was still computing (cooperative frames); the frame pass behind the kernel then only
 * takes the tiles nobody claimed. 0: the frame pass assembled every frame (MIUPS_EXP_NO_COOP_FRAMES, other shapes). */
int mi_engine_last_coop_frames(const mi_engine *e);

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
/* ... or pin a buffer the caller already owns (hipHostRegister / hipHostUnregister): do it ONCE for a buffer that is
 * passed to mi_engine_process_host / mi_multi_process_host again and again; unregister before freeing it */
int mi_host_register(void *p, size_t bytes);
void mi_host_unregister(void *p);
/* measured device-to-device copy rate of `device`, GB/s counting bytes read + bytes written (16-byte-per-lane copy
 * kernel over `bytes` per buffer, best of `iters`): the bench line's copy ceiling beside the HBM spec peak */
int mi_device_copy_rate(int device, size_t bytes, int iters, double *gbps);

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
/* The rule the host-buffer paths rest on: never two asynchronous copies in flight on host ranges that are
 * not page-locked when they share a page OR are both large enough for the runtime to pin them (>= 128 KiB). Every copy
 * mi_engine_process_host / mi_multi_process_host / mi_ups_process_block issues is audited against it; this returns how
 * many broke it since the process started. Tests require 0. */
unsigned long long mi_debug_unsafe_host_copies(void);
/* test hooks: the next mi_engine_process_host (of slot `slot`'s engine for the multi form) fails right after issuing the
 * host-to-device copies of its sub-batch `sub_batch` (0-based), as if the runtime had refused a call there. The call must
 * return an error only after every copy in flight has completed, and the engine stays usable (reset it). */
void mi_debug_fail_host_call_at(mi_engine *e, int sub_batch);
/* diagnostic: on SIGABRT print the native stack of the aborting thread to stderr, then continue to the previous handler
 * (the HIP runtime aborts without a message on some internal failures) */
void mi_debug_install_abort_backtrace(void);

/* Timing of the dominant kernel(s) on the stream they run on: with slots > 0
 * every process call brackets its main kernel(s) with a hipEvent pair (ring of
 * `slots` pairs; 0 disables). last_kernel_ms waits for the latest call;
 * kernel_ms_stats waits for and averages every recorded call. */
int mi_engine_enable_kernel_timing(mi_engine *e, int slots);
/* Only every `every`-th process call carries the pair (default 1 = each call). An event pair costs about 8 us of stream
 * time (profiles/r03_n_step_overhead.txt): a timed region samples instead of paying that on every call. */
int mi_engine_set_kernel_timing_stride(mi_engine *e, int every);
double mi_engine_last_kernel_ms(mi_engine *e);
int mi_engine_kernel_ms_stats(mi_engine *e, double *avg, double *min_ms, double *max_ms, int *count);
/* Per kernel class of the latest call (diagnostic, off by default): out4 = ms of [0] planarize, [1] transform,
 * [2] frame assembly (interleave kernels), [3] history carry; -1 where the call had no such launch. Every launch gets
 * its own event pair on the stream it runs on, which perturbs the call: use it beside, not inside, a timed region.
 * Classes that overlap on two streams (pipelined launches) add up to more than the call. */
int mi_engine_enable_class_timing(mi_engine *e, int on);
int mi_engine_last_class_ms(mi_engine *e, double *out4);

/* ----------------------------------------------------------------- (2b) --
 * Multi-GPU: the independent units of the path sharded over the GPUs of one node. Two partitions:
 *   default                    stream s runs on slot s mod n_devices (many independent streams);
 *   MI_MULTI_SPLIT_CHANNELS    (OR it into `flags`) the channels of EVERY stream are cut into n_devices contiguous groups,
 *                              group i = channels [i C / n, (i+1) C / n) on slot i: ONE wide stream over several GPUs. Each
 *                              slot moves its column group straight out of / into the caller's interleaved frames with
 *                              pitched 2-D copies: no host de-interleave, each device's link carries only its channels.
 * Every slot owns a device, its own filter tables, histories, HIP streams, staging and one host worker thread pinned to
 * the CPUs local to its device (sysfs local_cpulist). No data is exchanged between devices (reference: channels are
 * independent objects, alsa_streamer_main.cpp:247-250,536-553). A device may be listed twice (two slots on
 * one GPU). Creation fails with a message when a listed device is not visible.
 * Host buffers: use mi_host_alloc memory, or pin your own buffer ONCE with mi_host_register, for buffers that are passed
 * again and again. Pageable buffers work too: the call page-locks them for its own duration (hipHostRegister) because the
 * slots copy different ranges of them at the same time, which is not safe on pageable memory; when that is
 * refused the slots run one after the other.
 * mi_multi_set_eq is all-or-nothing: every slot's new tables are built first and published together; when one build
 * fails no slot changes.
 */
#define MI_MULTI_SPLIT_CHANNELS 0x10000
/*   MI_MULTI_SPLIT_TIME        the BLOCKS of every stream are cut into n_devices contiguous ranges, range i on slot i, all
 *                              channels. Copies stay contiguous (full link rate per device: pitched DMA of narrow channel
 *                              groups measured 3-11 GB/s against 54, profiles/r03_h_multi_split.txt). A block depends on
 *                              earlier ones only through the input history, so each slot is handed the (taps-1)/L input
 *                              frames in front of its range; the object keeps the tail of the previous call for the ranges
 *                              at the start of the next one. Output is bit-identical to one engine. For calls of many
 *                              blocks (file / batch processing); a one-block call runs on slot 0 alone. */
#define MI_MULTI_SPLIT_TIME 0x20000
typedef struct mi_multi mi_multi;
int mi_multi_create(const char *json_path, int flags, const int *devices, size_t n_devices, int streams, int channels,
                    int in_fmt, int out_fmt, mi_multi **out, char *err, size_t errcap);
void mi_multi_destroy(mi_multi *m);
int mi_multi_set_eq(mi_multi *m, const char *apo_text, double fs_out);
int mi_multi_eq_residual(const mi_multi *m, mi_eq_residual *out);
int mi_multi_set_eq_limit(mi_multi *m, double max_tail_l1, int strict);
int mi_multi_reset(mi_multi *m);
/* all streams, HOST buffers, stream s at base + s*stride; returns when every device has finished */
int mi_multi_process_host(mi_multi *m, const void *h_in, size_t in_stream_stride_bytes, void *h_out,
                          size_t out_stream_stride_bytes, size_t blocks);
size_t mi_multi_in_frames_per_block(const mi_multi *m);
size_t mi_multi_out_frames_per_block(const mi_multi *m);
int mi_multi_device_of_stream(const mi_multi *m, int stream);
int mi_multi_device_of_channel(const mi_multi *m, int channel);
/* the channel partition itself (pure): first_channel_of_slot has slots + 1 entries, group i = [f[i], f[i+1]) */
int mi_multi_partition_channels(int channels, int slots, int *first_channel_of_slot);
/* the CPU list slot's worker thread is pinned to ("" = platform gave none, or none of them is allowed to this process) */
int mi_multi_worker_cpus(const mi_multi *m, int slot, char *out, size_t cap);
/* test hook: the next mi_multi_set_eq fails while building slot `slot`'s tables */
void mi_debug_multi_fail_next_eq_on_slot(mi_multi *m, int slot);
/* test hook: mi_debug_fail_host_call_at for the engine of slot `slot` */
void mi_debug_multi_fail_host_call_at(mi_multi *m, int slot, int sub_batch);
/* the partition itself (pure): slot_of_stream[s] = s mod slots */
int mi_multi_partition(int streams, int slots, int *slot_of_stream);

/* ----------------------------------------------------------------- (2c) --
 * Resident spectra for every (rate family, ratio, phase) a filter directory serves: a change of input rate inside a
 * family, of the ratio or of the phase setting is mi_engine_rebind to a filter that is already in HBM (reference
 * keys: src/alsa/alsa_filter_selector.cpp:33-55; "same-family switching is instant",
 * src/audio/auto_negotiation.cpp:139-152).
 */
typedef struct mi_bank mi_bank;
int mi_bank_load(int device, const char *filter_dir, mi_bank **out, char *warnings, size_t warncap, char *err,
                 size_t errcap);
void mi_bank_release(mi_bank *b);
size_t mi_bank_size(const mi_bank *b);
/* entry i: family base rate (44100 | 48000), ratio, phase ("min" | "linear"), sidecar path, geometry */
int mi_bank_entry(const mi_bank *b, size_t i, unsigned *family_base_rate, unsigned *ratio, char *phase, size_t phasecap,
                  char *path, size_t pathcap, mi_ups_config *config);
/* a new handle on the resident filter for this key (release it with mi_filter_release; the tables stay in the
 * bank), or NULL with the selector's message */
mi_filter *mi_bank_select(const mi_bank *b, unsigned input_rate, unsigned ratio, const char *phase, char *err,
                          size_t errcap);

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
/* The EQ-folded FIR itself, host only (what mi_*_set_eq hands to the table build): out_fir gets n_taps doubles (may be
 * NULL), *out the residual (response_dev against the host evaluation of the cascade; limit = the default). */
int mi_eq_fold_host(const float *taps, size_t n_taps, size_t fft_size, const char *apo_text, double fs_out, double *out_fir,
                    mi_eq_residual *out);

/* Rate-family detection and output-rate negotiation (src/audio/auto_negotiation.cpp:13-155). The DAC is given as the
 * reference's capability probe reports it: valid flag, min/max rate, optional list of discrete rates. */
typedef struct mi_negotiated {
  int input_rate;
  int family;      /* 0 unknown, 1 = 44.1k family, 2 = 48k family */
  int output_rate;
  int ratio;
  int valid;
  int requires_reconfiguration;
  char error[256];
} mi_negotiated;
int mi_rate_family(int sample_rate);
int mi_same_family(int rate_a, int rate_b);
int mi_upsample_ratio(int input_rate, int output_rate);
int mi_negotiate(int input_rate, int dac_valid, int dac_min_rate, int dac_max_rate, const int *dac_rates, size_t n_rates,
                 int current_output_rate, mi_negotiated *out);

/* config.json of the reference's control plane (release/config.example.json): the keys the data plane acts on. */
typedef struct mi_runtime_config {
  int eq_enabled;
  char eq_profile[256];
  char eq_profile_path[1024];
  unsigned ratio;
  char phase_type[32];
  char filter_directory[1024];
  unsigned sample_rate, channels, period_frames, buffer_frames;
  char format[32];
  char input_device[128], output_device[128];
} mi_runtime_config;
int mi_parse_runtime_config(const char *json_text, mi_runtime_config *out, char *err, size_t errcap);

/* One OPRA EQ record (JSON: {"parameters": {"gain_db", "bands": [{"type", "frequency", "gain_db", "q", "slope"}]}, ...})
 * -> the Equalizer APO text the reference's control plane writes for it (scripts/integration/opra.py:50-244:
 * convert_opra_to_apo + EqProfile.to_apo_format; modern_target != 0: apply_modern_target_correction, the KB5000_7 band).
 * The text goes to mi_filter_set_eq / mi_ups_set_eq. Returns MI_OK, MI_ERR_FILTER with a message for malformed JSON,
 * MI_ERR_ARG when `out` is too small (needed size in *needed, if given). */
int mi_opra_to_apo(const char *eq_json, int modern_target, char *out, size_t cap, size_t *needed, char *err, size_t errcap);

/* SPSC PCM ring (contract of include/io/audio_ring_buffer.h:23-100, on bytes) -- exported for the tests */
typedef struct mi_ring mi_ring;
mi_ring *mi_ring_create(size_t capacity_bytes);
void mi_ring_destroy(mi_ring *r);
int mi_ring_write(mi_ring *r, const void *data, size_t bytes);  /* 1 = written, 0 = would not fit (nothing written) */
int mi_ring_read(mi_ring *r, void *dst, size_t bytes);           /* 1 = read, 0 = not that much available */
size_t mi_ring_available_to_read(const mi_ring *r);
size_t mi_ring_available_to_write(const mi_ring *r);
void mi_ring_clear(mi_ring *r);

/* The streaming loop of the reference (alsa_streamer_main.cpp:428-611) over callbacks: read one period, stage it,
 * process every complete block that fits, drain period*ratio chunks, write silence when nothing is ready, drop the
 * staging on overflow. process == NULL or block_in_frames == 0: PCM pass-through. `running` is polled; set it to 0
 * from a signal handler to stop. between_blocks (may be NULL) runs before every process call. */
typedef long (*mi_read_fn)(void *user, void *dst, size_t frames);
typedef int (*mi_write_fn)(void *user, const void *src, size_t frames);
typedef int (*mi_process_fn)(void *user, const void *in, void *out, size_t blocks);
typedef void (*mi_between_fn)(void *user);
typedef void (*mi_log_fn)(void *user, const char *message);
typedef struct mi_loop_params {
  unsigned channels;
  int format;
  size_t period_frames, block_in_frames, block_out_frames, max_blocks_per_call;
  int drain_at_end; /* additive: process the zero-padded tail and flush (the reference stops at the first short read) */
  int pinned_rings; /* additive: the two staging rings live in mi_host_alloc memory, and a batch that does not wrap a
                     * ring's end is handed to `process` IN PLACE (no bounce copy; DMA-able by mi_engine_process_host) */
} mi_loop_params;
typedef struct mi_loop_stats {
  size_t periods_read, blocks_processed, frames_written, silence_frames_written, input_overflows, output_overflows,
      process_calls;
  size_t in_place_calls; /* process calls whose input and output were both contiguous pieces of the rings */
} mi_loop_stats;
int mi_stream_loop_run(const mi_loop_params *p, mi_read_fn read, mi_write_fn write, mi_process_fn process,
                       mi_between_fn between, mi_log_fn log, void *user, const volatile int *running,
                       mi_loop_stats *stats);

/* Layout facts of the fused kernel (K = 2^log2k complex points per channel-block,
 * 5 <= log2k <= 14), exported so that the CPU tests can verify bank-conflict
 * freedom and the mirror pairing without a GPU:
 *   mi_lds_swizzle        LDS word index -> physical word index
 *   mi_fused_set_of_block sixteen-bin set {a + t*K/16} held by LDS block b after the forward FFT
 *   mi_fused_block_a      first LDS block of thread tau in the two passes next to the spectral stage
 *   mi_fused_plan_radices the pass plan itself */
int mi_lds_swizzle(int word_index);
int mi_fused_set_of_block(int block, int log2k);
int mi_fused_block_a(int tau, int log2k);
/* radices of the in-LDS passes, forward order (2^(log2k mod 4), 16, .., 16); returns the pass count, -1 when out is
 * too small */
int mi_fused_plan_radices(int log2k, int *out, size_t cap);

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
