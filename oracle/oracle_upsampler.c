/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the reference's
 * overlap-save FIR upsampler path. Never linked into, loaded by, or called
 * from the product path (totton-rasp-gpu-dsp_amd/). Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * Parity status: PINNED. tests/test_oracle.py checks this file bit-for-bit
 * against oracle/_ref (the reference's own C++ compiled from /root/reference)
 * when that library is present, and against the committed golden vectors in
 * tests/golden/ (generated from oracle/_ref by tests/golden/make_golden.py)
 * everywhere else.
 *
 * Plain C, fp32 arithmetic in the reference's own operation order. Build with
 * -ffp-contract=off so no FMA contraction changes the rounding.
 *
 * What is restated (paths relative to /root/reference):
 *   orc_fft()            src/vulkan/fft_utils.h:14-61   (BitReverse + Fft)
 *   orc_ups_prepare()    src/vulkan/vulkan_streaming_upsampler.cpp:726-753
 *   orc_ups_process()    src/vulkan/vulkan_streaming_upsampler.cpp:500-596
 *   orc_ups_reset()      src/vulkan/vulkan_streaming_upsampler.cpp:598-600
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

/* fft_utils.h:14-28 -- in-place bit-reversal permutation (interleaved re/im) */
static void orc_bit_reverse(float *d, size_t n) {
  size_t j = 0;
  for (size_t i = 1; i < n; ++i) {
    size_t bit = n >> 1;
    while (j & bit) {
      j ^= bit;
      bit >>= 1;
    }
    j ^= bit;
    if (i < j) {
      float tr = d[2 * i], ti = d[2 * i + 1];
      d[2 * i] = d[2 * j];
      d[2 * i + 1] = d[2 * j + 1];
      d[2 * j] = tr;
      d[2 * j + 1] = ti;
    }
  }
}

/* fft_utils.h:30-61 -- radix-2 DIT, fp32, twiddles by recurrence w *= wlen,
 * inverse scaled by 1/n. `d` holds n interleaved (re, im) pairs. */
void orc_fft(float *d, size_t n, int inverse) {
  if (n <= 1) {
    return;
  }
  orc_bit_reverse(d, n);
  const float kPi = 3.14159265358979323846f;
  for (size_t len = 2; len <= n; len <<= 1) {
    const float angle = (inverse ? 2.0f : -2.0f) * kPi / (float)len;
    const float wlr = cosf(angle), wli = sinf(angle);
    for (size_t i = 0; i < n; i += len) {
      float wr = 1.0f, wi = 0.0f;
      for (size_t j = 0; j < len / 2; ++j) {
        float *a = d + 2 * (i + j);
        float *b = d + 2 * (i + j + len / 2);
        const float ur = a[0], ui = a[1];
        /* v = data[i+j+len/2] * w */
        const float vr = b[0] * wr - b[1] * wi;
        const float vi = b[0] * wi + b[1] * wr;
        a[0] = ur + vr;
        a[1] = ui + vi;
        b[0] = ur - vr;
        b[1] = ui - vi;
        /* w *= wlen */
        const float nwr = wr * wlr - wi * wli;
        const float nwi = wr * wli + wi * wlr;
        wr = nwr;
        wi = nwi;
      }
    }
  }
  if (inverse) {
    const float inv = 1.0f / (float)n;
    for (size_t i = 0; i < 2 * n; ++i) {
      d[i] *= inv;
    }
  }
}

typedef struct {
  size_t taps, fft, block, factor;
  float *spectrum; /* fft interleaved complex: filterSpectrum_ */
  float *overlap;  /* fft-block floats: overlap_ (zero-stuffed input domain) */
  float *time;     /* scratch: timeBuffer */
  float *freq;     /* scratch: freqBuffer */
} orc_ups;

void orc_ups_destroy(orc_ups *u) {
  if (!u) {
    return;
  }
  free(u->spectrum);
  free(u->overlap);
  free(u->time);
  free(u->freq);
  free(u);
}

/* PrepareSpectrum (:726-753): H = FFT_N(zero-pad(h)) with the fp32 radix-2
 * FFT above; overlap = zeros(N - B). Geometry validation lives in the caller
 * (tests restate LoadFilterConfig's rules separately). */
orc_ups *orc_ups_prepare(const float *coeffs, size_t taps, size_t fft,
                         size_t block, size_t factor) {
  if (taps > fft || block >= fft) {
    return NULL;
  }
  orc_ups *u = (orc_ups *)calloc(1, sizeof(orc_ups));
  u->taps = taps;
  u->fft = fft;
  u->block = block;
  u->factor = factor ? factor : 1;
  u->spectrum = (float *)calloc(2 * fft, sizeof(float));
  u->overlap = (float *)calloc(fft - block, sizeof(float));
  u->time = (float *)calloc(fft, sizeof(float));
  u->freq = (float *)calloc(2 * fft, sizeof(float));
  for (size_t i = 0; i < taps; ++i) {
    u->spectrum[2 * i] = coeffs[i];
  }
  orc_fft(u->spectrum, fft, 0);
  return u;
}

void orc_ups_reset(orc_ups *u) {
  memset(u->overlap, 0, (u->fft - u->block) * sizeof(float));
}

/* Copy of the spectrum the reference multiplies by (for the Vulkan-path
 * simulation: inaccurate H x accurate signal FFTs). */
void orc_ups_get_spectrum(const orc_ups *u, float *out_reim) {
  memcpy(out_reim, u->spectrum, 2 * u->fft * sizeof(float));
}

/* ProcessBlock, CPU-fallback branch (:500-534, :577-595).
 * Returns block size, or 0 for every case in which the reference returns an
 * empty vector. */
long orc_ups_process(orc_ups *u, const float *input, size_t count, float *out) {
  if (!u || !input || count == 0) {
    return 0;
  }
  const size_t L = u->factor > 1 ? u->factor : 1;
  if (u->block % L != 0) {
    return 0;
  }
  const size_t max_in = L > 1 ? u->block / L : u->block;
  if (max_in == 0 || count != max_in) {
    return 0;
  }
  const size_t n = u->fft;
  const size_t ov = n - u->block;
  const size_t up = count * L;
  if (ov + up > n) {
    return 0;
  }
  memset(u->time, 0, n * sizeof(float));
  memcpy(u->time, u->overlap, ov * sizeof(float));
  for (size_t i = 0; i < count; ++i) {
    u->time[ov + i * L] = input[i];
  }
  for (size_t i = 0; i < n; ++i) {
    u->freq[2 * i] = u->time[i];
    u->freq[2 * i + 1] = 0.0f;
  }
  orc_fft(u->freq, n, 0);
  for (size_t i = 0; i < n; ++i) {
    const float ar = u->freq[2 * i], ai = u->freq[2 * i + 1];
    const float br = u->spectrum[2 * i], bi = u->spectrum[2 * i + 1];
    u->freq[2 * i] = ar * br - ai * bi;
    u->freq[2 * i + 1] = ar * bi + ai * br;
  }
  orc_fft(u->freq, n, 1);
  for (size_t i = 0; i < up; ++i) {
    out[i] = u->freq[2 * (ov + i)];
  }
  memcpy(u->overlap, u->time + (n - ov), ov * sizeof(float));
  return (long)up;
}
