/* TEST-ONLY declaration stub of the slice of <alsa/asoundlib.h> that csrc/streamer_main.cpp uses behind HAVE_ALSA.
 * Neither image of this project carries the ALSA headers, so the branch was never compiled (missing item 5);
 * with this directory on the include path it is compiled AND run against tests/cpp/alsa_stub/fake_alsa.cpp, an
 * in-memory device pair that produces short reads, -EAGAIN and XRUNs on purpose. Signatures follow alsa-lib's public API
 * (pcm.h); nothing here ships with the product. */
#ifndef MIUPS_TEST_ALSA_STUB_H
#define MIUPS_TEST_ALSA_STUB_H

#include <errno.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct _snd_pcm snd_pcm_t;
typedef unsigned long snd_pcm_uframes_t;
typedef long snd_pcm_sframes_t;
typedef enum _snd_pcm_stream { SND_PCM_STREAM_PLAYBACK = 0, SND_PCM_STREAM_CAPTURE } snd_pcm_stream_t;
typedef enum _snd_pcm_access { SND_PCM_ACCESS_RW_INTERLEAVED = 3 } snd_pcm_access_t;
typedef enum _snd_pcm_format {
  SND_PCM_FORMAT_UNKNOWN = -1,
  SND_PCM_FORMAT_S16_LE = 2,
  SND_PCM_FORMAT_S32_LE = 10,
  SND_PCM_FORMAT_S24_3LE = 32
} snd_pcm_format_t;

int snd_pcm_open(snd_pcm_t **pcm, const char *name, snd_pcm_stream_t stream, int mode);
int snd_pcm_close(snd_pcm_t *pcm);
int snd_pcm_set_params(snd_pcm_t *pcm, snd_pcm_format_t format, snd_pcm_access_t access, unsigned int channels,
                       unsigned int rate, int soft_resample, unsigned int latency);
snd_pcm_sframes_t snd_pcm_readi(snd_pcm_t *pcm, void *buffer, snd_pcm_uframes_t size);
snd_pcm_sframes_t snd_pcm_writei(snd_pcm_t *pcm, const void *buffer, snd_pcm_uframes_t size);
int snd_pcm_wait(snd_pcm_t *pcm, int timeout);
int snd_pcm_recover(snd_pcm_t *pcm, int err, int silent);
int snd_pcm_drop(snd_pcm_t *pcm);
int snd_pcm_drain(snd_pcm_t *pcm);
const char *snd_strerror(int errnum);

#ifdef __cplusplus
}
#endif
#endif
