// TEST-ONLY fake of the ALSA calls declared in alsa/asoundlib.h (this directory): a capture device that plays back the
// file named by FAKE_ALSA_CAPTURE and a playback device that records into FAKE_ALSA_PLAYBACK. On purpose it is an awkward
// device: reads and writes are short, every 5th call says -EAGAIN, every 7th reports an XRUN (-EPIPE) that stays until
// snd_pcm_recover has been called, and the capture ends with an unrecoverable error (-EBADFD), which is how the streamer
// learns that a device has gone (alsa_common.cpp:269-336 in the reference: recover on -EPIPE / -ESTRPIPE / -EINTR).
#include <alsa/asoundlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

struct _snd_pcm {
  FILE *f = nullptr;
  bool capture = false, xrun = false;
  unsigned long calls = 0, frameBytes = 4, recovered = 0, waits = 0;
};

extern "C" {

int snd_pcm_open(snd_pcm_t **pcm, const char *name, snd_pcm_stream_t stream, int) {
  const bool cap = stream == SND_PCM_STREAM_CAPTURE;
  const char *path = std::getenv(cap ? "FAKE_ALSA_CAPTURE" : "FAKE_ALSA_PLAYBACK");
  if (!name || std::strncmp(name, "fake", 4) != 0 || !path) {
    return -ENOENT;
  }
  FILE *f = std::fopen(path, cap ? "rb" : "wb");
  if (!f) {
    return -ENOENT;
  }
  *pcm = new _snd_pcm();
  (*pcm)->f = f;
  (*pcm)->capture = cap;
  return 0;
}

int snd_pcm_close(snd_pcm_t *pcm) {
  if (pcm) {
    if (const char *stats = std::getenv("FAKE_ALSA_STATS")) {
      if (FILE *s = std::fopen(stats, "a")) {
        std::fprintf(s, "%s calls=%lu recovered=%lu waits=%lu\n", pcm->capture ? "capture" : "playback", pcm->calls,
                     pcm->recovered, pcm->waits);
        std::fclose(s);
      }
    }
    std::fclose(pcm->f);
    delete pcm;
  }
  return 0;
}

int snd_pcm_set_params(snd_pcm_t *pcm, snd_pcm_format_t format, snd_pcm_access_t access, unsigned int channels, unsigned int rate,
                       int, unsigned int latency) {
  if (!pcm || access != SND_PCM_ACCESS_RW_INTERLEAVED || channels == 0 || rate == 0 || latency == 0) {
    return -EINVAL;
  }
  const unsigned long b = format == SND_PCM_FORMAT_S16_LE ? 2 : (format == SND_PCM_FORMAT_S24_3LE ? 3 : 4);
  pcm->frameBytes = b * channels;
  return 0;
}

static long Awkward(snd_pcm_t *pcm) {  // 0 = go ahead
  ++pcm->calls;
  if (pcm->xrun) {
    return -EPIPE;
  }
  if (pcm->calls % 7 == 0) {
    pcm->xrun = true;
    return -EPIPE;
  }
  if (pcm->calls % 5 == 0) {
    return -EAGAIN;
  }
  return 0;
}

snd_pcm_sframes_t snd_pcm_readi(snd_pcm_t *pcm, void *buffer, snd_pcm_uframes_t size) {
  if (const long e = Awkward(pcm)) {
    return e;
  }
  const snd_pcm_uframes_t want = size / 2 + 1 < size ? size / 2 + 1 : size;  // short reads
  const size_t got = std::fread(buffer, pcm->frameBytes, want, pcm->f);
  return got ? static_cast<snd_pcm_sframes_t>(got) : -EBADFD;  // end of the recording: the device is gone
}

snd_pcm_sframes_t snd_pcm_writei(snd_pcm_t *pcm, const void *buffer, snd_pcm_uframes_t size) {
  if (const long e = Awkward(pcm)) {
    return e;
  }
  const snd_pcm_uframes_t take = size / 3 + 1 < size ? size / 3 + 1 : size;  // short writes
  return static_cast<snd_pcm_sframes_t>(std::fwrite(buffer, pcm->frameBytes, take, pcm->f));
}

int snd_pcm_wait(snd_pcm_t *pcm, int) {
  ++pcm->waits;
  return 1;
}

int snd_pcm_recover(snd_pcm_t *pcm, int err, int) {
  if (err == -EPIPE || err == -ESTRPIPE || err == -EINTR) {
    pcm->xrun = false;
    ++pcm->recovered;
    return 0;
  }
  return err;
}

int snd_pcm_drop(snd_pcm_t *) { return 0; }
int snd_pcm_drain(snd_pcm_t *pcm) { return std::fflush(pcm->f); }
const char *snd_strerror(int errnum) { return std::strerror(errnum < 0 ? -errnum : errnum); }
}
