// Log-mel front end of the contact-microphone modality (logmel.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace mrgan {

constexpr int LM_NFFT = 2048;          // librosa default n_fft (mr_gan.py:45 passes none)
constexpr int LM_HOP = 512;            // librosa default hop_length = n_fft / 4
constexpr int LM_MAX_MELS = 256;

inline int logmel_frames(long n_samples) { return 1 + (int)(n_samples / LM_HOP); }   // centred frames

// out[trial][m * n_frames + f] = log-mel power in dB relative to the trial's maximum, floored at -80 dB.
// Errors: -1 bad argument, -10 HIP failure (message in *err, static storage).
int launch_logmel(const float* y, long n_trials, long n_samples, long ld_y, int sr, int n_mels, float* out, long ld_out,
                  hipStream_t s, const char** err);

}  // namespace mrgan
