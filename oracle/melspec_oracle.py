"""CPU restatement of the log-mel front end of the contact-microphone modality (mr_gan.py:42-47) -- TEST INFRASTRUCTURE:
only tests/ may import this; the product path is the HIP kernel behind mrgan_logmel (mr_gan_amd/csrc/logmel.hip).

The reference calls librosa 0.5.1 (`melspectrogram(y, sr=48000, n_mels=128)` then
`logamplitude(S, ref_power=np.max)`); librosa is a third-party dependency that is neither in the reference tree nor
installed here and cannot be fetched, so this is a numpy (fp64) restatement of that version's published algorithm --
PARITY UNPINNED against librosa itself (no fixture of the reference holds a spectrogram):
  stft: n_fft 2048, hop 512, periodic Hann window, centred frames with reflect padding -> |X|^2
  mel basis: Slaney scale (linear below 1 kHz, log above), fmin 0, fmax sr/2, area ("Slaney") normalisation
  logamplitude: 10 log10(max(S, 1e-10)) - 10 log10(max(max S, 1e-10)), floored at max - 80 dB
"""
import numpy as np


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels=128, fmin=0.0, fmax=None):
    fmax = sr / 2.0 if fmax is None else fmax
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    weights = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    weights *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]        # Slaney area normalisation
    return weights


def power_stft(y, n_fft=2048, hop_length=512):
    y = np.asarray(y, dtype=np.float64)
    window = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n_fft) / n_fft)       # periodic Hann (fftbins=True)
    ypad = np.pad(y, n_fft // 2, mode='reflect')
    n_frames = 1 + (len(ypad) - n_fft) // hop_length
    idx = np.arange(n_fft)[:, None] + hop_length * np.arange(n_frames)[None, :]
    frames = ypad[idx] * window[:, None]
    return np.abs(np.fft.rfft(frames, axis=0)) ** 2


_BASIS = {}


def log_melspectrogram(y, sr=48000, n_mels=128, n_fft=2048, hop_length=512, amin=1e-10, top_db=80.0):
    key = (sr, n_fft, n_mels)
    if key not in _BASIS:
        _BASIS[key] = mel_filterbank(sr, n_fft, n_mels)
    S = _BASIS[key] @ power_stft(y, n_fft, hop_length)
    log_spec = 10.0 * np.log10(np.maximum(amin, S)) - 10.0 * np.log10(np.maximum(amin, S.max()))
    return np.maximum(log_spec, log_spec.max() - top_db)
