"""Log-mel front end of the contact-microphone modality (mr_gan.py:42-47) on the GPU.

    S = librosa.feature.melspectrogram(contact, sr=48000, n_mels=128); log_S = librosa.logamplitude(S, ref_power=np.max)

is one launch of logmel_kernel for ALL trials of a data set (mrgan_logmel, include/mrgan_abi.h; csrc/logmel.hip): a trial per
workgroup, frames -> FFT -> power -> mel -> dB inside LDS.  There is no CPU path here; the numpy restatement used by the
tests lives in oracle/melspec_oracle.py."""
import ctypes as C

import numpy as np
import torch

from mr_gan_amd import engine as E


def logmel_frames(n_samples):
    return int(E.load_library().mrgan_logmel_frames(C.c_int64(int(n_samples))))


def log_melspectrogram_device(y, sr=48000, n_mels=128):
    """y: float32 device tensor [trials, samples] -> device tensor [trials, n_mels * frames] (log_S.flatten() per trial)"""
    if not y.is_cuda or y.dtype != torch.float32 or y.dim() != 2 or y.stride(1) != 1:
        raise ValueError("log_melspectrogram_device needs a float32 device matrix with contiguous rows")
    lib = E.load_library()
    frames = logmel_frames(y.shape[1])
    out = torch.empty((y.shape[0], n_mels * frames), dtype=torch.float32, device=y.device)
    with torch.cuda.device(y.device):
        rc = lib.mrgan_logmel(C.c_void_p(y.data_ptr()), C.c_int64(y.shape[0]), C.c_int64(y.shape[1]), C.c_int64(y.stride(0)),
                              C.c_int32(int(sr)), C.c_int32(int(n_mels)), C.c_void_p(out.data_ptr()), C.c_int64(out.stride(0)),
                              C.c_void_p(torch.cuda.current_stream(y.device).cuda_stream))
    if rc != 0:
        raise RuntimeError("mrgan_logmel failed (%d): %s" % (rc, lib.mrgan_last_error().decode()))
    return out


def log_melspectrogram_batch(signals, sr=48000, n_mels=128, device='cuda:0', chunk=4096):
    """signals: sequence of 1-D arrays (one per trial, lengths may differ) -> list of float32 vectors log_S.flatten()"""
    if not torch.cuda.is_available():
        raise RuntimeError("the log-mel front end runs on the GPU only (mrgan_logmel); no HIP device is visible")
    out = [None] * len(signals)
    by_len = {}
    for i, s in enumerate(signals):
        by_len.setdefault(len(s), []).append(i)
    for n, idx in by_len.items():
        for c0 in range(0, len(idx), chunk):
            part = idx[c0:c0 + chunk]
            host = np.stack([np.asarray(signals[i], dtype=np.float32) for i in part])
            feats = log_melspectrogram_device(torch.from_numpy(host).to(device), sr, n_mels).cpu().numpy()
            for k, i in enumerate(part):
                out[i] = feats[k]
    return out
