"""Time the log-mel front end (csrc/logmel.hip) on the MREO-sized workload: 6000 trials x 9600 samples -> 6000 x 2432 features.
Run on the GPU box:  python scripts/logmel_bench.py [--trials 6000] [--cpu-trials 200]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mr_gan_amd.melspec import log_melspectrogram_device       # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--trials', type=int, default=6000)
    ap.add_argument('--samples', type=int, default=9600)
    ap.add_argument('--cpu-trials', type=int, default=200)
    ap.add_argument('--reps', type=int, default=20)
    a = ap.parse_args()
    y = torch.randn((a.trials, a.samples), device='cuda:0')
    out = log_melspectrogram_device(y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        out = log_melspectrogram_device(y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    nbytes = a.trials * (a.samples + out.shape[1]) * 4
    flops = a.trials * (out.shape[1] // 128) * 5 * 2048 * 11
    print('gpu: %.3f ms per %d trials = %.2f M trials/s; algorithmic %.1f MB -> %.1f GB/s; FFT %.1f GFLOP -> %.2f TFLOP/s'
          % (ms, a.trials, a.trials / ms / 1e3, nbytes / 1e6, nbytes / ms / 1e6, flops / 1e9, flops / ms / 1e9))
    if a.cpu_trials:
        from oracle.melspec_oracle import log_melspectrogram
        yh = y[:a.cpu_trials].cpu().numpy().astype(np.float64)
        t0 = time.time()
        for r in yh:
            log_melspectrogram(r)
        dt = time.time() - t0
        print('cpu (numpy restatement, 1 process): %.2f ms per trial = %.0f trials/s' % (dt / a.cpu_trials * 1e3, a.cpu_trials / dt))


if __name__ == '__main__':
    main()
