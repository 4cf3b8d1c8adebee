"""forward product at the wide-stack shape of BASELINE configs[4] (3 x 8192 rows, K = N = 4096): bf16 kernel vs fp8 kernel"""
import sys
sys.path.insert(0, '.')
import torch
from mr_gan_amd import engine as E
M, N, K = 24576, 4096, 4096
x = torch.randn(M, K, device="cuda:0"); w = torch.randn(K, N, device="cuda:0") / 64
gf = 2.0 * M * N * K / 1e9
for cfg in (1, 3):
    _, us8 = E.debug_gemm_fp8(x, w, None, act=1, scale_a=32.0, scale_b=2048.0, reps=20, kc_cfg=cfg)
    print("fp8  e4m3 forward %d x %d x %d (tile cfg %d): %.1f us  %.0f TFLOP/s" % (M, N, K, cfg, us8, 1e3 * gf / us8))
for cfg in (1, 2, 3):
    us = E.debug_gemm_time(4, 8192, N, K, 3, 1, reps=20, kc_cfg=cfg)
    print("bf16 forward (plain relu, tile cfg %d): %.1f us  %.0f TFLOP/s" % (cfg, us, 1e3 * gf / us))
