import sys
sys.path.insert(0, '.')
import torch
from mr_gan_amd import engine as E
E.load_library('mr_gan_amd/lib/libmrgan_hip_stamps.so')
M, N, K = 24576, 4096, 4096
x = torch.randn(M, K, device="cuda:0"); w = torch.randn(K, N, device="cuda:0") / 64
for cfg in (1, 3):
    _, us8 = E.debug_gemm_fp8(x, w, None, act=1, scale_a=32.0, scale_b=2048.0, reps=20, kc_cfg=cfg)
    print(cfg, us8)
