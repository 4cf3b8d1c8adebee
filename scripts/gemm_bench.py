"""Per-shape timing of the bf16 GEMM kernels on the shapes of BASELINE config 2 (B=4096, D=512).
usage: python scripts/gemm_bench.py [ablate bits ...]   (env knobs of gemm_bf16.hip apply)"""
import sys
sys.path.insert(0, '.')
import torch  # noqa: F401  (initialises HIP)
from mr_gan_amd import engine as E

B = 4096
SHAPES = [  # (name, op, m, n, k, nbatch, splits)
    ("fwd D1 2Bx512->1024", 0, B, 1024, 512, 2, 1),
    ("fwd D2 2Bx1024->512", 0, B, 512, 1024, 2, 1),
    ("fwd G3 2Bx512->512", 0, B, 512, 512, 2, 1),
    ("fwd D1 3Bx512->1024", 0, B, 1024, 512, 3, 1),
    ("fwd D1 relu+mask only", 3, B, 1024, 512, 3, 1),
    ("fwd D1 plain relu", 4, B, 1024, 512, 3, 1),
    ("fwd D2 3Bx1024->512", 0, B, 512, 1024, 3, 1),
    ("fwd D3 3Bx512->256", 0, B, 256, 512, 3, 1),
    ("fwd D4 3Bx256->256", 0, B, 256, 256, 3, 1),
    ("fwd G2 Bx512->512", 0, B, 512, 512, 1, 1),
    ("dx  D2 3B:512->1024", 1, B, 512, 1024, 3, 1),
    ("dx  D3 3B:256->512", 1, B, 256, 512, 3, 1),
    ("dx  G3 softplus'+h+cs", 5, B, 512, 512, 1, 1),
    ("dx  G2 linear+xhat", 6, B, 512, 512, 1, 1),
    ("dx  G2 linear+colsum", 7, B, 512, 512, 1, 1),
    ("dx  G2 linear", 8, B, 512, 512, 1, 1),
    ("dx  D1 B:1024->512 +cs", 7, B, 1024, 512, 1, 1),
    ("dw  D1 512x1024 /3B", 2, B, 1024, 512, 3, 8),
    ("dw  D3 512x256 /3B", 2, B, 256, 512, 3, 16),
    ("dw  D4 256x256 /3B", 2, B, 256, 256, 3, 16),
]
import os
lib = E.load_library(os.environ.get("MRGAN_BENCH_LIB"))      # e.g. mr_gan_amd/lib/libmrgan_hip_stamps.so (make STAMPS=1)
bits = [int(a) for a in sys.argv[1:]] or [0]
print("%-22s %8s " % ("shape", "GFLOP") + " ".join("abl%-3d us / TF   " % b for b in bits))
for name, op, m, n, k, nb, sp in SHAPES:
    gf = 2.0 * m * nb * n * k / 1e9
    row = "%-22s %8.2f " % (name, gf)
    for b in bits:
        us = E.debug_gemm_time(op, m, n, k, nb, sp, reps=30, ablate=b, kc_cfg=int(os.environ.get('KC_CFG', '-1')))
        row += "%8.1f /%6.0f   " % (us, gf / us * 1e3)
    print(row)
