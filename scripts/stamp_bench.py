import sys
sys.path.insert(0, '.')
import torch
from mr_gan_amd import engine as E
lib = E.load_library()
for bits in (0, 64, 128, 192):
    print("plain relu D1, ablate", bits, "us", round(E.debug_gemm_time(4, 4096, 1024, 512, 3, 1, reps=30, ablate=bits), 1))
