"""per-phase cycles of the chain launches of one bench-sized step (diagnostic build: make STAMPS=1)"""
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
from mr_gan_amd import engine as E
E.load_library("mr_gan_amd/lib/libmrgan_hip_stamps.so")
B, D = 4096, 512
cfg = E.default_config(D, B)
cfg.dtype, cfg.seed = 1, 1
eng = E.Engine(cfg, "cuda:0")
import os
if os.environ.get("CHAIN_ABLATE"):
    eng.debug_ablate(int(os.environ["CHAIN_ABLATE"]))      # CH_ABL_* bits (chain.h): timing experiments, results wrong
rs = np.random.RandomState(0)
for net in (E.NET_G, E.NET_D):
    ws = []
    for i in range(eng.num_tensors(net)):
        shp = eng.full_shape(net, i)
        ws.append(rs.uniform(-0.05, 0.05, size=shp).astype(np.float32) if len(shp) == 2 else np.zeros(shp, np.float32))
    eng.set_weights(net, ws)
t = lambda a, dt=torch.float32: torch.from_numpy(a).to("cuda:0", dt)
x = t(rs.randn(B, D).astype(np.float32)); y = t(rs.randint(0, 6, B).astype(np.int32), torch.int32)
for it in range(2):
    print("--- step", it, file=sys.stderr)
    eng.disc_step(E.Engine.disc_args(x, y, x))
    eng.gen_step(E.Engine.gen_args(x))
