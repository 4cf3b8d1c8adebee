"""chain path vs per-layer path: per-tensor gradient differences (diagnostic). usage: python scripts/chain_probe.py D B"""
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
from mr_gan_amd import engine as E
from tests.helpers import SEED, Case, rel_err
D, B = int(sys.argv[1]), int(sys.argv[2])
DEV = "cuda:0"
t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(DEV, dt)
case = Case(D=D, B=B, steps=1, device_z=True)
res = []
for chain in (int(sys.argv[3]) if len(sys.argv) > 3 else 1, 0):
    cfg = E.default_config(D, B)
    cfg.dtype, cfg.seed, cfg.flags = 1, SEED, E.FLAG_FLAT_GRADS
    eng = E.Engine(cfg, DEV)
    eng.set_tuning(E.TUNE_CHAIN, chain)
    eng.set_weights(E.NET_G, [p.astype(np.float32) for p in case.g0])
    eng.set_weights(E.NET_D, [p.astype(np.float32) for p in case.d0])
    da = E.Engine.disc_args(t(case.x_lab[0]), t(case.labels[0], torch.int32), t(case.x_unl[0]))
    eng.disc_step(da, E.D_GEN, E.D_MAIN, want_outputs=False)
    gd = eng.get_slot(E.NET_D, 2)
    bufs = {(k, l): eng.debug_buffer(k, l).cpu().numpy() for k in (0, 1) for l in range(5)}
    out = eng.disc_step(da, E.D_ADAM, E.D_ADAM)
    eng.set_weights(E.NET_D, [p.astype(np.float32) for p in case.d0])      # same D weights for the G sub-step
    ga = E.Engine.gen_args(t(case.x_unl2[0]))
    eng.gen_step(ga, E.G_GEN, E.G_TAIL, want_outputs=False)
    gg = eng.get_slot(E.NET_G, 2)
    lg = eng.gen_step(ga, E.G_ADAM, E.G_ADAM)
    res.append((gd, out, gg, lg, bufs))
    eng.close()
(gd1, out1, gg1, lg1, b1), (gd0, out0, gg0, lg0, b0) = res
for key in sorted(b1):
    d = np.abs(b1[key][:, :B] - b0[key][:, :B])
    print('buf', key, 'max diff', d.max(), 'frac bad', (d > 0).mean(), 'first bad', np.argwhere(d > 0)[:3].tolist())
print("losses", out1, out0, "gen", lg1, lg0)
for i, (a, b) in enumerate(zip(gd1, gd0)):
    print("dD%d" % i, a.shape, rel_err(a, b))
for i, (a, b) in enumerate(zip(gg1, gg0)):
    print("dG%d" % i, a.shape, rel_err(a, b))
import torch as _t
from oracle.mrgan_oracle import bf16_round
d4 = b0[(1, 4)][0, :B, :250].astype(np.float64)
W = bf16_round(case.d0[8].astype(np.float64))      # W5 of D: [250, 250] (layer index 4)
legacy = b0[(1, 3)][0, :B, :250]
chain = b1[(1, 3)][0, :B, :250]
m = legacy != 0
print("legacy vs dY W^T :", np.abs(np.where(m, d4 @ W.T, 0) - legacy).max(), " legacy vs dY W:", np.abs(np.where(m, d4 @ W, 0) - legacy).max())
print("chain  vs dY W^T :", np.abs(np.where(m, d4 @ W.T, 0) - chain).max(), " chain vs dY W:", np.abs(np.where(m, d4 @ W, 0) - chain).max())
print("chain nonzero pattern equals legacy:", np.array_equal(chain != 0, m), " rows with any diff:", np.unique(np.argwhere(chain != legacy)[:, 0])[:20])
r = 1
print("row1 chain ", chain[r, :8]); print("row1 legacy", legacy[r, :8]); print("row1 dYW^T", (d4 @ W.T)[r, :8])
good = [r for r in range(128) if np.array_equal(chain[r], legacy[r])]
print("rows equal:", good)
full = (d4 @ W.T)
for r in (1, 2, 3, 5):
    # is the chain row a masked version of another row's product?
    cand = [q for q in range(256) if np.abs(np.where(m[r], full[q], 0) - chain[r]).max() < 2e-5]
    print("chain row", r, "matches product row(s)", cand)
