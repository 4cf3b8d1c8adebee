import sys; sys.path.insert(0, '.')
import numpy as np, torch
from mr_gan_amd import engine as E
from oracle import mrgan_oracle as O
from tests.helpers import Case, SEED
from tests.test_gpu_parity import _engine, _load, _t
B, D = 64, 32
case = Case(D=D, B=B, steps=1, device_z=True)
orc = O.MRGANOracle(case.g0, case.d0)
inp = case.disc_inputs(0, 0)
(ll, lu, err), gd, aux = orc.disc_grads(**inp)
l_lab = aux['l_lab']; labels = inp['labels']
row_loss = O.logsumexp(l_lab) - l_lab[np.arange(B), labels]
print("oracle loss_lab", ll, "shard sums/B", row_loss[:32].sum()/B, row_loss[32:].sum()/B)
lse_u = O.logsumexp(aux['l_unl']); lse_f = O.logsumexp(aux['l_fake'])
ru = 0.5*(O.softplus(lse_u)-lse_u) + 0.5*O.softplus(lse_f)
print("oracle loss_unl", lu, "shard sums/B", ru[:32].sum()/B, ru[32:].sum()/B)
flags = E.FLAG_FLAT_GRADS | E.FLAG_SYNC_STATS
ranks = [_engine(D, B // 2, 0, flags=flags, rank=r, world=2) for r in range(2)]
for e in ranks: _load(e, case)
h = B // 2
da = [E.Engine.disc_args(_t(case.x_lab[0][r*h:(r+1)*h]), _t(case.labels[0][r*h:(r+1)*h], torch.int32), _t(case.x_unl[0][r*h:(r+1)*h])) for r in range(2)]
for e, a in zip(ranks, da): e.disc_step(a, E.D_GEN, E.D_GEN, want_outputs=False)
views = [e.region(E.REGION_BN_STATS) for e in ranks]
print("bn stats rank sums", [float(v[:5].sum()) for v in views])
tot = views[0] + views[1]
for v in views: v.copy_(tot)
h1 = O.softplus(inp['z'] @ case.g0[0] + case.g0[1])
print("oracle sum h (first 5 cols)", h1[:, :5].sum(), "engine", float(tot[:5].sum()))
for e, a in zip(ranks, da): e.disc_step(a, E.D_MAIN, E.D_MAIN, want_outputs=False)
for r, e in enumerate(ranks):
    v = e.region(E.REGION_GRAD_D)
    print("rank", r, "tail", v[-4:].cpu().numpy())
# single-rank full batch for comparison
e1 = _engine(D, B, 0, flags=flags)
_load(e1, case)
a1 = E.Engine.disc_args(_t(case.x_lab[0]), _t(case.labels[0], torch.int32), _t(case.x_unl[0]))
e1.disc_step(a1, E.D_GEN, E.D_MAIN, want_outputs=False)
print("world=1 tail", e1.region(E.REGION_GRAD_D)[-4:].cpu().numpy())
