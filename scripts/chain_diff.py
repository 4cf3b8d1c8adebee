"""chain launches vs per-layer launches: per-tensor gradient differences (debug aid for gemm_chain.hip)"""
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
from mr_gan_amd import engine as E
from tests.helpers import Case, rel_err, SEED
D, B = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (400, 256)
case = Case(D=D, B=B, steps=1, device_z=True)
t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0", dt)
res = []
for chain in (1, 0):
    cfg = E.default_config(D, B); cfg.dtype = 1; cfg.seed = SEED; cfg.flags = E.FLAG_FLAT_GRADS
    eng = E.Engine(cfg, "cuda:0")
    eng.set_tuning(E.TUNE_CHAIN, chain)
    eng.set_weights(E.NET_G, [p.astype(np.float32) for p in case.g0]); eng.set_weights(E.NET_D, [p.astype(np.float32) for p in case.d0])
    da = E.Engine.disc_args(t(case.x_lab[0]), t(case.labels[0], torch.int32), t(case.x_unl[0]))
    eng.disc_step(da, E.D_GEN, E.D_MAIN, want_outputs=False)
    gd = eng.get_slot(E.NET_D, 2)
    acts = [eng.debug_buffer(0, l).cpu().numpy() for l in range(5)] + [eng.debug_buffer(1, l).cpu().numpy() for l in range(5)] + [eng.debug_buffer(2, 0).cpu().numpy()]
    out = eng.disc_step(da, E.D_ADAM, E.D_ADAM)
    ga = E.Engine.gen_args(t(case.x_unl2[0]))
    eng.gen_step(ga, E.G_GEN, E.G_TAIL, want_outputs=False)
    gg = eng.get_slot(E.NET_G, 2)
    lg = eng.gen_step(ga, E.G_ADAM, E.G_ADAM)
    res.append((gd, out, gg, lg, acts))
    eng.close()
(gd1, o1, gg1, l1, a1), (gd0, o0, gg0, l0, a0) = res
print("D losses", o1, o0, " G loss", l1, l0)
for i, (a, b) in enumerate(zip(gd1, gd0)):
    print("dD%-2d rel_err %.2e" % (i, rel_err(a, b)))
for i, (a, b) in enumerate(zip(gg1, gg0)):
    print("dG%-2d rel_err %.2e" % (i, rel_err(a, b)))
names = ["xin%d" % l for l in range(5)] + ["dpre%d" % l for l in range(5)] + ["feat"]
for n, a, b in zip(names, a1, a0):
    d = np.abs(a[:, :B] - b[:, :B])
    print("%-6s max|diff| %.3e  (max|ref| %.3e)  rows with diff per segment: %s" % (n, d.max(), np.abs(b[:, :B]).max(), [int((d[s].max(axis=1) > 0).sum()) for s in range(3)]))
