"""Per-tensor gradient error of the engine against the oracle mirror and the fp64 oracle (diagnostic).
usage: python scripts/parity_probe.py D B [dtype 0|1|2]      (2 = fp8)"""
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
from mr_gan_amd import engine as E
from oracle import mrgan_oracle as O
from tests.helpers import SEED, Case, frob_rel_err, cosine

D, B = int(sys.argv[1]), int(sys.argv[2])
dtype = int(sys.argv[3]) if len(sys.argv) > 3 else 1
DEV = "cuda:0"
t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(DEV, dt)
case = Case(D=D, B=B, steps=1)
mir = O.MRGANMirror(case.g0, case.d0, quantize={0: None, 1: 'bf16', 2: 'fp8'}[dtype])
orc = O.MRGANOracle(case.g0, case.d0)
(ll, lu, err), gd_m, _ = mir.disc_grads(**case.disc_inputs(0, 0))
_, gd_o, _ = orc.disc_grads(**case.disc_inputs(0, 0))
cfg = E.default_config(D, B)
cfg.dtype, cfg.seed, cfg.flags = dtype, SEED, E.FLAG_FLAT_GRADS
eng = E.Engine(cfg, DEV)
eng.set_weights(E.NET_G, [p.astype(np.float32) for p in case.g0])
eng.set_weights(E.NET_D, [p.astype(np.float32) for p in case.d0])
da = E.Engine.disc_args(t(case.x_lab[0]), t(case.labels[0], torch.int32), t(case.x_unl[0]), t(case.z1[0]))
eng.disc_step(da, E.D_GEN, E.D_MAIN, want_outputs=False)
print("D step: tensor  vs-mirror  vs-fp64   mirror-vs-fp64")
for i, (a, m, o) in enumerate(zip(eng.get_slot(E.NET_D, 2), gd_m, gd_o)):
    print("  dD%-2d %-12s %.2e  %.2e  %.2e" % (i, a.shape, frob_rel_err(a, m), frob_rel_err(a, o), frob_rel_err(m, o)))
out = eng.disc_step(da, E.D_ADAM, E.D_ADAM)
print("losses engine", out, "mirror", (ll, lu, err))
mir.adam.apply(mir.d, gd_m, 'd')
if dtype == 2:                      # what mrgan_set_weights does to the fp8 weight copies (two passes)
    for _ in range(2):
        mir._refresh_w8(); mir.slots.update()
    print("fp8 scales (mirror):", {k: float(v) for k, v in sorted(mir.slots.scale.items(), key=str)})
orc.d = [p.copy() for p in mir.d]
eng.set_weights(E.NET_D, [p.astype(np.float32) for p in mir.d])
loss, gg_m, _ = mir.gen_grads(**case.gen_inputs(0, 1))
_, gg_o, _ = orc.gen_grads(**case.gen_inputs(0, 1))
ga = E.Engine.gen_args(t(case.x_unl2[0]), t(case.z2[0]))
eng.gen_step(ga, E.G_GEN, E.G_TAIL, want_outputs=False)
for i, (a, m, o) in enumerate(zip(eng.get_slot(E.NET_G, 2), gg_m, gg_o)):
    print("  dG%-2d %-12s %.2e  %.2e  %.2e" % (i, a.shape, frob_rel_err(a, m), frob_rel_err(a, o), frob_rel_err(m, o)))
print("loss_gen engine", eng.gen_step(ga, E.G_ADAM, E.G_ADAM), "mirror", loss)
