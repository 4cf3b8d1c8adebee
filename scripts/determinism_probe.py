"""run the same (D, G) sub-steps on fresh handles several times and compare every gradient bit for bit (race detector)"""
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
from mr_gan_amd import engine as E
from tests.helpers import Case, SEED
D, B, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 4
case = Case(D=D, B=B, steps=1, device_z=True)
t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0", dt)
ref = None
for rep in range(reps):
    cfg = E.default_config(D, B); cfg.dtype = 1; cfg.seed = SEED; cfg.flags = E.FLAG_FLAT_GRADS
    eng = E.Engine(cfg, "cuda:0")
    eng.set_weights(E.NET_G, [p.astype(np.float32) for p in case.g0]); eng.set_weights(E.NET_D, [p.astype(np.float32) for p in case.d0])
    da = E.Engine.disc_args(t(case.x_lab[0]), t(case.labels[0], torch.int32), t(case.x_unl[0]))
    eng.disc_step(da, E.D_GEN, E.D_MAIN, want_outputs=False)
    gd = eng.get_slot(E.NET_D, 2)
    eng.disc_step(da, E.D_ADAM, E.D_ADAM)
    ga = E.Engine.gen_args(t(case.x_unl2[0]))
    eng.gen_step(ga, E.G_GEN, E.G_TAIL, want_outputs=False)
    gg = eng.get_slot(E.NET_G, 2)
    acts = [eng.debug_buffer(0, l, 2).cpu().numpy() for l in range(5)] + [eng.debug_buffer(1, l, 1).cpu().numpy() for l in range(5)] + [eng.debug_buffer(2, 0, 2).cpu().numpy()]
    eng.close()
    cur = gd + gg + acts
    if ref is None:
        ref = cur
    else:
        names = ["dD%d" % i for i in range(12)] + ["dG%d" % i for i in range(8)] + ["xin%d" % l for l in range(5)] + ["dpre%d" % l for l in range(5)] + ["feat"]
        bad = [(n, float(np.abs(a - b).max())) for n, a, b in zip(names, cur, ref) if not np.array_equal(a, b)]
        print("rep", rep, "differs from rep 0 in:", bad if bad else "nothing")
