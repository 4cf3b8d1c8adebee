"""Generates tests/golden/*.npz from the CPU oracle (fp64).  Run from the repo root:
    python tests/golden/make_golden.py
The reference holds no golden vectors for this path and cannot run here, so these fixtures
pin the build's own oracle: they guard against silent drift of the restatement and give the GPU tests a
reference that does not depend on numpy's RNG or BLAS at test time.
Inputs are regenerated from the seeds stored in the file; expected outputs are stored."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import mrgan_oracle as O  # noqa: E402
from tests.helpers import Case  # noqa: E402


def summarize(ws):
    return np.array([[w.sum(), (w * w).sum(), np.abs(w).max()] for w in ws])


def make(D, B, steps, seed, device_z):
    case = Case(D=D, B=B, steps=steps, seed=seed, device_z=device_z)
    ref = case.run_oracle()
    return dict(D=D, B=B, steps=steps, seed=seed, device_z=int(device_z), noise_seed=case.noise_seed,
                disc=np.array(ref['disc'], dtype=np.float64), gen=np.array(ref['gen'], dtype=np.float64),
                logits0=ref['logits0'], logits=ref['logits'],
                g_summary=summarize(ref['g']), d_summary=summarize(ref['d']),
                d_w6=ref['d'][10], d_b1=ref['d'][1], g_gamma=ref['g'][2], g_beta=ref['g'][3])


def tiling_cases():
    out = {}
    for ntr, nlab in ((6000, 480), (6000, 960), (7100, 6000)):
        rs = np.random.RandomState(1234)
        out['tiling_%d_%d' % (ntr, nlab)] = O.tiled_permutation(rs.permutation, nlab, ntr)
    return out


if __name__ == '__main__':
    here = os.path.dirname(os.path.abspath(__file__))
    np.savez_compressed(os.path.join(here, 'case_d16_b50.npz'), **make(16, 50, 3, 7, False))
    np.savez_compressed(os.path.join(here, 'case_d400_b50_devz.npz'), **make(400, 50, 2, 11, True))
    np.savez_compressed(os.path.join(here, 'tiling.npz'), **tiling_cases())
    print('golden fixtures written to', here)
