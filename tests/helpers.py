"""Shared builders for the parity tests: one seeded problem, run through the CPU oracle.

Conventions the HIP path and these tests share (documented in DESIGN.md):
  * layer noise and z come from the device generator restated in oracle.device_normal, keyed by
    (seed, site, segment, sub-step) where sub-step = Keras `iterations` before the call;
  * D-step segments: 0 labeled, 1 unlabeled, 2 generated;  G-step segments: 0 generated, 1 real.
"""
import numpy as np

from oracle import mrgan_oracle as O

SEED = 0x5EED5EED


def layer_dims(D, d_hidden=O.D_HIDDEN):
    return (D,) + tuple(d_hidden)


def noise_set(seed, seg, step, B, D, row0=0, dtype=np.float64, d_hidden=O.D_HIDDEN):
    dims = layer_dims(D, d_hidden)
    return [O.device_normal(seed, l, seg, step, B, dims[l], row0=row0, dtype=dtype) for l in range(5)]


def draw_z(seed, step, B, row0=0, dtype=np.float64, nz=O.NOISE_SIZE):
    return O.device_normal(seed, O.SITE_Z, 0, step, B, nz, row0=row0, dtype=dtype)


class Case(object):
    """A reproducible training problem + its oracle trajectory."""

    def __init__(self, D=16, B=50, steps=3, seed=7, noise_seed=SEED, dtype=np.float64, device_z=False,
                 d_hidden=O.D_HIDDEN, g_hidden=O.G_HIDDEN):
        rng = np.random.default_rng(seed)
        self.D, self.B, self.steps, self.noise_seed = D, B, steps, noise_seed
        self.d_hidden, self.g_hidden = tuple(d_hidden), tuple(g_hidden)
        g, d = O.init_params(D, seed=seed, dtype=dtype, g_hidden=self.g_hidden, d_hidden=self.d_hidden)
        # non-trivial biases / BN affine so every path carries signal
        g = [p + 0.05 * rng.standard_normal(p.shape).astype(dtype) for p in g]
        d = [p + 0.05 * rng.standard_normal(p.shape).astype(dtype) for p in d]
        self.g0, self.d0 = [p.copy() for p in g], [p.copy() for p in d]
        self.x_lab = rng.standard_normal((steps, B, D)).astype(np.float32)
        self.labels = rng.integers(0, O.NUM_CLASSES, (steps, B)).astype(np.int32)
        self.x_unl = rng.standard_normal((steps, B, D)).astype(np.float32)
        self.x_unl2 = rng.standard_normal((steps, B, D)).astype(np.float32)
        self.z1 = None if device_z else rng.standard_normal((steps, B, O.NOISE_SIZE)).astype(np.float32)
        self.z2 = None if device_z else rng.standard_normal((steps, B, O.NOISE_SIZE)).astype(np.float32)
        self.probe = rng.standard_normal((64, D)).astype(np.float32)
        self.dtype = dtype

    def disc_inputs(self, t, it, rows=None, row0=0):
        """numpy inputs of D sub-step t executed at Keras iteration `it` (optionally a row shard)."""
        B = self.B
        sl = slice(row0, row0 + (rows or B))
        nB = rows or B
        z = self.z1[t][sl] if self.z1 is not None else draw_z(self.noise_seed, it, nB, row0)
        return dict(x_lab=self.x_lab[t][sl].astype(self.dtype), labels=self.labels[t][sl],
                    x_unl=self.x_unl[t][sl].astype(self.dtype), z=np.asarray(z, self.dtype),
                    n_lab=noise_set(self.noise_seed, 0, it, nB, self.D, row0, self.dtype, self.d_hidden),
                    n_unl=noise_set(self.noise_seed, 1, it, nB, self.D, row0, self.dtype, self.d_hidden),
                    n_fake=noise_set(self.noise_seed, 2, it, nB, self.D, row0, self.dtype, self.d_hidden))

    def gen_inputs(self, t, it, rows=None, row0=0):
        B = self.B
        sl = slice(row0, row0 + (rows or B))
        nB = rows or B
        z = self.z2[t][sl] if self.z2 is not None else draw_z(self.noise_seed, it, nB, row0)
        return dict(x_unl=self.x_unl2[t][sl].astype(self.dtype), z=np.asarray(z, self.dtype),
                    n_fake=noise_set(self.noise_seed, 0, it, nB, self.D, row0, self.dtype, self.d_hidden),
                    n_real=noise_set(self.noise_seed, 1, it, nB, self.D, row0, self.dtype, self.d_hidden))

    def run_oracle(self, mirror=False, quantize=None):
        """trajectory through the fp64 restatement, or (mirror=True) through the engine-dataflow mirror with the engine's
        storage roundings (quantize = None | 'bf16' | 'fp8')"""
        orc = O.MRGANMirror(self.g0, self.d0, quantize=quantize) if mirror else O.MRGANOracle(self.g0, self.d0)
        out = dict(disc=[], gen=[], logits0=orc.predict_logits(self.probe.astype(self.dtype)))
        for t in range(self.steps):
            out['disc'].append(orc.disc_step(**self.disc_inputs(t, orc.adam.iterations)))
            out['gen'].append(orc.gen_step(**self.gen_inputs(t, orc.adam.iterations)))
        out['logits'] = orc.predict_logits(self.probe.astype(self.dtype))
        out['g'], out['d'] = orc.g, orc.d
        out['oracle'] = orc
        return out


def rel_err(a, ref):
    """scale-relative error: max |a - ref| / max |ref|"""
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return float(np.max(np.abs(a - ref)) / (np.max(np.abs(ref)) + 1e-300))


def update_rel_err(w, w_ref, w0):
    """error of a weight tensor relative to the size of the update that produced it"""
    w, w_ref, w0 = [np.asarray(x, np.float64) for x in (w, w_ref, w0)]
    return float(np.linalg.norm(w - w_ref) / (np.linalg.norm(w_ref - w0) + 1e-300))


def frob_rel_err(a, ref):
    """Frobenius-relative error ||a - ref|| / ||ref||"""
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return float(np.linalg.norm(a - ref) / (np.linalg.norm(ref) + 1e-300))


def cosine(a, ref):
    a, ref = np.asarray(a, np.float64).ravel(), np.asarray(ref, np.float64).ravel()
    return float(a @ ref / (np.linalg.norm(a) * np.linalg.norm(ref) + 1e-300))


def stub_training(job, datasets, device):
    """CPU stand-in for scheduler.train_job: a deterministic 'test error' from the job's rows (and the worker's pid, so
    tests can see that several processes took part)."""
    import os
    X, y = datasets[job['dataset']]
    tr, te = job['train_idx'], job['test_idx']
    val = float(np.mean(X[tr]) - np.mean(X[te]) + 0.01 * job.get('percentlabeled', 0) + np.mean(y[te]))
    if job.get('explode'):
        raise ValueError("stub failure requested")
    if job.get('sleep'):
        import time
        time.sleep(job['sleep'])
    return (val, os.getpid(), device)


def oracle_fit(g0, d0, x_labeled, y_labeled, x_train, x_test, y_test, batch, epochs, seed, rng_seed, dtype=np.float32, quantize=None, log=None):
    """The epoch loop of mr_gan.py:183-230 through the CPU oracle, driven by the SAME streams as MRGAN.fit on the engine:
    index streams from `rng` in the order MRGAN._fit draws them (mr_gan.py:189-195), z and layer noise from the restated
    device generator keyed by (seed, site, segment, Keras iteration).  Returns the final whole-test-set error (mr_gan.py:230)."""
    from threadpoolctl import threadpool_limits
    # batch-50 products: OpenBLAS with many threads is pathologically slow on them (X^T dY at 50 x 1200 x 1000: 19 ms with 8
    # threads, 0.5 ms with 4), so the loop runs on 4 BLAS threads
    with threadpool_limits(limits=4):
        return _oracle_fit(g0, d0, x_labeled, y_labeled, x_train, x_test, y_test, batch, epochs, seed, np.random.RandomState(rng_seed), dtype,
                           quantize, log)


def _oracle_fit(g0, d0, x_labeled, y_labeled, x_train, x_test, y_test, batch, epochs, seed, rng, dtype, quantize, log):
    orc = O.MRGANMirror(g0, d0, quantize=quantize) if quantize else O.MRGANOracle([p.astype(dtype) for p in g0], [p.astype(dtype) for p in d0])
    xl, xu = x_labeled.astype(dtype), x_train.astype(dtype)
    yl = np.asarray(y_labeled).astype(np.int64)
    n_train, n_lab, D = xu.shape[0], xl.shape[0], xu.shape[1]
    nb = n_train // batch
    it = 0
    for epoch in range(epochs):
        inds = O.tiled_permutation(rng.permutation, n_lab, n_train)
        unl = [rng.permutation(n_train) for _ in range(3)]
        for t in range(nb):
            sl = slice(t * batch, (t + 1) * batch)
            ns = lambda seg, k: noise_set(seed, seg, k, batch, D, 0, dtype)
            orc.disc_step(xl[inds[sl]], yl[inds[sl]], xu[unl[0][sl]], draw_z(seed, it, batch, dtype=dtype), ns(0, it), ns(1, it), ns(2, it))
            orc.gen_step(xu[unl[1][sl]], draw_z(seed, it + 1, batch, dtype=dtype), ns(0, it + 1), ns(1, it + 1))
            it += 2
        if log is not None:
            log.append(float(orc.test_error(x_test.astype(dtype), y_test)))
    return float(orc.test_error(x_test.astype(dtype), y_test))


def oracle_fit_job(kw):
    """oracle_fit in a worker process (multiprocessing 'spawn'): the worker imports numpy only -- with torch's and
    scikit-learn's OpenMP pools loaded beside OpenBLAS these batch-50 products run ~80x slower (120 ms per forward, measured)."""
    log = []
    err = oracle_fit(log=log, **kw)
    return err, log


def run_oracle_fits(jobs, workers=2):
    """[kwargs of oracle_fit] -> [(final error, per-epoch errors)], each in a clean worker process with 4 BLAS threads"""
    import multiprocessing as mp
    import os
    old = {k: os.environ.get(k) for k in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS")}
    os.environ["OPENBLAS_NUM_THREADS"] = "4"
    os.environ["OMP_NUM_THREADS"] = "4"
    try:
        with mp.get_context("spawn").Pool(min(workers, len(jobs))) as pool:
            asyncs = [pool.apply_async(oracle_fit_job, (j,)) for j in jobs]
            return asyncs, pool, [a.get(timeout=600) for a in asyncs]
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
