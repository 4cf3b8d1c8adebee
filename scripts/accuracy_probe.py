"""Accuracy parity on the MREO-shaped surrogate: HIP fp32 / HIP bf16 / CPU oracle driven by identical
index and noise streams.  usage: python scripts/accuracy_probe.py [epochs] [sep] [trials] [--no-oracle]
(trials = rows per object: 100 is the real data set's size, N = 7200)"""
import sys
import time
sys.path.insert(0, '.')
import numpy as np


def problem(n_lab, sep, trials):
    from sklearn.model_selection import StratifiedKFold
    from mr_gan_amd import select_labeled, standard_scale, synthetic_mreo
    X, y, _ = synthetic_mreo(sep=sep, trials=trials)
    tr, te = next(iter(StratifiedKFold(n_splits=6, shuffle=True, random_state=0).split(X, y)))
    Xtr, Xte = standard_scale(X[tr], X[te])
    ytr, yte = y[tr], y[te]
    perm = np.random.RandomState(1).permutation(len(ytr))
    Xtr, ytr = Xtr[perm], ytr[perm]
    xl, yl, _ = select_labeled(Xtr, ytr, n_lab)
    return Xtr, ytr, Xte, yte, xl, yl


if __name__ == "__main__":
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    sep = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    trials = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    with_oracle = "--no-oracle" not in sys.argv
    from mr_gan_amd import MRGAN
    from tests.helpers import run_oracle_fits
    seed = 1234
    res, jobs, keys = {}, [], []
    for n_lab in (max(2, trials // 2), 5 * trials):          # 50 and 500 per class at the real size
        Xtr, ytr, Xte, yte, xl, yl = problem(n_lab, sep, trials)
        for dt in ('float32', 'bfloat16'):
            m = MRGAN(Xtr.shape[1], batch_size=50, dtype=dt, seed=seed)
            if dt == 'float32':
                jobs.append(dict(g0=m.get_weights('generator'), d0=m.get_weights('discriminator'), x_labeled=xl, y_labeled=yl, x_train=Xtr,
                                 x_test=Xte, y_test=yte, batch=50, epochs=epochs, seed=seed, rng_seed=5))
                keys.append(n_lab)
            t0 = time.time()
            hist = m.fit(xl, yl, Xtr, epochs=epochs, validation_data=(Xte, yte), rng=np.random.RandomState(5))
            res[(n_lab, dt)] = (m.evaluate(Xte, yte), [h['test_err'] for h in hist][-5:], time.time() - t0)
            print('done', n_lab, dt, "%.4f" % res[(n_lab, dt)][0], "%.1fs" % res[(n_lab, dt)][2], flush=True)
            m.engine.close()
    if with_oracle:
        t0 = time.time()
        _, _, out = run_oracle_fits(jobs)
        for k, (e, log) in zip(keys, out):
            res[(k, 'oracle')] = (e, log[-5:], time.time() - t0)
    for k in sorted(res, key=str):
        print(k, "final err %.4f" % res[k][0], "last5", np.round(res[k][1], 4), "%.1fs" % res[k][2], flush=True)
