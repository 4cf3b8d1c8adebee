"""Supervised neural-network baseline of the reference (mr_nn.py:69-119 and its --tables 2 4 loops, :128-168) on the HIP
engine: the discriminator stack of mr_gan.py (GaussianNoise 0.3 / 0.5, Dense 1000-500-250-250-250 relu, linear Dense(6))
trained on the labeled subset only with loss = 'mse' against the one-hot label and Keras' default Adam.

    python -m mr_gan_amd.mr_nn --tables 2 4 [-v] [--dtype bfloat16]

Every train_on_batch is one mrgan_sup_step (include/mrgan_abi.h): the same stage / dense / dX / dW / Adam kernels as the GAN
step, with the mse head.  There is no CPU path."""
import argparse
import sys

import numpy as np
import torch

from mr_gan_amd import engine as E
from mr_gan_amd.data import MATERIALS, select_labeled, standard_scale
from mr_gan_amd.model import glorot_uniform
from mr_gan_amd.mr_gan import dataset
from mr_gan_amd.mr_svm import baseline_tables

NN_LR, NN_BETA_1 = 0.001, 0.9              # Keras-2.0.9 Adam defaults (mr_nn.py:112 optimizer='adam')


class MRNN(object):
    """model.compile(loss='mse', optimizer='adam') + fit / evaluate of mr_nn.py:101-118 on one MI355X."""

    def __init__(self, input_dim, batch_size=20, dtype='float32', seed=None, device='cuda:0', num_classes=6,
                 d_hidden=(1000, 500, 250, 250, 250), lr=NN_LR, beta_1=NN_BETA_1, init_weights=True):
        self.input_dim, self.batch_size = int(input_dim), int(batch_size)
        self.seed = int(np.random.randint(1 << 31)) if seed is None else int(seed)      # mr_nn.py:71 is unseeded
        cfg = E.default_config(self.input_dim, self.batch_size)
        cfg.dtype = {'float32': E.F32, 'fp32': E.F32, 'bfloat16': E.BF16, 'bf16': E.BF16}[dtype]
        cfg.num_classes = num_classes
        for i, w in enumerate(d_hidden):
            cfg.d_hidden[i] = w
        cfg.lr, cfg.beta1 = lr, beta_1
        cfg.seed = self.seed
        self.engine = E.Engine(cfg, device)
        self.device = self.engine.device
        self.stream = torch.cuda.Stream(self.device)
        if init_weights:
            rng = np.random.RandomState(self.seed)
            ws = []
            for i in range(self.engine.num_tensors(E.NET_D)):
                shp = self.engine.full_shape(E.NET_D, i)
                ws.append(glorot_uniform(rng, shp[0], shp[1]) if len(shp) == 2 else np.zeros(shp, np.float32))
            self.engine.set_weights(E.NET_D, ws)

    def _dev(self, a, dtype=torch.float32):
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        return torch.from_numpy(np.ascontiguousarray(a)).to(device=self.device, dtype=dtype).contiguous()

    def train_on_batch(self, x, labels):
        """one batch of at most batch_size rows -> (mse, training error)"""
        B, n = self.batch_size, len(x)
        xb = torch.zeros((B, self.input_dim), device=self.device)
        yb = torch.full((B,), -1, dtype=torch.int32, device=self.device)
        xb[:n], yb[:n] = self._dev(x), self._dev(labels, torch.int32)
        with torch.cuda.stream(self.stream):
            self.stream.wait_stream(torch.cuda.current_stream(self.device))
            return self.engine.sup_step(E.Engine.sup_args(xb, yb, rows_valid=0 if n == B else n))

    def fit(self, x, y, epochs=100, verbose=0, rng=None):
        """Keras fit(batch_size, epochs, shuffle=True): a fresh permutation per epoch, batches in order, the last one short."""
        rng = rng or np.random.RandomState(self.seed)
        B, n = self.batch_size, len(x)
        nb = (n + B - 1) // B
        xd, yd = self._dev(x), self._dev(y, torch.int32)
        xs = torch.zeros((nb * B, self.input_dim), device=self.device)
        ys = torch.full((nb * B,), -1, dtype=torch.int32, device=self.device)
        hist = []
        with torch.cuda.stream(self.stream):
            self.stream.wait_stream(torch.cuda.current_stream(self.device))
            for ep in range(epochs):
                perm = torch.from_numpy(rng.permutation(n)).to(self.device)
                xs[:n], ys[:n] = xd[perm], yd[perm]
                last = ep == epochs - 1 or verbose
                out = None
                for b in range(nb):
                    short = n - b * B if (b + 1) * B > n else 0
                    out = self.engine.sup_step(E.Engine.sup_args(xs[b * B:(b + 1) * B], ys[b * B:(b + 1) * B], rows_valid=short),
                                               want_outputs=bool(last and b == nb - 1))
                if out is not None:
                    hist.append(dict(epoch=ep, loss=out[0], train_err=out[1]))
                    if verbose:
                        print('Epoch %d: loss %.5f, train err %.4f' % (ep + 1, out[0], out[1]))
            self.stream.synchronize()
        return hist

    def predict_logits(self, X):
        with torch.cuda.stream(self.stream):
            return self.engine.predict_logits(self._dev(X)).cpu().numpy()

    def evaluate(self, X, y):
        """1 - accuracy over the whole set (mr_nn.py:118)"""
        with torch.cuda.stream(self.stream):
            return self.engine.eval_error(self._dev(X), self._dev(y, torch.int32))


def mr_nn(X, y, percentlabeled=50, trainTestSets=None, verbose=False, epochs=100, batch_size=20, dtype='float32',
          seed=None, device='cuda:0'):
    from sklearn.model_selection import train_test_split
    from sklearn.utils import shuffle
    rs = np.random.RandomState(seed if seed is not None else np.random.randint(1 << 31))     # mr_nn.py:71 is unseeded
    test_ratio = 200 * len(MATERIALS)                              # mr_nn.py:74
    num_labeled_examples = int(10 * percentlabeled)                # mr_nn.py:75
    if trainTestSets is None:                                      # mr_nn.py:78-81
        X_train, X_test, y_train, y_test = train_test_split(X, y, test_size=test_ratio, stratify=y, random_state=rs)
    else:
        X_train, X_test, y_train, y_test = trainTestSets
    if verbose:
        print('Num of class examples in test set:', [int(np.sum(y_test == i)) for i in range(len(MATERIALS))])
        print('X_train:', np.shape(X_train), 'y_train:', np.shape(y_train), 'X_test:', np.shape(X_test), 'y_test:', np.shape(y_test))
    X_train, X_test = standard_scale(X_train, X_test)              # mr_nn.py:86-88
    X_train, y_train = shuffle(X_train, y_train, random_state=rs)  # mr_nn.py:91
    x_labeled, y_labeled, _ = select_labeled(X_train, y_train, num_labeled_examples)
    if verbose:
        print('x_labeled:', np.shape(x_labeled), 'y_labeled:', np.shape(y_labeled))
    model = MRNN(X_train.shape[1], batch_size=batch_size, dtype=dtype, seed=int(rs.randint(1 << 31)), device=device)
    model.fit(x_labeled, y_labeled, epochs=epochs, rng=rs)         # mr_nn.py:117
    testerror = model.evaluate(X_test, y_test)                     # mr_nn.py:118
    model.engine.close()
    return testerror


def main(argv=None, dataset_fn=dataset, fn=None):
    parser = argparse.ArgumentParser(description='Supervised NN baseline for material recognition on haptic data.')
    parser.add_argument('-t', '--tables', nargs='+', help='[Required] Tables to recompute', required=True)
    parser.add_argument('-v', '--verbose', help='Verbose', action='store_true')
    parser.add_argument('--dtype', default='float32', choices=['float32', 'bfloat16'])
    args = parser.parse_args(argv)
    if fn is None:
        def fn(X, y, **kw):
            return mr_nn(X, y, dtype=args.dtype, **kw)
    baseline_tables(args.tables, fn, dataset_fn, args.verbose)


if __name__ == '__main__':
    main()
