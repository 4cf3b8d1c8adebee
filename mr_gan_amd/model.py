"""Keras-shaped front end of the HIP training path.

The reference script exposes no class -- its model lives in local variables of mr_gan()
(mr_gan.py:109-171).  The fit / evaluate / predict shape mirrors how the same authors drive Keras
models elsewhere in the repository (mr_nn.py:117-118, others/mr_gan_autoencoder.py:125-139).
"""
import sys
import time

import numpy as np
import torch

from mr_gan_amd import engine as E
from mr_gan_amd.data import tiled_permutation


def glorot_uniform(rng, fan_in, fan_out):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=(fan_in, fan_out)).astype(np.float32)


class MRGAN(object):
    """Feature-matching semi-supervised GAN of mr_gan.py on one MI355X.

    dtype: 'float32' (fp32 MFMA: logits within 1e-3 of the reference arithmetic) or 'bfloat16'
    (bf16 MFMA, fp32 accumulate and fp32 master weights: the throughput mode).
    """

    def __init__(self, input_dim, batch_size=50, dtype='float32', seed=None, device='cuda:0', num_classes=6,
                 noise_size=100, g_hidden=(500, 500), d_hidden=(1000, 500, 250, 250, 250), lr=0.0006, beta_1=0.5,
                 unlabeled_weight=1.0, use_graph=True, rank=0, world=1, flags=0, init_weights=True):
        self.input_dim = int(input_dim)
        self.batch_size = int(batch_size)
        self.seed = int(np.random.randint(1 << 31)) if seed is None else int(seed)   # mr_gan.py:75 is unseeded
        cfg = E.default_config(self.input_dim, self.batch_size)
        cfg.dtype = {'float32': E.F32, 'fp32': E.F32, 'bfloat16': E.BF16, 'bf16': E.BF16, 'fp8': E.FP8, 'float8': E.FP8}[dtype]
        cfg.num_classes, cfg.noise_size = num_classes, noise_size
        cfg.g_hidden[0], cfg.g_hidden[1] = g_hidden
        for i, w in enumerate(d_hidden):
            cfg.d_hidden[i] = w
        cfg.lr, cfg.beta1, cfg.unlabeled_weight = lr, beta_1, unlabeled_weight
        cfg.seed = self.seed
        cfg.rank, cfg.world = rank, world
        cfg.flags = flags | (E.FLAG_GRAPH if use_graph else 0)
        self.cfg = cfg
        self.engine = E.Engine(cfg, device)
        self.device = self.engine.device
        # a private HIP stream: graph capture is not permitted on the legacy default stream
        self.stream = torch.cuda.Stream(self.device)
        self.iterations = 0
        self.history = []
        if init_weights:
            self.initialize(self.seed)

    # ---- weights ---------------------------------------------------------------------------------------
    def initialize(self, seed):
        """Keras defaults: Dense kernels glorot_uniform, biases zero, BN gamma one / beta zero."""
        rng = np.random.RandomState(seed)
        for net in (E.NET_G, E.NET_D):
            ws = []
            for i in range(self.engine.num_tensors(net)):
                shp = self.engine.full_shape(net, i)
                if len(shp) == 2:
                    ws.append(glorot_uniform(rng, shp[0], shp[1]))
                elif net == E.NET_G and i == 2:
                    ws.append(np.ones(shp, np.float32))
                else:
                    ws.append(np.zeros(shp, np.float32))
            self.engine.set_weights(net, ws)

    def get_weights(self, net='discriminator'):
        return self.engine.get_weights(E.NET_D if net.startswith('d') else E.NET_G)

    def set_weights(self, weights, net='discriminator'):
        self.engine.set_weights(E.NET_D if net.startswith('d') else E.NET_G, weights)

    # ---- the three compiled functions (mr_gan.py:169-171) on host arrays --------------------------------------
    def _dev(self, a, dtype=torch.float32):
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        return torch.from_numpy(np.ascontiguousarray(a)).to(device=self.device, dtype=dtype).contiguous()

    def _on_stream(self):
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        return torch.cuda.stream(self.stream)

    def train_batch_disc(self, x_lab, labels, x_unl, noise=None):
        with self._on_stream():
            return self._train_batch_disc(x_lab, labels, x_unl, noise)

    def train_batch_gen(self, x_unl, noise=None):
        with self._on_stream():
            return self._train_batch_gen(x_unl, noise)

    def test_batch(self, x, labels):
        with self._on_stream():
            return self.engine.eval_error(self._dev(x), self._dev(labels, torch.int32))

    def fit(self, *args, **kwargs):
        with self._on_stream():
            return self._fit(*args, **kwargs)

    def predict_logits(self, X):
        with self._on_stream():
            return self.engine.predict_logits(self._dev(X)).cpu().numpy()

    def evaluate(self, X, y):
        """test_batch([0, X, y]) over the whole set (mr_gan.py:230) -> error rate."""
        with self._on_stream():
            return self.engine.eval_error(self._dev(X), self._dev(y, torch.int32))

    def _train_batch_disc(self, x_lab, labels, x_unl, noise=None):
        a = E.Engine.disc_args(self._dev(x_lab), self._dev(labels, torch.int32), self._dev(x_unl),
                               None if noise is None else self._dev(noise))
        self._hold = a
        self.iterations += 1
        return self.engine.disc_step(a)

    def _train_batch_gen(self, x_unl, noise=None):
        a = E.Engine.gen_args(self._dev(x_unl), None if noise is None else self._dev(noise))
        self.iterations += 1
        return self.engine.gen_step(a)

    # ---- Keras-shaped API ---------------------------------------------------------------------------------------
    def _fit(self, x_labeled, y_labeled, x_unlabeled, epochs=100, batch_size=None, verbose=0, validation_data=None,
            x_unlabeled_pool=None, rng=None, z_source='device'):
        """The epoch loop of mr_gan.py:183-228.

        x_labeled / y_labeled: the class-sorted labeled subset (mr_gan.py:102-103);
        x_unlabeled: X_train (every training row is used as unlabeled data, mr_gan.py:193-194);
        x_unlabeled_pool: table-6 restricted pool (mr_gan.py:197-202), else None.
        The matrices are uploaded once and stay resident in HBM; per epoch only the permutation index
        streams (mr_gan.py:189-195) go to the device, and the per-batch scalars are accumulated on the
        device and read back once per epoch.
        z_source: 'device' draws the generator input on the GPU (the engine's counter-based generator);
        'host' draws it as the reference does, np.random.normal(0, 1, [B, noise]) twice per iteration (mr_gan.py:206, :212),
        from `rng`, and uploads one epoch's worth at a time."""
        if batch_size is not None and batch_size != self.batch_size:
            raise ValueError("batch_size is fixed at construction (%d)" % self.batch_size)
        rng = rng or np.random.RandomState(self.seed ^ 0x5bd1e995)
        B = self.batch_size
        xl = self._dev(x_labeled)
        xu = self._dev(x_unlabeled)
        yl = np.asarray(y_labeled).astype(np.int32)
        pool = self._dev(x_unlabeled_pool) if x_unlabeled_pool is not None else None
        n_train, n_lab = xu.shape[0], xl.shape[0]
        nb = n_train // B                                          # mr_gan.py:173 (remainder rows dropped)
        if nb < 1:
            raise ValueError("fewer training rows (%d) than one batch (%d)" % (n_train, B))
        xt = yt = None
        if validation_data is not None:
            xt, yt = self._dev(validation_data[0]), self._dev(validation_data[1], torch.int32)
        for epoch in range(1, epochs + 1):
            begin = time.time()
            inds = tiled_permutation(rng, n_lab, n_train)          # mr_gan.py:189
            if pool is None:
                unl = [rng.permutation(n_train).astype(np.int32) for _ in range(3)]      # :193-195 (third is unused)
                unl_src = xu
            else:
                unl = [tiled_permutation(rng, pool.shape[0], n_train) for _ in range(3)]  # :197-202
                unl_src = pool
            idx_lab = self._dev(inds, torch.int32)
            lab_stream = self._dev(yl[inds], torch.int32)
            idx_unl = self._dev(unl[0], torch.int32)
            idx_unl2 = self._dev(unl[1], torch.int32)
            z1 = z2 = None
            if z_source == 'host':
                zz = rng.normal(0.0, 1.0, size=(nb, 2, B, self.cfg.noise_size)).astype(np.float32)       # :206 then :212, per iteration
                z1, z2 = self._dev(zz[:, 0].reshape(nb * B, -1)), self._dev(zz[:, 1].reshape(nb * B, -1))
            elif z_source != 'device':
                raise ValueError("z_source must be 'device' or 'host'")
            dargs = E.Engine.disc_args(xl, lab_stream, unl_src, z1, idx_lab, idx_unl, stream_mode=1)
            gargs = E.Engine.gen_args(unl_src, z2, idx_unl2, stream_mode=1)
            self.engine.set_iterations(self.iterations, 0)
            for _ in range(nb):                                    # mr_gan.py:204-213
                self.engine.train_pair(dargs, gargs)
            self.iterations += 2 * nb
            m = self.engine.read_metrics(reset=True)               # one sync per epoch
            loss_lab, loss_unl, train_err, loss_gen = [v / nb for v in m[:4]]
            test_err = float('nan')
            if xt is not None:
                nte = (xt.shape[0] // B) * B                       # mr_gan.py:221-223: mean over whole test batches
                if nte > 0:
                    test_err = self.engine.eval_error(xt[:nte], yt[:nte])
            rec = dict(epoch=epoch, time=time.time() - begin, loss_lab=loss_lab, loss_unl=loss_unl, train_err=train_err,
                       loss_gen=loss_gen, test_err=test_err)
            self.history.append(rec)
            if verbose:
                # mr_gan.py:227
                print('Epoch %d, time = %ds, loss labeled = %.4f, loss unlabeled = %.4f, train error = %.4f, test error = %.4f'
                      % (epoch, rec['time'], loss_lab, loss_unl, train_err, test_err))
                sys.stdout.flush()
        return self.history

    def predict(self, X):
        return np.argmax(self.predict_logits(X), axis=1)
