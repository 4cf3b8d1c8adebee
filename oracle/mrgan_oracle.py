"""CPU oracle for the mr_gan.py training path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-numpy restatement of the arithmetic that the reference builds
through Keras 2.0.9 / Theano 0.9.0 in /root/reference/mr_gan.py:109-171 and drives from
the loop at mr_gan.py:183-230.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it; the product path (mr_gan_amd/) never does.

PARITY UNPINNED: the reference ships no tests, golden vectors, saved weights or data, and
cannot be parsed or imported in this container (Python-2 source, Keras/Theano absent; see
SURVEY.md section 8c).  The semantics below follow the reference source line by line plus the
published Keras-2.0.9/Theano-0.9 behaviour recorded in SURVEY.md rows A2, A3, A5, A8; the
only independent pins are (i) torch.autograd fp64 for every gradient
(tests/test_oracle.py) and (ii) sklearn for the data prologue.

Everything random is an explicit input (weights, z, layer noise), exactly as SURVEY 8c
defines "identical seeds/inputs".  The noise generator restated at the bottom
(counter hash + Box-Muller) is the *build's* device generator, not the reference's
MRG31k3p stream, which cannot be reproduced.
"""
import numpy as np

# mr_gan.py:77-79, :111-128, :165 -- literals of the reference
NOISE_SIZE = 100
G_HIDDEN = (500, 500)
D_HIDDEN = (1000, 500, 250, 250, 250)
D_SIGMAS = (0.3, 0.5, 0.5, 0.5, 0.5)       # GaussianNoise before dense 1..5 (mr_gan.py:118-126)
NUM_CLASSES = 6
BN_EPS = 2e-5                               # mr_gan.py:112
ADAM_LR, ADAM_B1, ADAM_B2, ADAM_EPS = 0.0006, 0.5, 0.999, 1e-8   # mr_gan.py:165 + Keras defaults
UNLABELED_WEIGHT = 1.0                      # mr_gan.py:79
NN_ADAM_LR, NN_ADAM_B1 = 0.001, 0.9         # mr_nn.py:112 optimizer='adam': Keras-2.0.9 defaults


# ----------------------------------------------------------------------------------------
# elementary functions (Keras backend / Theano semantics)
# ----------------------------------------------------------------------------------------
def softplus(x):
    # T.nnet.softplus == log1p(exp(x)), evaluated stably
    return np.logaddexp(0.0, x).astype(x.dtype, copy=False)


def sigmoid(x):
    out = np.empty_like(x)
    pos = x >= 0
    out[pos] = 1.0 / (1.0 + np.exp(-x[pos]))
    ex = np.exp(x[~pos])
    out[~pos] = ex / (1.0 + ex)
    return out


def relu(x):
    # T.nnet.relu(x) = 0.5*(x+|x|)
    return np.maximum(x, 0)


def logsumexp(x, axis=1):
    # K.logsumexp (Theano backend) is max-shifted
    m = np.max(x, axis=axis, keepdims=True)
    return (m + np.log(np.sum(np.exp(x - m), axis=axis, keepdims=True))).squeeze(axis)


# ----------------------------------------------------------------------------------------
# parameters (Keras tensor order: SURVEY rows A2, A3)
# ----------------------------------------------------------------------------------------
def glorot_uniform(rng, fan_in, fan_out, dtype):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=(fan_in, fan_out)).astype(dtype)


def g_shapes(D, nz=NOISE_SIZE, hidden=G_HIDDEN):
    h1, h2 = hidden
    return [(nz, h1), (h1,), (h1,), (h1,), (h1, h2), (h2,), (h2, D), (D,)]


def d_shapes(D, hidden=D_HIDDEN, K=NUM_CLASSES):
    dims = (D,) + tuple(hidden) + (K,)
    out = []
    for i in range(len(dims) - 1):
        out += [(dims[i], dims[i + 1]), (dims[i + 1],)]
    return out


def init_params(D, seed=0, dtype=np.float64, nz=NOISE_SIZE, g_hidden=G_HIDDEN, d_hidden=D_HIDDEN,
                K=NUM_CLASSES):
    """Keras defaults: Dense kernel glorot_uniform, bias zeros; BN gamma ones, beta zeros."""
    rng = np.random.default_rng(seed)
    g = []
    for i, shp in enumerate(g_shapes(D, nz, g_hidden)):
        if len(shp) == 2:
            g.append(glorot_uniform(rng, shp[0], shp[1], dtype))
        elif i == 2:                       # gamma
            g.append(np.ones(shp, dtype))
        else:
            g.append(np.zeros(shp, dtype))
    d = []
    for shp in d_shapes(D, d_hidden, K):
        d.append(glorot_uniform(rng, shp[0], shp[1], dtype) if len(shp) == 2 else np.zeros(shp, dtype))
    return g, d


# ----------------------------------------------------------------------------------------
# generator  (mr_gan.py:110-114)
# ----------------------------------------------------------------------------------------
def gen_forward(g, z):
    W1, b1, gamma, beta, W2, b2, W3, b3 = g
    pre1 = z @ W1 + b1
    h1 = softplus(pre1)
    # BatchNormalization(epsilon=2e-5) at learning phase 1: batch mean, *biased* batch variance
    mu = h1.mean(axis=0)
    var = ((h1 - mu) ** 2).mean(axis=0)
    rstd = 1.0 / np.sqrt(var + BN_EPS)
    xhat = (h1 - mu) * rstd
    hbn = gamma * xhat + beta
    pre2 = hbn @ W2 + b2
    h2 = softplus(pre2)
    x = h2 @ W3 + b3
    cache = dict(z=z, pre1=pre1, h1=h1, mu=mu, var=var, rstd=rstd, xhat=xhat, hbn=hbn, pre2=pre2, h2=h2)
    return x, cache


def gen_backward(g, cache, dx):
    """dx = dLoss/d(generator output) -> grads for the 8 generator tensors (Keras order)."""
    W1, b1, gamma, beta, W2, b2, W3, b3 = g
    B = dx.shape[0]
    dW3 = cache['h2'].T @ dx
    db3 = dx.sum(axis=0)
    dh2 = dx @ W3.T
    dpre2 = dh2 * sigmoid(cache['pre2'])
    dW2 = cache['hbn'].T @ dpre2
    db2 = dpre2.sum(axis=0)
    dhbn = dpre2 @ W2.T
    dgamma = (dhbn * cache['xhat']).sum(axis=0)
    dbeta = dhbn.sum(axis=0)
    dh1 = (gamma * cache['rstd'] / B) * (B * dhbn - dbeta - cache['xhat'] * dgamma)
    dpre1 = dh1 * sigmoid(cache['pre1'])
    dW1 = cache['z'].T @ dpre1
    db1 = dpre1.sum(axis=0)
    return [dW1, db1, dgamma, dbeta, dW2, db2, dW3, db3]


# ----------------------------------------------------------------------------------------
# discriminator  (mr_gan.py:117-128; mid_output = first five dense layers :133)
# ----------------------------------------------------------------------------------------
def disc_forward(d, x, noise=None, sigmas=D_SIGMAS):
    """noise: None (learning phase 0) or list of 5 standard-normal arrays n_l shaped like the
    input of dense l (GaussianNoise adds sigma_l * n_l).  Returns logits, features, cache."""
    nl = len(d) // 2
    a = x
    ins, pres = [], []
    for l in range(nl - 1):
        if noise is not None:
            a = a + np.asarray(sigmas[l], dtype=a.dtype) * noise[l]
        ins.append(a)
        pre = a @ d[2 * l] + d[2 * l + 1]
        pres.append(pre)
        a = relu(pre)
    feat = a                                  # disc_mid_output: post-ReLU, no noise after it
    logits = feat @ d[-2] + d[-1]
    return logits, feat, dict(ins=ins, pres=pres, feat=feat)


def disc_backward(d, cache, dlogits=None, dfeat=None, want_param_grads=True):
    """Backprop through the discriminator.  Either dlogits (D-step) or dfeat (G-step, through
    mid_output only).  Returns (grads list or None, dLoss/dx)."""
    nl = len(d) // 2
    grads = [None] * len(d)
    if dlogits is not None:
        if want_param_grads:
            grads[-2] = cache['feat'].T @ dlogits
            grads[-1] = dlogits.sum(axis=0)
        da = dlogits @ d[-2].T
    else:
        da = dfeat
    for l in range(nl - 2, -1, -1):
        dpre = da * (cache['pres'][l] > 0)
        if want_param_grads:
            grads[2 * l] = cache['ins'][l].T @ dpre
            grads[2 * l + 1] = dpre.sum(axis=0)
        da = dpre @ d[2 * l].T                # additive noise: d(in)/d(prev out) = identity
    return (grads if want_param_grads else None), da


# ----------------------------------------------------------------------------------------
# losses (mr_gan.py:146-154, :161-162) with closed-form gradients
# ----------------------------------------------------------------------------------------
def disc_losses(l_lab, labels, l_unl, l_fake):
    B = l_lab.shape[0]
    lse_lab = logsumexp(l_lab)
    loss_lab = -np.mean(l_lab[np.arange(B), labels]) + np.mean(lse_lab)
    lse_unl = logsumexp(l_unl)
    lse_fake = logsumexp(l_fake)
    loss_unl = (-0.5 * np.mean(lse_unl) + 0.5 * np.mean(softplus(lse_unl))
                + 0.5 * np.mean(softplus(lse_fake)))
    train_err = np.mean(np.argmax(l_lab, axis=1) != labels)
    return loss_lab, loss_unl, train_err


def disc_loss_grads(l_lab, labels, l_unl, l_fake, unlabeled_weight=UNLABELED_WEIGHT):
    B = l_lab.shape[0]
    dt = l_lab.dtype

    def softmax(l):
        e = np.exp(l - l.max(axis=1, keepdims=True))
        return e / e.sum(axis=1, keepdims=True)
    onehot = np.zeros_like(l_lab)
    onehot[np.arange(B), labels] = 1
    d_lab = (softmax(l_lab) - onehot) / B
    s_unl = sigmoid(logsumexp(l_unl))[:, None]
    d_unl = (0.5 / l_unl.shape[0]) * softmax(l_unl) * (s_unl - 1.0) * unlabeled_weight
    s_fake = sigmoid(logsumexp(l_fake))[:, None]
    d_fake = (0.5 / l_fake.shape[0]) * softmax(l_fake) * s_fake * unlabeled_weight
    return d_lab.astype(dt), d_unl.astype(dt), d_fake.astype(dt)


def mse_loss_grad(logits, labels):
    """Keras 'mse' (losses.py: K.mean(K.square(y_pred - y_true), axis=-1), then the batch mean) on the linear outputs against
    np_utils.to_categorical(labels) (mr_nn.py:99, :112); accuracy metric = argmax match (mr_nn.py:118)."""
    B, K = logits.shape
    y = np.zeros_like(logits)
    y[np.arange(B), labels] = 1.0
    d = logits - y
    loss = np.mean(d * d)
    err = np.mean(np.argmax(logits, axis=1) != labels)
    return loss, err, (2.0 / (B * K)) * d


def fm_loss(f_fake, f_real):
    mom_gen = f_fake.mean(axis=0)
    mom_real = f_real.mean(axis=0)
    return np.mean((mom_gen - mom_real) ** 2)


def fm_loss_grad(f_fake, f_real):
    B, J = f_fake.shape
    diff = f_fake.mean(axis=0) - f_real.mean(axis=0)
    return np.broadcast_to((2.0 / (J * B)) * diff, f_fake.shape).astype(f_fake.dtype)


# ----------------------------------------------------------------------------------------
# Keras-2.0.9 Adam with ONE shared iteration counter for both update lists
# ----------------------------------------------------------------------------------------
class Adam(object):
    def __init__(self, g, d, lr=ADAM_LR, b1=ADAM_B1, b2=ADAM_B2, eps=ADAM_EPS):
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.iterations = 0
        self.mg = [np.zeros_like(p) for p in g]
        self.vg = [np.zeros_like(p) for p in g]
        self.md = [np.zeros_like(p) for p in d]
        self.vd = [np.zeros_like(p) for p in d]

    def lr_t(self):
        t = self.iterations + 1
        return self.lr * np.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t)

    def apply(self, params, grads, which):
        ms, vs = (self.mg, self.vg) if which == 'g' else (self.md, self.vd)
        lr_t = self.lr_t()
        for i, (p, g) in enumerate(zip(params, grads)):
            dt = p.dtype
            g = np.asarray(g, dtype=dt)
            # m = b1 m + (1 - b1) g ; v = b2 v + (1 - b2) g^2 ; p -= lr_t m / (sqrt(v) + eps)   (in place: no temporaries)
            ms[i] *= dt.type(self.b1); ms[i] += dt.type(1.0 - self.b1) * g
            vs[i] *= dt.type(self.b2); vs[i] += dt.type(1.0 - self.b2) * g * g
            den = np.sqrt(vs[i]); den += dt.type(self.eps)
            np.divide(ms[i], den, out=den); den *= dt.type(lr_t)
            params[i] = p - den
        self.iterations += 1


# ----------------------------------------------------------------------------------------
# the three compiled functions of mr_gan.py:169-171
# ----------------------------------------------------------------------------------------
class MRGANOracle(object):
    """State = Keras shared variables (weights, Adam slots, iteration counter)."""

    def __init__(self, g, d, sigmas=D_SIGMAS, **adam):
        self.g = [np.array(p) for p in g]
        self.d = [np.array(p) for p in d]
        self.adam = Adam(self.g, self.d, **adam)
        self.sigmas = sigmas

    # train_batch_disc([1, x_lab, labels, x_unl, noise]) -> [loss_lab, loss_unl, train_err]
    def disc_grads(self, x_lab, labels, x_unl, z, n_lab, n_unl, n_fake):
        x_fake, _ = gen_forward(self.g, z)
        l_lab, _, c_lab = disc_forward(self.d, x_lab, n_lab, self.sigmas)
        l_unl, _, c_unl = disc_forward(self.d, x_unl, n_unl, self.sigmas)
        l_fake, _, c_fake = disc_forward(self.d, x_fake, n_fake, self.sigmas)
        out = disc_losses(l_lab, labels, l_unl, l_fake)
        dl_lab, dl_unl, dl_fake = disc_loss_grads(l_lab, labels, l_unl, l_fake)
        grads = None
        for c, dl in ((c_lab, dl_lab), (c_unl, dl_unl), (c_fake, dl_fake)):
            g_, _ = disc_backward(self.d, c, dlogits=dl)
            grads = g_ if grads is None else [a + b for a, b in zip(grads, g_)]
        aux = dict(l_lab=l_lab, l_unl=l_unl, l_fake=l_fake, x_fake=x_fake)
        return out, grads, aux

    def disc_step(self, x_lab, labels, x_unl, z, n_lab, n_unl, n_fake):
        out, grads, _ = self.disc_grads(x_lab, labels, x_unl, z, n_lab, n_unl, n_fake)
        self.adam.apply(self.d, grads, 'd')
        return out

    # train_batch_gen([1, x_unl, noise]) -> loss_gen
    def gen_grads(self, x_unl, z, n_fake, n_real):
        x_fake, gc = gen_forward(self.g, z)
        _, f_fake, c_fake = disc_forward(self.d, x_fake, n_fake, self.sigmas)
        _, f_real, _ = disc_forward(self.d, x_unl, n_real, self.sigmas)
        loss = fm_loss(f_fake, f_real)
        df = fm_loss_grad(f_fake, f_real)
        _, dx = disc_backward(self.d, c_fake, dfeat=df, want_param_grads=False)
        grads = gen_backward(self.g, gc, dx)
        return loss, grads, dict(f_fake=f_fake, f_real=f_real, x_fake=x_fake, dx=dx)

    def gen_step(self, x_unl, z, n_fake, n_real):
        loss, grads, _ = self.gen_grads(x_unl, z, n_fake, n_real)
        self.adam.apply(self.g, grads, 'g')
        return loss

    # NN baseline (mr_nn.py:101-118): model.fit == train_on_batch of the same stack, loss='mse' against the one-hot label,
    # optimizer='adam' (Keras defaults: construct with lr=NN_ADAM_LR, b1=NN_ADAM_B1) -> [mse, training error]
    def sup_grads(self, x, labels, noise):
        logits, _, c = disc_forward(self.d, x, noise, self.sigmas)
        loss, err, dl = mse_loss_grad(logits, labels)
        grads, _ = disc_backward(self.d, c, dlogits=dl)
        return (loss, err), grads, dict(logits=logits)

    def sup_step(self, x, labels, noise):
        out, grads, _ = self.sup_grads(x, labels, noise)
        self.adam.apply(self.d, grads, 'd')
        return out

    # test_batch([0, x, labels]) -> err   (noise layers are identity at phase 0)
    def predict_logits(self, x):
        return disc_forward(self.d, x, None)[0]

    def test_error(self, x, labels):
        return np.mean(np.argmax(self.predict_logits(x), axis=1) != labels)


# ----------------------------------------------------------------------------------------
# Device-dataflow mirror: the SAME algebra as MRGANOracle, evaluated in the order and with the storage roundings of
# the HIP engine (mr_gan_amd/csrc/engine.hip), so that a reduced-precision engine can be held to a tight tolerance.
#   quantize=None   : no rounding anywhere -> must agree with MRGANOracle to fp64 round-off (tests/test_oracle.py),
#                     which pins the mirror's structure to the autograd-checked restatement.
#   quantize='fp8'  : the bf16 dataflow, with the operands of the discriminator's dense products (forward, dX, dW)
#                     additionally rounded to OCP fp8 (e4m3 activations / weights, e5m2 gradients) under delayed per-tensor
#                     power-of-two scales, as gemm_fp8.hip stores them: fp8(v * scale) from the accumulator of the producing
#                     product / loss head / feature-matching kernel, fp8(bf16(v) * scale) where a bf16 tensor is converted
#                     (xin_0, weights).
#   quantize='bf16' : every tensor the engine STORES as bf16 is rounded (RNE) where the engine rounds it: GEMM weight
#                     copies, z, the noisy layer inputs, h1 / BN(h1) / h2, every dpre / dX activation.  Batch
#                     statistics, bias gradients and column sums come from the unrounded fp32 values, as on the device;
#                     the 250x6 loss head uses the fp32 master W6.
# ----------------------------------------------------------------------------------------
def bf16_round(x):
    """round-to-nearest-even to bfloat16, returned in x's dtype"""
    x = np.asarray(x)
    f = np.ascontiguousarray(x, dtype=np.float32)
    u = f.view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)).view(np.float32)
    return r.astype(x.dtype if x.dtype.kind == 'f' else np.float32)


def _ident(x):
    return x


FP8_FORMATS = {'e4m3': (3, -6, 448.0), 'e5m2': (2, -14, 57344.0)}      # OCP: mantissa bits, smallest normal exponent, largest finite
FP8_TARGETS = {'e4m3': 224.0, 'e5m2': 28672.0}                          # engine.hip FP8_TARGET_*: half the largest finite value
FP8_DRY_PASSES = 5                                                      # engine.hip FP8_DRY_PASSES


def fp8_round(x, fmt):
    """round to nearest even onto the OCP fp8 grid `fmt` (subnormals included), saturating -- what v_cvt_pk_fp8_f32 /
    v_cvt_pk_bf8_f32 do to the clamped input (gemm_fp8.hip pack4)"""
    mant, emin, lim = FP8_FORMATS[fmt]
    x = np.clip(np.asarray(x, dtype=np.float64), -lim, lim)
    _, e = np.frexp(np.abs(x))                        # |x| = m 2^e, m in [0.5, 1)
    quantum = np.exp2(np.maximum(e - 1, emin) - mant)
    return np.rint(x / quantum) * quantum


class Fp8Slots(object):
    """delayed per-tensor power-of-two scales (gemm.h Fp8Slot, fp8_update_scales_kernel)"""

    def __init__(self):
        self.scale, self.amax, self.fmt = {}, {}, {}

    def quant(self, v, key, fmt):
        """v -> dequantised fp8(v * scale) / scale; records max |v| (as fp32) for the next pass's scale.  v is bf16-valued where
        the engine quantises a stored bf16 tensor (xin_0, the bf16 weight copies) and the unrounded result where a product's
        epilogue, the loss head or the feature-matching kernel packs its fp32 value"""
        self.fmt[key] = fmt
        sc = self.scale.get(key, np.float32(1.0))
        self.amax[key] = max(self.amax.get(key, np.float32(0.0)), np.float32(np.abs(v).max() if v.size else 0.0))
        return fp8_round(np.asarray(v, np.float32).astype(np.float64) * float(sc), fmt) / float(sc)

    def update(self):
        for key, a in self.amax.items():
            if a > 0:
                ratio = np.float32(FP8_TARGETS[self.fmt[key]]) / np.float32(a)
                _, e = np.frexp(ratio)
                self.scale[key] = np.float32(2.0) ** int(np.clip(e - 1, -100, 100))
            self.amax[key] = np.float32(0.0)


class MRGANMirror(object):
    """train_batch_disc / train_batch_gen in the engine's dataflow.  Same call signatures as MRGANOracle."""

    def __init__(self, g, d, quantize=None, sigmas=D_SIGMAS, unlabeled_weight=UNLABELED_WEIGHT, **adam):
        self.g = [np.array(p) for p in g]
        self.d = [np.array(p) for p in d]
        self.adam = Adam(self.g, self.d, **adam)
        self.sigmas = sigmas
        self.uw = unlabeled_weight
        self.quantize = quantize
        self.q = {None: _ident, 'bf16': bf16_round, 'fp8': bf16_round}[quantize]
        # fp8: the discriminator's dense products take e4m3 activations / weights and e5m2 gradients (gemm_fp8.hip); the
        # generator, the loss head and predict_logits stay in the bf16 dataflow
        self.fp8 = quantize == 'fp8'
        if self.fp8:
            self.slots = Fp8Slots()
            self.cal = [False, False]
            for _ in range(2):                       # mrgan_set_weights: the first pass only measures max |w|
                self._refresh_w8()
                self._refresh_gw8()
                self.slots.update()

    # ---- fp8 mode ----------------------------------------------------------------------------------------
    def _refresh_w8(self):
        nl = len(self.d) // 2
        self.w8 = [self.slots.quant(bf16_round(self.d[2 * l]), ('w', l), 'e4m3') for l in range(nl - 1)]

    def _refresh_gw8(self):
        """the generator's second dense layer (the one wide product of the generator) also runs in fp8"""
        self.gw8 = self.slots.quant(bf16_round(self.g[4]), ('gw',), 'e4m3')

    def _disc_fwd8(self, xin0, noise, kind):
        """one segment through dense 1..5 with e4m3 operands; xin0 = the bf16 noisy input rows"""
        q, sl = self.q, self.slots
        nl = len(self.d) // 2
        xin, masks = [sl.quant(xin0, ('x', kind, 0), 'e4m3')], []
        a = None
        for l in range(nl - 1):
            a = relu(xin[l] @ self.w8[l] + self.d[2 * l + 1])
            masks.append(a > 0)
            if l < nl - 2:
                # packed straight from the fp32 accumulator by the forward epilogue: one rounding (no bf16 in between)
                xin.append(sl.quant(a + np.asarray(self.sigmas[l + 1], a.dtype) * noise[l + 1], ('x', kind, l + 1), 'e4m3'))
        return dict(xin=xin, masks=masks, feat=a, feat_q=q(a))

    def _disc_bwd8(self, c, dpre_top, kind, to_input=False):
        """dX chain with e5m2 gradients; returns the dequantised g8[l], the bias sums, and (to_input) d loss / d x"""
        q, sl = self.q, self.slots
        nl = len(self.d) // 2
        dpre, db = [None] * (nl - 1), [None] * (nl - 1)
        dpre[nl - 2] = sl.quant(dpre_top, ('g', kind, nl - 2), 'e5m2')
        for l in range(nl - 2, 0, -1):
            v = (dpre[l] @ self.w8[l].T) * c['masks'][l - 1]
            db[l - 1] = v.sum(axis=0)
            dpre[l - 1] = sl.quant(v, ('g', kind, l - 1), 'e5m2')          # from the dX epilogue's accumulator, one rounding
        dx = dpre[0] @ self.w8[0].T if to_input else None
        return dpre, db, dx

    def _disc_grads8(self, x_lab, labels, x_unl, z, n_lab, n_unl, n_fake):
        q = self.q
        nl = len(self.d) // 2
        xf, _ = self._gen_fwd(z, n_fake[0])
        segs = [self._disc_fwd8(self._stage(x_lab, n_lab[0]), n_lab, 0), self._disc_fwd8(self._stage(x_unl, n_unl[0]), n_unl, 0),
                self._disc_fwd8(xf, n_fake, 0)]
        W6, b6 = self.d[-2], self.d[-1]
        logits = [c['feat_q'] @ W6 + b6 for c in segs]
        out = disc_losses(logits[0], labels, logits[1], logits[2])
        dls = disc_loss_grads(logits[0], labels, logits[1], logits[2], self.uw)
        grads = [np.zeros_like(p) for p in self.d]
        for c, dl in zip(segs, dls):
            grads[-2] += c['feat_q'].T @ dl
            grads[-1] += dl.sum(axis=0)
            dp = (dl @ W6.T) * (c['feat_q'] > 0)
            grads[2 * (nl - 2) + 1] += dp.sum(axis=0)
            dpre, db, _ = self._disc_bwd8(c, dp, 0)            # the loss head packs e5m2 from the fp32 value
            for l in range(nl - 1):
                grads[2 * l] += c['xin'][l].T @ dpre[l]
                if l < nl - 2:
                    grads[2 * l + 1] += db[l]
        return out, grads, dict(l_lab=logits[0], l_unl=logits[1], l_fake=logits[2])

    def _calibrate(self, kind, fn):
        """first sub-step of a kind: dry passes settle the delayed scales (mrgan_disc_step / mrgan_gen_step)"""
        if self.fp8 and not self.cal[kind]:
            for _ in range(FP8_DRY_PASSES):
                fn()
                self.slots.update()
            self.cal[kind] = True

    # ---- generator forward (engine: stage z -> G1+stats -> bn_apply -> G2 -> G3(+noise)) ----
    def _gen_fwd(self, z, n0):
        q = self.q
        W1, b1, gamma, beta, W2, b2, W3, b3 = self.g
        zq = q(z)
        h1 = softplus(zq @ q(W1) + b1)
        B = h1.shape[0]
        mu = h1.sum(axis=0) / B
        var = np.maximum((h1 * h1).sum(axis=0) / B - mu * mu, 0.0)          # bn_apply_kernel: E[h^2] - E[h]^2 of the unrounded h
        rstd = 1.0 / np.sqrt(var + BN_EPS)
        h1q = q(h1)
        scale = gamma * rstd
        hbn = q(h1q * scale + (beta - mu * scale))
        if getattr(self, 'fp8', False):
            hbn = self.slots.quant(hbn, ('gx',), 'e4m3')       # e4m3 copy of BN(h1): operand of G2's forward and weight gradient
            h2q = q(softplus(hbn @ self.gw8 + b2))
        else:
            h2q = q(softplus(hbn @ q(W2) + b2))
        x = h2q @ q(W3) + b3
        xin = q(x + np.asarray(self.sigmas[0], x.dtype) * n0) if n0 is not None else q(x)
        return xin, dict(zq=zq, h1q=h1q, mu=mu, rstd=rstd, hbn=hbn, h2q=h2q, B=B)

    # ---- discriminator forward over one segment whose (noisy, rounded) first-layer input is xin0 ----
    def _disc_fwd(self, xin0, noise):
        q = self.q
        nl = len(self.d) // 2
        xin, masks = [xin0], []
        a = None
        for l in range(nl - 1):
            pre = xin[l] @ q(self.d[2 * l]) + self.d[2 * l + 1]
            a = relu(pre)
            masks.append(a > 0)
            if l < nl - 2:
                xin.append(q(a + np.asarray(self.sigmas[l + 1], a.dtype) * noise[l + 1]) if noise is not None else q(a))
        feat_q = q(a)
        return dict(xin=xin, masks=masks, feat=a, feat_q=feat_q)

    def _stage(self, x, n0):
        return self.q(x + np.asarray(self.sigmas[0], x.dtype) * n0)

    # dX chain from dpre of the feature layer down to layer `stop`; returns the list of (rounded) dpre_l and bias sums
    def _disc_bwd(self, c, dpre_top, want_bias):
        q = self.q
        nl = len(self.d) // 2
        dpre = [None] * (nl - 1)
        db = [None] * (nl - 1)
        dpre[nl - 2] = dpre_top
        for l in range(nl - 2, 0, -1):
            v = (dpre[l] @ q(self.d[2 * l]).T) * c['masks'][l - 1]
            if want_bias:
                db[l - 1] = v.sum(axis=0)
            dpre[l - 1] = q(v)
        return dpre, db

    def disc_grads(self, x_lab, labels, x_unl, z, n_lab, n_unl, n_fake):
        if self.fp8:
            a = (x_lab, labels, x_unl, z, n_lab, n_unl, n_fake)
            self._calibrate(0, lambda: self._disc_grads8(*a))
            return self._disc_grads8(*a)
        q = self.q
        nl = len(self.d) // 2
        xf, _ = self._gen_fwd(z, n_fake[0])
        segs = [self._disc_fwd(self._stage(x_lab, n_lab[0]), n_lab), self._disc_fwd(self._stage(x_unl, n_unl[0]), n_unl),
                self._disc_fwd(xf, n_fake)]
        W6, b6 = self.d[-2], self.d[-1]
        logits = [c['feat_q'] @ W6 + b6 for c in segs]
        out = disc_losses(logits[0], labels, logits[1], logits[2])
        dls = disc_loss_grads(logits[0], labels, logits[1], logits[2], self.uw)
        grads = [np.zeros_like(p) for p in self.d]
        for c, dl in zip(segs, dls):
            grads[-2] += c['feat_q'].T @ dl
            grads[-1] += dl.sum(axis=0)
            dp = (dl @ W6.T) * (c['feat_q'] > 0)
            grads[2 * (nl - 2) + 1] += dp.sum(axis=0)
            dpre, db = self._disc_bwd(c, q(dp), True)
            for l in range(nl - 1):
                grads[2 * l] += c['xin'][l].T @ dpre[l]
                if l < nl - 2:
                    grads[2 * l + 1] += db[l]
        return out, grads, dict(l_lab=logits[0], l_unl=logits[1], l_fake=logits[2])

    def disc_step(self, *a, **k):
        out, grads, _ = self.disc_grads(*a, **k)
        self.adam.apply(self.d, grads, 'd')
        if self.fp8:                                   # D_ADAM: refresh the fp8 weight copies, then the scale update
            self._refresh_w8()
            self.slots.update()
        return out

    # supervised step of the NN baseline in the engine's dataflow (engine.hip sup_step)
    def sup_grads(self, x, labels, noise):
        q = self.q
        nl = len(self.d) // 2
        c = self._disc_fwd(self._stage(x, noise[0]), noise)
        W6, b6 = self.d[-2], self.d[-1]
        logits = c['feat_q'] @ W6 + b6
        loss, err, dl = mse_loss_grad(logits, labels)
        grads = [np.zeros_like(p) for p in self.d]
        grads[-2] = c['feat_q'].T @ dl
        grads[-1] = dl.sum(axis=0)
        dp = (dl @ W6.T) * (c['feat_q'] > 0)
        grads[2 * (nl - 2) + 1] = dp.sum(axis=0)
        dpre, db = self._disc_bwd(c, q(dp), True)
        for l in range(nl - 1):
            grads[2 * l] = c['xin'][l].T @ dpre[l]
            if l < nl - 2:
                grads[2 * l + 1] = db[l]
        return (loss, err), grads, dict(logits=logits)

    def sup_step(self, x, labels, noise):
        out, grads, _ = self.sup_grads(x, labels, noise)
        self.adam.apply(self.d, grads, 'd')
        return out

    def gen_grads(self, x_unl, z, n_fake, n_real):
        if self.fp8:
            self._calibrate(1, lambda: self._gen_grads(x_unl, z, n_fake, n_real))
        return self._gen_grads(x_unl, z, n_fake, n_real)

    def _gen_grads(self, x_unl, z, n_fake, n_real):
        q = self.q
        W1, b1, gamma, beta, W2, b2, W3, b3 = self.g
        xf, gc = self._gen_fwd(z, n_fake[0])
        if self.fp8:
            cf = self._disc_fwd8(xf, n_fake, 1)
            cr = self._disc_fwd8(self._stage(x_unl, n_real[0]), n_real, 1)
        else:
            cf = self._disc_fwd(xf, n_fake)
            cr = self._disc_fwd(self._stage(x_unl, n_real[0]), n_real)
        B, J = cf['feat'].shape
        diff = cf['feat'].sum(axis=0) / B - cr['feat'].sum(axis=0) / B          # moments of the unrounded features
        loss = np.mean(diff * diff)
        gj = (2.0 / (J * B)) * diff
        if self.fp8:
            _, _, v = self._disc_bwd8(cf, np.where(cf['masks'][-1], gj, 0.0), 1, to_input=True)     # fm_kernel: e5m2 from fp32
        else:
            dpre, _ = self._disc_bwd(cf, q(np.where(cf['masks'][-1], gj, 0.0)), False)
            v = dpre[0] @ q(self.d[0]).T                                          # d loss / d x_fake (noise is additive)
        db3 = v.sum(axis=0)
        dxf = q(v)
        dW3 = gc['h2q'].T @ dxf
        v = (dxf @ q(W3).T) * (-np.expm1(-gc['h2q']))                             # softplus'(pre) = 1 - exp(-h)
        db2 = v.sum(axis=0)
        dpre2 = q(v)
        if self.fp8:
            dpre2 = self.slots.quant(dpre2, ('gg',), 'e5m2')
            dW2 = gc['hbn'].T @ dpre2
            v = dpre2 @ self.gw8.T
        else:
            dW2 = gc['hbn'].T @ dpre2
            v = dpre2 @ q(W2).T
        xh = (gc['h1q'] - gc['mu']) * gc['rstd']
        dbeta = v.sum(axis=0)
        dgamma = (v * xh).sum(axis=0)
        d = q(v)
        dh = (gamma * gc['rstd'] / B) * (B * d - dbeta - xh * dgamma)
        o = dh * (-np.expm1(-gc['h1q']))
        db1 = o.sum(axis=0)
        dW1 = gc['zq'].T @ q(o)
        return loss, [dW1, db1, dgamma, dbeta, dW2, db2, dW3, db3], dict(f_fake=cf['feat'], f_real=cr['feat'])

    def gen_step(self, *a, **k):
        loss, grads, _ = self.gen_grads(*a, **k)
        self.adam.apply(self.g, grads, 'g')
        if self.fp8:
            self._refresh_gw8()
            self.slots.update()
        return loss

    def predict_logits(self, x):
        c = self._disc_fwd(self.q(x), None)
        return c['feat_q'] @ self.d[-2] + self.d[-1]

    def test_error(self, x, labels):
        return np.mean(np.argmax(self.predict_logits(x), axis=1) != labels)


# ----------------------------------------------------------------------------------------
# data prologue and epoch staging (mr_gan.py:96-107, :189-202) -- restated for the host tests
# ----------------------------------------------------------------------------------------
def standard_scale(X_train, X_test):
    """sklearn StandardScaler: per-feature mean and *population* std; zero std -> scale 1."""
    mu = X_train.mean(axis=0)
    sd = X_train.std(axis=0)
    sd = np.where(sd == 0.0, 1.0, sd)
    return (X_train - mu) / sd, (X_test - mu) / sd


def select_labeled(X_train, y_train, n_lab, K=NUM_CLASSES):
    # mr_gan.py:102-103 : first n_lab rows of each class, classes concatenated in order
    x = np.concatenate([X_train[y_train == j][:n_lab] for j in range(K)], axis=0)
    y = np.concatenate([[j] * n_lab for j in range(K)], axis=0)
    return x, y


def tiled_permutation(rng_perm, n_pool, n_total):
    """mr_gan.py:189: floor(n_total/n_pool) permutations of range(n_pool) followed by a
    permutation of range(n_total % n_pool) -- NOT a random subset: the tail only touches the
    first n_total % n_pool rows of the (class-sorted) pool.  rng_perm(n) -> permutation."""
    parts = [rng_perm(n_pool) for _ in range(n_total // n_pool)] + [rng_perm(n_total % n_pool)]
    return np.concatenate(parts).astype(np.int64)


# ----------------------------------------------------------------------------------------
# The build's device noise generator, restated (mr_gan_amd/csrc/common.h: mix32 / noise_key / noise_rowhash /
# noise_block): counter hash -> 32 odd signed bytes per (row, 32-column block) -> +-1 Sylvester-Hadamard mix (one
# v_mfma_i32_32x32x32_i8 per 32x32 block on the device) -> scale.  Integer arithmetic throughout: bit-exact.
# ----------------------------------------------------------------------------------------
SITE_Z = 16     # generator input z; sites 0..4 are the GaussianNoise layers before dense 1..5
NOISE_SCALE = 1.0 / np.sqrt(32.0 * (128.0 ** 2 - 1.0) / 3.0)
_HADAMARD32 = np.array([[1 - 2 * (bin(k & j).count("1") & 1) for j in range(32)] for k in range(32)], dtype=np.int64)


def mix32(x):
    """'lowbias32' integer finaliser on uint32 arrays"""
    x = np.asarray(x, dtype=np.uint32).copy()
    with np.errstate(over='ignore'):
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x7feb352d)
        x ^= x >> np.uint32(15)
        x *= np.uint32(0x846ca68b)
        x ^= x >> np.uint32(16)
    return x


def noise_key(seed, site_seg, step):
    lo, hi = np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF)
    return mix32(lo ^ mix32(hi ^ mix32(np.uint32(step) ^ mix32(np.uint32(site_seg)))))


def device_noise_sums(seed, site, seg, step, rows, cols, row0=0):
    """The integer sums s[rows, cols] behind device_normal (each a signed sum of 32 odd bytes)."""
    nb = (cols + 31) // 32
    key = noise_key(seed, site * 256 + seg, step)
    with np.errstate(over='ignore'):
        r = (np.arange(rows, dtype=np.uint32) + np.uint32(row0))
        rowhash = mix32(key + r * np.uint32(0x9E3779B1))                                        # [rows]
        m = np.arange(nb * 8, dtype=np.uint32) * np.uint32(0x85EBCA77)                          # word index cblk*8 + m
        w = mix32(rowhash[:, None] ^ m[None, :]) | np.uint32(0x01010101)                        # [rows, nb*8]
    # byte t of word m is a[k = 4m + t] (little endian), signed
    a = np.ascontiguousarray(w).view(np.int8).reshape(rows * nb, 32).astype(np.float32)
    s = a @ _HADAMARD32.astype(np.float32)              # |s| <= 32 * 127 < 2^24: exact in float32, and BLAS-fast
    return np.rint(s).astype(np.int64).reshape(rows, nb * 32)[:, :cols]


def device_normal(seed, site, seg, step, rows, cols, row0=0, dtype=np.float64):
    """Standard normals [rows, cols] exactly as the HIP kernels draw them.  row0 = global index of the first row
    within its segment (data-parallel ranks)."""
    return (device_noise_sums(seed, site, seg, step, rows, cols, row0) * NOISE_SCALE).astype(dtype)
