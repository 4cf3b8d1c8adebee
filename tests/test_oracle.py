"""Pins for the CPU oracle (oracle/mrgan_oracle.py).

The reference holds no golden vectors (parity unpinned), so the oracle's closed-form
backward passes are pinned against torch.autograd in fp64 on the same forward expressions as
mr_gan.py:110-154, its Adam against a literal transcription of the Keras-2.0.9 formula, and the
data prologue against sklearn.
"""
import numpy as np
import pytest
import torch

from oracle import mrgan_oracle as O


def _rand_problem(D=24, B=10, seed=3):
    rng = np.random.default_rng(seed)
    g, d = O.init_params(D, seed=seed)
    # non-trivial biases / BN affine so every gradient path is exercised
    g = [p + 0.05 * rng.standard_normal(p.shape) for p in g]
    d = [p + 0.05 * rng.standard_normal(p.shape) for p in d]
    x_lab = rng.standard_normal((B, D))
    x_unl = rng.standard_normal((B, D))
    labels = rng.integers(0, 6, B)
    z = rng.standard_normal((B, O.NOISE_SIZE))
    dims = (D,) + O.D_HIDDEN
    noise = lambda: [rng.standard_normal((B, dims[l])) for l in range(5)]
    return g, d, x_lab, labels, x_unl, z, noise(), noise(), noise()


def _t(a):
    return torch.tensor(a, dtype=torch.float64, requires_grad=True)


def _torch_gen(g, z):
    W1, b1, gamma, beta, W2, b2, W3, b3 = g
    h = torch.nn.functional.softplus(z @ W1 + b1)
    mu = h.mean(0)
    var = ((h - mu) ** 2).mean(0)
    h = gamma * (h - mu) / torch.sqrt(var + O.BN_EPS) + beta
    h = torch.nn.functional.softplus(h @ W2 + b2)
    return h @ W3 + b3


def _torch_disc(d, x, noise, upto_feat=False):
    a = x
    for l in range(5):
        a = a + O.D_SIGMAS[l] * torch.tensor(noise[l])
        a = torch.relu(a @ d[2 * l] + d[2 * l + 1])
    if upto_feat:
        return a
    return a @ d[10] + d[11]


def test_disc_step_grads_match_autograd():
    g, d, x_lab, labels, x_unl, z, n1, n2, n3 = _rand_problem()
    orc = O.MRGANOracle(g, d)
    (ll, lu, err), grads, _ = orc.disc_grads(x_lab, labels, x_unl, z, n1, n2, n3)
    tg = [_t(p) for p in g]
    td = [_t(p) for p in d]
    l_lab = _torch_disc(td, torch.tensor(x_lab), n1)
    l_unl = _torch_disc(td, torch.tensor(x_unl), n2)
    l_fake = _torch_disc(td, _torch_gen(tg, torch.tensor(z)), n3)
    B = x_lab.shape[0]
    lab = torch.tensor(labels)
    sp = torch.nn.functional.softplus
    # mr_gan.py:146-149
    loss_lab = -l_lab[torch.arange(B), lab].mean() + torch.logsumexp(l_lab, 1).mean()
    loss_unl = (-0.5 * torch.logsumexp(l_unl, 1).mean() + 0.5 * sp(torch.logsumexp(l_unl, 1)).mean()
                + 0.5 * sp(torch.logsumexp(l_fake, 1)).mean())
    (loss_lab + loss_unl).backward()
    assert abs(ll - loss_lab.item()) < 1e-12 and abs(lu - loss_unl.item()) < 1e-12
    for a, b in zip(grads, td):
        np.testing.assert_allclose(a, b.grad.numpy(), rtol=1e-9, atol=1e-12)
    assert err == np.mean(l_lab.detach().numpy().argmax(1) != labels)


def test_gen_step_grads_match_autograd():
    g, d, x_lab, labels, x_unl, z, n1, n2, _ = _rand_problem(seed=5)
    orc = O.MRGANOracle(g, d)
    loss, grads, _ = orc.gen_grads(x_unl, z, n1, n2)
    tg = [_t(p) for p in g]
    td = [_t(p) for p in d]
    f_fake = _torch_disc(td, _torch_gen(tg, torch.tensor(z)), n1, upto_feat=True)
    f_real = _torch_disc(td, torch.tensor(x_unl), n2, upto_feat=True)
    tl = ((f_fake.mean(0) - f_real.mean(0)) ** 2).mean()      # mr_gan.py:152-154
    tl.backward()
    assert abs(loss - tl.item()) < 1e-14
    for a, b in zip(grads, tg):
        np.testing.assert_allclose(a, b.grad.numpy(), rtol=1e-8, atol=1e-13)


def test_adam_shared_counter_and_formula():
    # Keras 2.0.9 Adam.get_updates: t = iterations + 1 on ONE counter shared by both lists
    g, d, x_lab, labels, x_unl, z, n1, n2, n3 = _rand_problem(seed=7)
    orc = O.MRGANOracle(g, d)
    d0 = [p.copy() for p in orc.d]
    g0 = [p.copy() for p in orc.g]
    _, gd, _ = orc.disc_grads(x_lab, labels, x_unl, z, n1, n2, n3)
    orc.disc_step(x_lab, labels, x_unl, z, n1, n2, n3)
    lr1 = O.ADAM_LR * np.sqrt(1 - 0.999 ** 1) / (1 - 0.5 ** 1)
    for p0, p1, gr in zip(d0, orc.d, gd):
        m = 0.5 * gr
        v = 0.001 * gr * gr
        np.testing.assert_allclose(p1, p0 - lr1 * m / (np.sqrt(v) + 1e-8), rtol=1e-12, atol=1e-15)
    _, gg, _ = orc.gen_grads(x_unl, z, n1, n2)
    orc.gen_step(x_unl, z, n1, n2)
    lr2 = O.ADAM_LR * np.sqrt(1 - 0.999 ** 2) / (1 - 0.5 ** 2)       # t = 2 for the first G step
    for p0, p1, gr in zip(g0, orc.g, gg):
        m = 0.5 * gr
        v = 0.001 * gr * gr
        np.testing.assert_allclose(p1, p0 - lr2 * m / (np.sqrt(v) + 1e-8), rtol=1e-12, atol=1e-15)
    assert orc.adam.iterations == 2


def test_standard_scale_matches_sklearn():
    from sklearn import preprocessing
    rng = np.random.default_rng(0)
    Xtr = rng.standard_normal((60, 7)) * 3 + 1
    Xtr[:, 3] = 2.0                      # constant feature
    Xte = rng.standard_normal((11, 7))
    sc = preprocessing.StandardScaler()
    a = sc.fit_transform(Xtr)
    b = sc.transform(Xte)
    a2, b2 = O.standard_scale(Xtr, Xte)
    np.testing.assert_allclose(a, a2, atol=1e-12)
    np.testing.assert_allclose(b, b2, atol=1e-12)


def test_tiled_permutation_tail_quirk():
    rng = np.random.RandomState(1)
    inds = O.tiled_permutation(rng.permutation, 480, 6000)
    assert inds.shape == (6000,)
    assert sorted(inds[:480]) == list(range(480))
    assert sorted(inds[5760:]) == list(range(240))      # tail touches only the first 240 pool rows


def test_noise_hash_properties():
    # mix32 is a bijection with good avalanche: flipping one input bit flips ~half the output bits
    x = np.arange(1 << 16, dtype=np.uint32) * np.uint32(2654435761)
    h = O.mix32(x)
    assert len(np.unique(h)) == len(h)
    for bit in (0, 7, 31):
        d = h ^ O.mix32(x ^ np.uint32(1 << bit))
        pop = np.unpackbits(d.view(np.uint8)).mean() * 32
        assert 15.0 < pop < 17.0
    # distinct sites / steps / segments give unrelated streams
    a = O.device_normal(1, 0, 0, 0, 64, 64)
    for other in (O.device_normal(1, 1, 0, 0, 64, 64), O.device_normal(1, 0, 1, 0, 64, 64), O.device_normal(1, 0, 0, 1, 64, 64),
                  O.device_normal(2, 0, 0, 0, 64, 64)):
        assert abs(np.corrcoef(a.ravel(), other.ravel())[0, 1]) < 0.06


def test_device_normal_moments():
    n = O.device_normal(seed=1234, site=1, seg=2, step=7, rows=2048, cols=256)
    assert abs(n.mean()) < 0.01 and abs(n.std() - 1.0) < 0.01
    assert abs((n ** 4).mean() - (3.0 - 0.0375)) < 0.05            # Irwin-Hall(32) kurtosis: 3 - 1.2/32
    assert abs(np.corrcoef(n[:-1].ravel(), n[1:].ravel())[0, 1]) < 0.01     # neighbouring rows
    assert abs(np.corrcoef(n[:, :-1].ravel(), n[:, 1:].ravel())[0, 1]) < 0.01   # neighbouring columns
    # row-offset consistency (data-parallel shards draw the same global stream)
    n2 = O.device_normal(seed=1234, site=1, seg=2, step=7, rows=1024, cols=256, row0=1024)
    np.testing.assert_array_equal(n[1024:], n2)


def test_device_normal_is_integer_exact_and_near_gaussian():
    """The generator is integer arithmetic up to one final scale: the sums are even integers in [-4064, 4064], the 32
    columns of a block are exactly orthogonal sign mixes of the same 32 odd bytes (sum of squares over a block row is
    32 * sum a_k^2), and the marginal passes a Kolmogorov-Smirnov test against N(0,1)."""
    from scipy import stats
    s = O.device_noise_sums(seed=99, site=2, seg=1, step=5, rows=512, cols=96)
    assert s.dtype.kind == 'i' and np.all(s % 2 == 0) and np.abs(s).max() <= 32 * 127
    blk = s[:, 32:64].astype(np.int64)
    e = (blk * blk).sum(axis=1)                       # Parseval: |H a|^2 = 32 |a|^2, a odd => a^2 = 1 mod 8
    assert np.all(e % 32 == 0) and np.all((e // 32) % 8 == 0)
    n = O.device_normal(seed=99, site=2, seg=1, step=5, rows=4096, cols=128).ravel()
    assert stats.kstest(n[:100000], 'norm').pvalue > 1e-3
    assert 4.0 < np.abs(n).max() < 9.73


def test_bf16_round_is_rne():
    x = np.array([1.0, 1.00390625, 1.01171875, -2.5, 3.14159, 1e-30, 65504.0, 0.0], dtype=np.float64)
    want = torch.tensor(x, dtype=torch.float32).to(torch.bfloat16).to(torch.float64).numpy()
    np.testing.assert_array_equal(O.bf16_round(x), want)
    r = np.random.default_rng(0).standard_normal(10000) * 10.0 ** np.random.default_rng(1).integers(-6, 6, 10000)
    np.testing.assert_array_equal(O.bf16_round(r), torch.tensor(r, dtype=torch.float32).to(torch.bfloat16).to(torch.float64).numpy())


def test_mirror_without_rounding_equals_oracle():
    """MRGANMirror restates the same algebra in the engine's dataflow (E[h^2]-mu^2 variance, softplus' from h, column
    sums, per-segment accumulation).  With quantize=None it must reproduce the autograd-pinned MRGANOracle: this is
    what ties the reduced-precision mirrors used by the GPU tests to the pinned restatement."""
    g, d, x_lab, labels, x_unl, z, n1, n2, n3 = _rand_problem(D=40, B=12, seed=9)
    a, b = O.MRGANOracle(g, d), O.MRGANMirror(g, d, quantize=None)
    for step in range(2):
        (la, ga, _), (lb, gb, _) = a.disc_grads(x_lab, labels, x_unl, z, n1, n2, n3), b.disc_grads(x_lab, labels, x_unl, z, n1, n2, n3)
        np.testing.assert_allclose(la, lb, rtol=1e-12)
        for u, v in zip(ga, gb):
            np.testing.assert_allclose(u, v, rtol=1e-9, atol=1e-13)
        a.adam.apply(a.d, ga, 'd'); b.adam.apply(b.d, gb, 'd')
        (la, ga, _), (lb, gb, _) = a.gen_grads(x_unl, z, n1, n2), b.gen_grads(x_unl, z, n1, n2)
        assert abs(la - lb) < 1e-12 * abs(la)
        for u, v in zip(ga, gb):
            np.testing.assert_allclose(u, v, rtol=1e-8, atol=1e-13)
        a.adam.apply(a.g, ga, 'g'); b.adam.apply(b.g, gb, 'g')
    np.testing.assert_allclose(a.predict_logits(x_lab), b.predict_logits(x_lab), rtol=1e-10)


def test_bf16_mirror_stays_close_to_fp64_oracle():
    """sanity of the quantised mirror: bf16 storage rounding moves the gradients of a small batch by <= ~10 %, not more"""
    g, d, x_lab, labels, x_unl, z, n1, n2, n3 = _rand_problem(D=40, B=32, seed=4)
    a, b = O.MRGANOracle(g, d), O.MRGANMirror(g, d, quantize='bf16')
    (_, ga, _), (_, gb, _) = a.disc_grads(x_lab, labels, x_unl, z, n1, n2, n3), b.disc_grads(x_lab, labels, x_unl, z, n1, n2, n3)
    for u, v in zip(ga, gb):
        assert np.linalg.norm(u - v) < 0.25 * np.linalg.norm(u)
    assert max(np.linalg.norm(u - v) / np.linalg.norm(u) for u, v in zip(ga, gb)) > 1e-4      # the rounding is really applied


def test_fp32_arithmetic_alone_moves_post_adam_logits_by_more_than_1e3():
    """Why tests/test_gpu_parity.py does not hold post-update logits to north_star's 1e-3 at (D, B) = (800, 256): the
    restatement itself, evaluated in float32 instead of float64 on identical inputs, differs from its own fp64 run by
    MORE than that after three Adam updates (early Adam moves a weight by ~lr * sign(g), so a gradient element at
    rounding level may step the other way), while logits BEFORE any update agree to ~1e-6.  The GPU test therefore bounds
    the engine's post-update deviation by a multiple of this float32-vs-float64 deviation, and holds the pre-update
    logits to 1e-5."""
    from tests.helpers import Case, rel_err
    c64 = Case(D=800, B=256, steps=3)
    c32 = Case(D=800, B=256, steps=3, dtype=np.float32)
    r64, r32 = c64.run_oracle(), c32.run_oracle()
    assert rel_err(r32['logits0'], r64['logits0']) < 1e-5
    e = rel_err(r32['logits'], r64['logits'])
    assert 1e-3 < e < 5e-2, e


def test_supervised_step_grads_match_autograd_and_mirror():
    """NN baseline (mr_nn.py:101-118): mse against the one-hot label through the noisy discriminator stack, Keras' default
    Adam.  The closed-form gradients are pinned to autograd; the engine-dataflow mirror without rounding must equal them."""
    g, d, x, labels, _, _, n1, _, _ = _rand_problem(D=30, B=20, seed=11)
    a = O.MRGANOracle(g, d, lr=O.NN_ADAM_LR, b1=O.NN_ADAM_B1)
    b = O.MRGANMirror(g, d, quantize=None, lr=O.NN_ADAM_LR, b1=O.NN_ADAM_B1)
    (loss, err), grads, aux = a.sup_grads(x, labels, n1)
    td = [_t(p) for p in d]
    logits = _torch_disc(td, torch.tensor(x), n1)
    onehot = torch.nn.functional.one_hot(torch.tensor(labels), 6).double()
    tl = torch.nn.functional.mse_loss(logits, onehot)            # mean over classes and rows, as Keras' 'mse'
    tl.backward()
    assert abs(loss - tl.item()) < 1e-13
    assert err == np.mean(logits.detach().numpy().argmax(1) != labels)
    for u, v in zip(grads, td):
        np.testing.assert_allclose(u, v.grad.numpy(), rtol=1e-9, atol=1e-13)
    for _ in range(2):
        oa, ob = a.sup_step(x, labels, n1), b.sup_step(x, labels, n1)
        np.testing.assert_allclose(oa, ob, rtol=1e-12)
    for u, v in zip(a.d, b.d):
        np.testing.assert_allclose(u, v, rtol=1e-9, atol=1e-13)
    # first Adam step of the Keras defaults moves every weight by lr (bias-corrected m / sqrt(v) = sign(g))
    moved = np.abs(a.d[0] - d[0])
    assert a.adam.iterations == 2 and moved.max() < 2.01 * O.NN_ADAM_LR


def test_fp8_rounding_matches_torch_float8_and_scale_rule():
    """fp8_round restates OCP e4m3 / e5m2 round-to-nearest-even with subnormals (what gemm_fp8.hip's v_cvt_pk_fp8_f32 /
    v_cvt_pk_bf8_f32 produce from clamped inputs): pinned against torch's float8 conversions.  Fp8Slots.update restates
    fp8_update_scales_kernel: scale = 2^floor(log2(target / amax)), from the fp32 quotient's exponent."""
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(50000) * 10.0 ** rng.uniform(-7, 4, 50000),
                        [0.0, 448.0, 449.0, -1e9, 2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -9, 57344.0, 2.0 ** -16, 2.0 ** -17, 3 * 2.0 ** -17]]).astype(np.float32)
    for fmt, tdt in (('e4m3', torch.float8_e4m3fn), ('e5m2', torch.float8_e5m2)):
        lim = O.FP8_FORMATS[fmt][2]
        want = torch.from_numpy(np.clip(x, -lim, lim)).to(tdt).to(torch.float64).numpy()
        np.testing.assert_array_equal(O.fp8_round(x, fmt), want)
    sl = O.Fp8Slots()
    v = O.bf16_round(rng.standard_normal((64, 64)) * 3.7)
    sl.quant(v, 'a', 'e4m3')
    sl.quant(v * 1e-5, 'g', 'e5m2')
    sl.update()
    amax = np.abs(v).max()
    assert sl.scale['a'] == 2.0 ** np.floor(np.log2(224.0 / amax)) and sl.scale['g'] == 2.0 ** np.floor(np.log2(28672.0 / (amax * 1e-5)))
    assert amax * sl.scale['a'] <= 224.0 < 2 * amax * sl.scale['a']
    q = sl.quant(v, 'a', 'e4m3')
    assert np.abs(q - v).max() <= 2.0 ** -4 * amax                   # 3 mantissa bits: half an ulp of the largest binade
    sl.update(); sl.update()                                           # a pass that wrote nothing keeps the scale
    assert sl.scale['a'] == 2.0 ** np.floor(np.log2(224.0 / amax))


def test_fp8_mirror_calibrates_and_stays_close_to_fp64_oracle():
    """MRGANMirror(quantize='fp8'): the dry passes of the first sub-steps must leave every gradient tensor well scaled (no
    layer flushed to zero by the e5m2 range), and fp8 keeps the gradient directions (cosine > 0.9)"""
    g, d, x_lab, labels, x_unl, z, n1, n2, n3 = _rand_problem(D=40, B=32, seed=4)
    a, b = O.MRGANOracle(g, d), O.MRGANMirror(g, d, quantize='fp8')
    (_, ga, _), (_, gb, _) = a.disc_grads(x_lab, labels, x_unl, z, n1, n2, n3), b.disc_grads(x_lab, labels, x_unl, z, n1, n2, n3)
    assert b.cal == [True, False] and all(('g', 0, l) in b.slots.scale and b.slots.scale[('g', 0, l)] > 1e3 for l in range(5))
    for u, v in zip(ga, gb):
        cos = float((u * v).sum() / np.sqrt((u * u).sum() * (v * v).sum()))
        assert cos > 0.9 and np.abs(v).max() > 0
    (la, ha, _), (lb, hb, _) = a.gen_grads(x_unl, z, n1, n2), b.gen_grads(x_unl, z, n1, n2)
    assert b.cal == [True, True] and abs(la - lb) < 0.1 * abs(la)
    for u, v in zip(ha, hb):
        assert float((u * v).sum() / np.sqrt((u * u).sum() * (v * v).sum())) > 0.8
    # two dry-calibrated mirrors are deterministic
    c = O.MRGANMirror(g, d, quantize='fp8')
    _, gc, _ = c.disc_grads(x_lab, labels, x_unl, z, n1, n2, n3)
    for u, v in zip(gb, gc):
        np.testing.assert_array_equal(u, v)
