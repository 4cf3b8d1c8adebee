"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on identical inputs.

Tolerances.  north_star: logits within 1e-3 rel of the reference arithmetic for the fp32 mode.  "rel" is
scale-relative (max |err| / max |ref|).  Weights after Adam steps are compared relative to the size of the
update: early Adam moves every weight by ~lr * sign(g), so an element whose true gradient is at rounding level
may legitimately move the other way (documented in DESIGN.md).  bf16 mode is held to 3e-2 on logits and is pinned
for accuracy, not logits, by north_star (+-0.5 % accuracy).
"""
import numpy as np
import pytest
import torch

from oracle import mrgan_oracle as O
from tests.helpers import SEED, Case, cosine, frob_rel_err, noise_set, rel_err, update_rel_err

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _engine(D, B, dtype, flags=0, rank=0, world=1, seed=SEED, d_hidden=None, g_hidden=None):
    from mr_gan_amd import engine as E
    cfg = E.default_config(D, B)
    cfg.dtype = dtype
    cfg.seed = seed
    cfg.flags = flags
    cfg.rank, cfg.world = rank, world
    for i, w in enumerate(d_hidden or ()):
        cfg.d_hidden[i] = w
    for i, w in enumerate(g_hidden or ()):
        cfg.g_hidden[i] = w
    return E.Engine(cfg, DEV)


def _load(eng, case):
    from mr_gan_amd import engine as E
    eng.set_weights(E.NET_G, [p.astype(np.float32) for p in case.g0])
    eng.set_weights(E.NET_D, [p.astype(np.float32) for p in case.d0])


def _t(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV, dtype)


# ---------------------------------------------------------------------------------------------------------
# kernel level
# ---------------------------------------------------------------------------------------------------------
def test_tr_probe_layout():
    """ds_read_b64_tr_b16 delivers, to lane l of a 32x32x16 A/B fragment, the 8 consecutive k (rows of the
    [k][free] LDS image) of free index l&31, k = 8*(l>>5) + j."""
    from mr_gan_amd import engine as E
    got = E.debug_tr_probe(DEV)[1]
    lanes = np.arange(64)
    want = (((8 * (lanes[:, None] >> 5) + np.arange(8)[None, :]) << 8) | (lanes[:, None] & 31)).astype(np.uint16)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("dtype,tol", [(0, 2e-6), (1, 1.5e-2)])
@pytest.mark.parametrize("m,n,k", [(50, 64, 128), (300, 192, 64), (1024, 256, 512)])
def test_gemm_products(dtype, tol, m, n, k):
    from mr_gan_amd import engine as E
    rng = np.random.default_rng(m + n + k)
    x = rng.standard_normal((m, k)).astype(np.float32)
    w = (rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    dy = rng.standard_normal((m, n)).astype(np.float32)
    x64, w64, dy64 = x.astype(np.float64), w.astype(np.float64), dy.astype(np.float64)
    # asymmetric operands: a transposed C write or a swapped fragment map cannot cancel out
    for act, f in ((0, lambda v: v), (1, lambda v: np.maximum(v, 0)), (2, lambda v: np.logaddexp(0, v))):
        y = E.debug_gemm(dtype, 0, _t(x), _t(w), _t(b), act=act).cpu().numpy()
        assert rel_err(y, f(x64 @ w64 + b)) < tol, ("fwd", act)
    dx = E.debug_gemm(dtype, 1, _t(dy), _t(w)).cpu().numpy()
    assert rel_err(dx, dy64 @ w64.T) < tol
    for splits in (1, 3):
        dw = E.debug_gemm(dtype, 2, _t(x), _t(dy), splits=splits).cpu().numpy()
        assert rel_err(dw, x64.T @ dy64) < tol, splits


@pytest.mark.parametrize("m,n,k,cfg", [(200, 192, 256, -1), (1024, 512, 1024, 1), (1000, 512, 1024, 3), (8192, 4096, 4096, -1)])
def test_fp8_forward_product_matches_e4m3_quantised_reference(m, n, k, cfg):
    """gemm_fp8.hip (v_mfma_scale_f32_32x32x64_f8f6f4, 128x128 and 256x256 blocks): the product of the e4m3-quantised operands, exactly -- the reference
    quantises the same scaled fp32 inputs with torch.float8_e4m3fn (OCP, round to nearest even) and multiplies in fp64, so
    what is left is fp32 accumulation and the bf16 rounding of the output.  Asymmetric operands, ragged M."""
    from mr_gan_amd import engine as E
    rng = np.random.default_rng(m + k)
    x = rng.standard_normal((m, k)).astype(np.float32)
    w = (rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    pow2 = lambda v: float(2.0 ** np.floor(np.log2(224.0 / np.abs(v).max())))        # per-tensor power-of-two scale, 2x headroom
    sa, sb = pow2(x), pow2(w)
    q = lambda v, sc: torch.from_numpy(v * np.float32(sc)).to(torch.float8_e4m3fn).to(torch.float64).numpy() / sc
    for act, f in ((0, lambda v: v), (1, lambda v: np.maximum(v, 0))):
        got, _ = E.debug_gemm_fp8(_t(x), _t(w), _t(b), act=act, scale_a=sa, scale_b=sb, kc_cfg=cfg)
        want = f((torch.from_numpy(q(x, sa)).to(DEV) @ torch.from_numpy(q(w, sb)).to(DEV)).cpu().numpy() + b)
        assert rel_err(got.cpu().numpy(), want) < 6e-3, act             # bf16 output rounding (2^-9 of the largest element)
        assert rel_err(got.cpu().numpy(), f(x.astype(np.float64) @ w + b)) < 0.1     # and fp8 itself stays a ~3 % perturbation


def test_device_noise_matches_restatement():
    eng = _engine(16, 52, 0)
    # the generator is integer arithmetic (hash -> bytes -> i8 MFMA with a +-1 Hadamard operand) up to one fp32 scale:
    # the device must reproduce the restatement's INTEGER sums exactly -- this is also what pins the operand lane maps of
    # v_mfma_i32_32x32x32_i8 (a wrong k pairing or a transposed result changes the sums)
    for site, seg, step, rows, cols, row0 in [(0, 0, 0, 52, 16, 0), (3, 2, 7, 50, 250, 0), (16, 0, 5, 48, 100, 48), (2, 1, 9, 70, 96, 37)]:
        got = eng.debug_noise(site, seg, step, rows, cols, row0).cpu().numpy()
        sums = O.device_noise_sums(SEED, site, seg, step, rows, cols, row0=row0)
        np.testing.assert_array_equal(np.rint(got.astype(np.float64) / O.NOISE_SCALE).astype(np.int64), sums)
        np.testing.assert_allclose(got, O.device_normal(SEED, site, seg, step, rows, cols, row0=row0), rtol=2e-7, atol=0)
    big = eng.debug_noise(1, 1, 3, 2048, 512).cpu().numpy()
    assert abs(big.mean()) < 5e-3 and abs(big.std() - 1) < 5e-3
    eng.close()


# ---------------------------------------------------------------------------------------------------------
# the compiled functions of mr_gan.py:169-171
# ---------------------------------------------------------------------------------------------------------
def _run_engine(eng, case, device_z=False):
    from mr_gan_amd import engine as E
    out = dict(disc=[], gen=[])
    out['logits0'] = eng.predict_logits(_t(case.probe)).cpu().numpy()
    for t in range(case.steps):
        da = E.Engine.disc_args(_t(case.x_lab[t]), _t(case.labels[t], torch.int32), _t(case.x_unl[t]),
                                None if device_z else _t(case.z1[t]))
        out['disc'].append(eng.disc_step(da))
        ga = E.Engine.gen_args(_t(case.x_unl2[t]), None if device_z else _t(case.z2[t]))
        out['gen'].append(eng.gen_step(ga))
    out['logits'] = eng.predict_logits(_t(case.probe)).cpu().numpy()
    out['g'] = eng.get_weights(E.NET_G)
    out['d'] = eng.get_weights(E.NET_D)
    return out


@pytest.mark.parametrize("D,B", [(16, 50), (400, 50), (72, 132), (800, 256)])     # (800, .) = force-only modality (config 4)
def test_fp32_steps_match_oracle(D, B):
    case = Case(D=D, B=B, steps=3)
    ref = case.run_oracle()
    eng = _engine(D, B, 0)
    _load(eng, case)
    got = _run_engine(eng, case)
    assert rel_err(got['logits0'], ref['logits0']) < 1e-5
    # Everything after the first Adam update is bounded by what plain float32 arithmetic allows: the restatement evaluated
    # in float32 on the same inputs deviates from its own fp64 run (rounding differences of near-zero gradients pass
    # through Adam's m / (sqrt(v) + eps) as +-lr steps; tests/test_oracle.py shows > 1e-3 on logits at (800, 256)), so the
    # engine gets max(the tight tolerance, 3 x that float32-vs-float64 deviation).  The first sub-step has no such slack.
    r32 = Case(D=D, B=B, steps=3, dtype=np.float32).run_oracle()
    for t in range(case.steps):
        dev = max(abs(a - b) / max(abs(b), 1e-12) for a, b in zip(r32['disc'][t][:2], ref['disc'][t][:2]))
        np.testing.assert_allclose(got['disc'][t][:2], ref['disc'][t][:2], rtol=2e-4 if t == 0 else max(2e-4, 3 * dev), atol=2e-5)
        assert abs(got['disc'][t][2] - ref['disc'][t][2]) <= (1e-6 if t == 0 else 1.01 / B)
        dev = abs(r32['gen'][t] - ref['gen'][t]) / abs(ref['gen'][t])
        np.testing.assert_allclose(got['gen'][t], ref['gen'][t], rtol=max(2e-3, 3 * dev), atol=1e-9)
    # weights after three (D, G) pairs: pins the shared Adam counter (t = 2n-1 / 2n)
    for i, (w, wr, w0, w32) in enumerate(zip(got['d'], ref['d'], case.d0, r32['d'])):
        assert update_rel_err(w, wr, w0) < max(0.02, 3 * update_rel_err(w32, wr, w0)), ("D", i, update_rel_err(w, wr, w0), update_rel_err(w32, wr, w0))
    for i, (w, wr, w0, w32) in enumerate(zip(got['g'], ref['g'], case.g0, r32['g'])):
        assert update_rel_err(w, wr, w0) < max(0.02, 3 * update_rel_err(w32, wr, w0)), ("G", i, update_rel_err(w, wr, w0), update_rel_err(w32, wr, w0))
    # logits of fresh rows after the three updates.  north_star's 1e-3 holds wherever plain float32 arithmetic allows it:
    # the restatement evaluated in float32 on the same inputs deviates from its fp64 run by e32 (rounding differences of
    # near-zero gradients pass through Adam's m / (sqrt(v) + eps); tests/test_oracle.py shows e32 > 1e-3 at (800, 256)),
    # so the engine is bounded by max(1e-3, 2 * e32)
    e32 = rel_err(r32['logits'], ref['logits'])
    assert rel_err(got['logits'], ref['logits']) < max(1e-3, 2.0 * e32), (rel_err(got['logits'], ref['logits']), e32)
    assert eng.get_iterations() == 2 * case.steps
    eng.close()


def test_fp32_gradients_match_oracle():
    """Flat-gradient mode exposes the raw gradients of one D step and one G step."""
    from mr_gan_amd import engine as E
    case = Case(D=48, B=50, steps=1)
    orc = O.MRGANOracle(case.g0, case.d0)
    (ll, lu, err), gd, _ = orc.disc_grads(**case.disc_inputs(0, 0))
    eng = _engine(48, 50, 0, flags=E.FLAG_FLAT_GRADS | E.FLAG_SYNC_STATS)
    _load(eng, case)
    da = E.Engine.disc_args(_t(case.x_lab[0]), _t(case.labels[0], torch.int32), _t(case.x_unl[0]), _t(case.z1[0]))
    eng.disc_step(da, E.D_GEN, E.D_MAIN, want_outputs=False)
    got = eng.get_slot(E.NET_D, 2)
    for i, (a, b) in enumerate(zip(got, gd)):
        assert rel_err(a, b) < 2e-5, ("dD", i)          # measured ~5e-7 (scripts/parity_probe.py)
    out = eng.disc_step(da, E.D_ADAM, E.D_ADAM)
    np.testing.assert_allclose(out, (ll, lu, err), rtol=2e-4, atol=2e-5)
    orc.adam.apply(orc.d, gd, 'd')
    loss, gg, _ = orc.gen_grads(**case.gen_inputs(0, 1))
    ga = E.Engine.gen_args(_t(case.x_unl2[0]), _t(case.z2[0]))
    eng.gen_step(ga, E.G_GEN, E.G_TAIL, want_outputs=False)
    got = eng.get_slot(E.NET_G, 2)
    for i, (a, b) in enumerate(zip(got, gg)):
        assert rel_err(a, b) < 2e-4, ("dG", i)          # measured ~1e-5 on db1 (cancellation), ~1e-6 elsewhere
    assert abs(eng.gen_step(ga, E.G_ADAM, E.G_ADAM) - loss) < 2e-3 * abs(loss) + 1e-9
    eng.close()


@pytest.mark.parametrize("dtype,D,B,short", [(0, 48, 20, 0), (0, 400, 20, 7), (1, 400, 20, 0), (1, 72, 50, 33)])
def test_supervised_steps_match_oracle(dtype, D, B, short):
    """mrgan_sup_step = one train_on_batch of the NN baseline (mr_nn.py:101-118).  fp32 against the fp64 restatement, bf16
    against the bf16 mirror (tolerance rule of test_bf16_steps_match_bf16_mirror); `short` = Keras' short last batch."""
    from mr_gan_amd import engine as E
    case = Case(D=D, B=B, steps=3)
    kw = dict(lr=O.NN_ADAM_LR, b1=O.NN_ADAM_B1)
    ref = O.MRGANOracle(case.g0, case.d0, **kw)
    mir = O.MRGANMirror(case.g0, case.d0, quantize='bf16' if dtype else None, **kw)
    cfg = E.default_config(D, B)
    cfg.dtype, cfg.seed, cfg.lr, cfg.beta1 = dtype, SEED, O.NN_ADAM_LR, O.NN_ADAM_B1
    eng = E.Engine(cfg, DEV)
    _load(eng, case)
    for t in range(case.steps):
        n = short if (short and t == 1) else B
        x, y = case.x_lab[t].astype(np.float64), case.labels[t]
        noise = [m[:n] for m in noise_set(SEED, 0, t, B, D)]
        yb = y.copy()
        yb[n:] = -1
        got = eng.sup_step(E.Engine.sup_args(_t(case.x_lab[t]), _t(yb, torch.int32), rows_valid=0 if n == B else n))
        want, wm = ref.sup_step(x[:n], y[:n], noise), mir.sup_step(x[:n], y[:n], noise)
        slack = 0.0 if t == 0 else 0.25
        if dtype == 0:
            assert abs(got[0] - want[0]) < (2e-4 + slack * 0.02) * want[0], (t, got, want)
        else:
            assert abs(got[0] - wm[0]) < max(3e-3, 0.6 * abs(wm[0] - want[0]) / want[0] + slack * 0.2) * want[0], (t, got, wm, want)
        assert abs(got[1] - (want[1] if dtype == 0 else wm[1])) <= (1e-6 if t == 0 else 2.01 / n)
    w = eng.get_weights(E.NET_D)
    for i, (a, b, m, w0) in enumerate(zip(w, ref.d, mir.d, case.d0)):
        if dtype == 0:
            assert update_rel_err(a, b, w0) < 0.03, ("D", i, update_rel_err(a, b, w0))
        else:
            assert update_rel_err(a, m, w0) < max(0.05, 0.85 * update_rel_err(m, b, w0)), ("D", i, update_rel_err(a, m, w0), update_rel_err(m, b, w0))
    assert eng.get_iterations() == case.steps
    eng.close()


def test_nn_baseline_fit_learns_planted_structure():
    """MRNN.fit / evaluate (mr_nn.py:117-118) end to end on a separable problem, short last batch included"""
    from mr_gan_amd.mr_nn import MRNN
    rng = np.random.default_rng(5)
    centers = rng.standard_normal((6, 40)) * 2.0
    y = np.repeat(np.arange(6), 37).astype(np.int32)              # 222 rows: 11 batches of 20 + one of 2
    x = (centers[y] + rng.standard_normal((len(y), 40))).astype(np.float32)
    yt = np.repeat(np.arange(6), 50).astype(np.int32)
    xt = (centers[yt] + rng.standard_normal((len(yt), 40))).astype(np.float32)
    model = MRNN(40, seed=3)
    before = model.evaluate(xt, yt)
    hist = model.fit(x, y, epochs=15, rng=np.random.RandomState(1))
    after = model.evaluate(xt, yt)
    assert model.engine.get_iterations() == 15 * 12
    assert before > 0.5 and after < 0.05 and hist[-1]['loss'] < 0.1, (before, after, hist)
    model.engine.close()


def test_logmel_kernel_matches_numpy_restatement():
    """mrgan_logmel (csrc/logmel.hip) against oracle/melspec_oracle.py (fp64 numpy restatement of librosa 0.5.1's
    melspectrogram + logamplitude, mr_gan.py:42-47; parity unpinned against librosa itself).  fp32 FFT vs fp64: the values
    are dB relative to the trial maximum with an 80 dB floor, tolerance 0.02 dB."""
    from mr_gan_amd.melspec import log_melspectrogram_batch, log_melspectrogram_device, logmel_frames
    from oracle.melspec_oracle import log_melspectrogram
    rng = np.random.default_rng(0)
    sr, n = 48000, 9600
    t = np.arange(n) / sr
    sigs = [np.sin(2 * np.pi * 3000 * t),                                             # pure tone: most bands at the floor
            rng.standard_normal(n),                                                   # white noise
            0.01 * rng.standard_normal(n) + np.sin(2 * np.pi * 440 * t) * np.exp(-t * 30),      # decaying tap + noise floor
            np.cumsum(rng.standard_normal(n)) * 1e-3,                                 # red noise, large dynamic range
            np.zeros(n)]                                                              # silence: everything at amin
    sigs += [rng.standard_normal(n) * 10.0 ** rng.uniform(-3, 1) for _ in range(59)]
    assert logmel_frames(n) == 19
    got = log_melspectrogram_device(_t(np.stack(sigs)), sr=sr, n_mels=128).cpu().numpy()
    assert got.shape == (64, 128 * 19)
    for i, s in enumerate(sigs):
        want = log_melspectrogram(np.asarray(s, np.float32).astype(np.float64), sr=sr).flatten()
        assert np.abs(got[i] - want).max() < 0.02, (i, np.abs(got[i] - want).max())
        assert got[i].max() == 0.0 and got[i].min() >= -80.0
    # ragged trial lengths, another mel count and sample rate, and the reference's argument checks
    mixed = [rng.standard_normal(m) for m in (9600, 4800, 9600, 2049, 4800)]
    outs = log_melspectrogram_batch(mixed, sr=44100, n_mels=40)
    for s, o in zip(mixed, outs):
        want = log_melspectrogram(np.asarray(s, np.float32).astype(np.float64), sr=44100, n_mels=40).flatten()
        assert o.shape == want.shape and np.abs(o - want).max() < 0.02
    with pytest.raises(RuntimeError, match="more than 1024 samples"):
        log_melspectrogram_device(_t(rng.standard_normal((2, 1024))))


def test_dataset_contact_mic_modalities_on_the_gpu(tmp_path):
    """dataset() (mr_gan.py:23-71) with its default front end: the log-mel blocks of modalities 3 .. 6 come from ONE mrgan_logmel
    launch over all trials and match the rows built with the numpy restatement (tests/test_dataset.py) within 0.02 dB"""
    from mr_gan_amd import dataset
    from tests.test_dataset import _oracle_logmel, _write_fake_mreo
    _write_fake_mreo(str(tmp_path))
    for mod, off in ((3, 0), (5, 1200)):
        X, y = dataset(modalities=mod, data_dir=str(tmp_path))
        Xo, yo = dataset(modalities=mod, data_dir=str(tmp_path), logmel_fn=_oracle_logmel)
        assert X.shape == Xo.shape and np.array_equal(y, yo)
        np.testing.assert_array_equal(X[:, :off], Xo[:, :off])
        assert np.abs(X[:, off:] - Xo[:, off:]).max() < 0.02
    objs = dataset(modalities=6, leaveObjectOut=True, data_dir=str(tmp_path))
    assert len(objs) == 12 and np.array(objs['glass_obj1']['x']).shape == (3, 800 + 128 * 19)


def test_mr_nn_function_end_to_end():
    """mr_nn(X, y, ...) (mr_nn.py:69-119) on the MREO surrogate: split, scaling, labeled subset, 100-epoch-style fit (shortened),
    whole-test-set error; fp32 and bf16 engines"""
    from mr_gan_amd import synthetic_mreo
    from mr_gan_amd.mr_nn import mr_nn
    X, y, _ = synthetic_mreo(d=60, trials=40, sep=2.0)
    for dtype in ('float32', 'bfloat16'):
        err = mr_nn(X, y, percentlabeled=4, epochs=12, dtype=dtype, seed=11)
        assert 0.0 <= err < 0.4, (dtype, err)              # chance is 0.83; 40 labeled rows per class, 12 epochs: ~0.2


def test_device_z_matches_restatement():
    case = Case(D=16, B=52, steps=2, device_z=True)
    ref = case.run_oracle()
    eng = _engine(16, 52, 0)
    _load(eng, case)
    got = _run_engine(eng, case, device_z=True)
    for t in range(case.steps):
        np.testing.assert_allclose(got['disc'][t], ref['disc'][t], rtol=3e-4, atol=3e-5)
        np.testing.assert_allclose(got['gen'][t], ref['gen'][t], rtol=3e-3, atol=1e-9)
    eng.close()


def test_eval_error_and_logits_large():
    case = Case(D=400, B=128, steps=0)
    eng = _engine(400, 128, 0)
    _load(eng, case)
    rng = np.random.default_rng(0)
    n = 1200                                      # > 3*S rows: exercises the chunked evaluation
    X = rng.standard_normal((n, 400)).astype(np.float32)
    y = rng.integers(0, 6, n).astype(np.int32)
    orc = O.MRGANOracle(case.g0, case.d0)
    ref = orc.predict_logits(X.astype(np.float64))
    got = eng.predict_logits(_t(X)).cpu().numpy()
    assert rel_err(got, ref) < 1e-5
    assert abs(eng.eval_error(_t(X), _t(y, torch.int32)) - orc.test_error(X.astype(np.float64), y)) < 1e-6
    idx = rng.permutation(n)[:300].astype(np.int32)
    got = eng.predict_logits(_t(X), _t(idx, torch.int32)).cpu().numpy()
    assert rel_err(got, ref[idx]) < 1e-5
    eng.close()


def test_bf16_steps_match_bf16_mirror():
    """Three (D, G) pairs in bf16 mode.  Tight assertions are against the oracle's bf16 MIRROR (same algebra, rounded to
    bf16 exactly where the engine stores bf16: oracle/mrgan_oracle.py::MRGANMirror), which separates "bf16 rounding" from
    "bug"; the loose assertions against the fp64 oracle say how far bf16 itself moves the trajectory."""
    case = Case(D=400, B=128, steps=3)
    ref = case.run_oracle()
    mir = case.run_oracle(mirror=True, quantize='bf16')
    eng = _engine(400, 128, 1)
    _load(eng, case)
    got = _run_engine(eng, case)
    # (1) vs the bf16 mirror
    assert rel_err(got['logits0'], mir['logits0']) < 2e-3
    rel = lambda a, b: abs(a - b) / max(abs(b), 1e-12)
    for t in range(case.steps):
        # first sub-step: same weights on both sides, tight.  Later sub-steps run on weights that went through Adam, whose
        # early steps (~lr * sign(g)) amplify rounding-level gradient differences: bounded by a fraction of what bf16 itself
        # does to the trajectory (mirror vs fp64)
        for k in range(2):
            slack = 0.0 if t == 0 else max(1e-2, 2.0 * rel(mir['disc'][t][k], ref['disc'][t][k]))
            assert rel(got['disc'][t][k], mir['disc'][t][k]) < max(5e-4, slack), (t, k, got['disc'][t], mir['disc'][t], ref['disc'][t])
        assert abs(got['disc'][t][2] - mir['disc'][t][2]) <= 2.01 / 128                          # train error: a row or two may flip
        assert rel(got['gen'][t], mir['gen'][t]) < max(2e-3 if t == 0 else 2e-2, (0.6 if t == 0 else 2.0) * rel(mir['gen'][t], ref['gen'][t])), (t, got['gen'][t], mir['gen'][t], ref['gen'][t])
    report = []
    for name, ws, wm, wr_, w0 in (("D", got['d'], mir['d'], ref['d'], case.d0), ("G", got['g'], mir['g'], ref['g'], case.g0)):
        for i, (w, wm_i, wr, wi) in enumerate(zip(ws, wm, wr_, w0)):
            em, eo, emo = update_rel_err(w, wm_i, wi), update_rel_err(w, wr, wi), update_rel_err(wm_i, wr, wi)
            report.append((name, i, em, eo, emo))
    print("\nweights after 3 pairs, error / update size: vs mirror | vs fp64 | mirror vs fp64\n  " +
          "\n  ".join("%s%-2d %.3f %.3f %.3f" % r for r in report))
    for name, i, em, eo, emo in report:
        # weights after three Adam updates, relative to the size of the update.  Early Adam steps are ~lr * sign(g), so elements
        # whose gradients sit at rounding level step differently in ANY two evaluations; the engine must still be closer to the
        # mirror than bf16 storage moves the mirror away from fp64
        assert em < max(0.05, 0.9 * emo), (name, i, em, emo)
        assert eo < 0.5, (name, i, eo)                       # (2) loose, vs fp64
    assert rel_err(got['logits'], mir['logits']) < max(5e-3, 0.6 * rel_err(mir['logits'], ref['logits']))
    # (2) vs the fp64 oracle (what bf16 storage costs)
    assert rel_err(got['logits0'], ref['logits0']) < 3e-2
    for t in range(case.steps):
        np.testing.assert_allclose(got['disc'][t][:2], ref['disc'][t][:2], rtol=3e-2, atol=3e-3)
        np.testing.assert_allclose(got['gen'][t], ref['gen'][t], rtol=0.15, atol=1e-8)
    eng.close()


def _grad_parity(D, B, dtype, quantize, tol, tol_loss, d_hidden=None, g_hidden=None, eval_first=True, frac=0.6, loose=(0.995, 0.98, 0.25)):
    """One D sub-step and one G sub-step in flat-gradient mode: all 20 gradient tensors and the four losses.

    fp32 engine (quantize None): against the fp64 restatement at `tol`.
    bf16 engine: against the oracle MIRROR, which rounds to bf16 exactly where the engine stores bf16.  What is left
    between engine and mirror is fp32-vs-fp64 accumulation: a pre-activation is a sum of K signed terms, so its fp32 error
    relative to its own size is ~sqrt(K) * 1e-7 ~ 1e-5 .. 1e-4, which flips the bf16 rounding of a few per cent of the stored
    activations by one ulp (2^-8); ten chained layers and the cancellation in the bias gradients bring that to 1e-3 .. 2e-2
    on the gradients (measured: scripts/parity_probe.py).  A wrong kernel shows up as an error against the mirror as
    large as the error against the fp64 oracle, so the bound is: err(engine, mirror) < max(tol, frac * err(mirror, fp64)), frac = 0.6
    -- the mirror must explain most of what bf16 does -- and, labelled loose, the direction against fp64."""
    from mr_gan_amd import engine as E
    kw = {}
    if d_hidden:
        kw = dict(d_hidden=d_hidden, g_hidden=g_hidden)
    case = Case(D=D, B=B, steps=1, **kw)
    mir = O.MRGANMirror(case.g0, case.d0, quantize=quantize)
    orc = O.MRGANOracle(case.g0, case.d0)
    (ll, lu, err), gd_m, _ = mir.disc_grads(**case.disc_inputs(0, 0))
    (ll_o, lu_o, _), gd_o, _ = orc.disc_grads(**case.disc_inputs(0, 0))
    eng = _engine(D, B, dtype, flags=E.FLAG_FLAT_GRADS, **kw)
    _load(eng, case)
    if eval_first:
        # an evaluation first: it fills ALL rows of the activation buffers (also the padding rows of a ragged batch),
        # which the training step afterwards must tolerate
        rs = np.random.RandomState(5)
        eng.eval_error(_t(rs.randn(3 * 128 + 7, D).astype(np.float32)), _t(rs.randint(0, 6, size=3 * 128 + 7), torch.int32))
    da = E.Engine.disc_args(_t(case.x_lab[0]), _t(case.labels[0], torch.int32), _t(case.x_unl[0]), _t(case.z1[0]))
    eng.disc_step(da, E.D_GEN, E.D_MAIN, want_outputs=False)
    report = []

    def check(name, got, want_m, want_o, cos_min):
        for i, (a, m, o) in enumerate(zip(got, want_m, want_o)):
            em, eo, emo = frob_rel_err(a, m), frob_rel_err(a, o), frob_rel_err(m, o)
            report.append("%s%-2d %.1e %.1e %.1e" % (name, i, em, eo, emo))
            assert em < max(tol, frac * emo), (name + " vs mirror", i, em, emo)
            if quantize:      # loose, vs fp64: ten chained contractions on bf16 operands keep the gradient's direction
                assert cosine(a, o) > cos_min and eo < loose[2], (name + " vs fp64", i, cosine(a, o), eo)

    check("dD", eng.get_slot(E.NET_D, 2), gd_m, gd_o, loose[0])
    out = eng.disc_step(da, E.D_ADAM, E.D_ADAM)
    # losses: the same rule as the gradients -- within tol_loss of the mirror, or within `frac` of what the storage format itself
    # does to the loss (mirror vs fp64), whichever is larger (reductions of length 4096 in fp8: 2.0e-3 against a mirror that is
    # itself 1 % from fp64)
    for got_l, m_l, o_l in zip(out[:2], (ll, lu), (ll_o, lu_o)):
        np.testing.assert_allclose(got_l, m_l, rtol=max(tol_loss, frac * abs(m_l - o_l) / max(abs(o_l), 1e-12)), atol=tol_loss * 0.1)
    assert abs(out[2] - err) <= ((4.01 if quantize == 'fp8' else 1.01) / B if quantize else 1e-6)      # an argmax or two may flip
    # the G sub-step sees the D network AFTER its update: give engine, mirror and oracle the same updated weights
    mir.adam.apply(mir.d, gd_m, 'd')
    if quantize == 'fp8':           # mrgan_set_weights below re-measures the fp8 weight copies in two passes; so does the mirror
        for _ in range(2):
            mir._refresh_w8()
            mir.slots.update()
    orc.d = [p.copy() for p in mir.d]
    orc.adam.iterations = 1
    eng.set_weights(E.NET_D, [p.astype(np.float32) for p in mir.d])
    loss, gg_m, _ = mir.gen_grads(**case.gen_inputs(0, 1))
    _, gg_o, _ = orc.gen_grads(**case.gen_inputs(0, 1))
    ga = E.Engine.gen_args(_t(case.x_unl2[0]), _t(case.z2[0]))
    eng.gen_step(ga, E.G_GEN, E.G_TAIL, want_outputs=False)
    check("dG", eng.get_slot(E.NET_G, 2), gg_m, gg_o, loose[1])
    lg = eng.gen_step(ga, E.G_ADAM, E.G_ADAM)
    assert abs(lg - loss) < 5 * tol_loss * abs(loss) + 1e-12, (lg, loss)
    eng.close()
    print("\n(D=%d, B=%d) tensor: err vs mirror | vs fp64 | mirror vs fp64\n  " % (D, B) + "\n  ".join(report))


@pytest.mark.parametrize("D,B", [(400, 50),       # the reference's ragged batch: the weight gradients reduce over the padding rows too
                                 (400, 256),      # reference-sized input, several row tiles
                                 (3632, 512),     # all three modalities fused, one rank's shard of batch 4096
                                 (2432, 1024),    # config 4: contact-mic log-mel only
                                 (800, 1024),     # config 4: force only
                                 (400, 1024),     # config 4: temperature only
                                 (512, 4096)])    # config 2: the bench workload at full size
def test_bf16_gradients_match_bf16_mirror(D, B):
    _grad_parity(D, B, 1, 'bf16', tol=3e-3, tol_loss=5e-4)


@pytest.mark.parametrize("D,B", [(400, 256), (96, 50), (512, 1024)])
def test_chain_launches_equal_per_layer_launches(D, B):
    """The 256-wide tail D3..D5 + loss head (+ its dX chain) as row-block chain launches (gemm_chain.hip) against the same
    products launched layer by layer (B = 50: a ragged, partly empty row block).
    * Every dense product has the identical MFMA accumulation order and epilogue arithmetic in both forms: the stored layer
      inputs xin[l] and the features are BIT-IDENTICAL, and so is the whole G sub-step (no loss head in it) when both engines
      start it from the same discriminator weights.
    * The loss head inside the chain runs on the matrix cores (three-addend bf16 splits of the fp32 factors: exact products,
      fp32 accumulation), head_kernel of the per-layer path is an fmaf chain: the same fp32 arithmetic in another summation
      order.  The losses agree to 1e-6; dlogits differ by ~1e-7 relative, which flips the bf16 rounding of a few stored
      dL/d(pre) values by one ulp (measured at (400, 256): 19 of 768 rows hold such an element) -- the gradients therefore agree
      to ~1e-4 of their largest element instead of bit for bit."""
    from mr_gan_amd import engine as E
    case = Case(D=D, B=B, steps=1, device_z=True)
    res, engines = [], []
    for chain in (1, 0):
        eng = _engine(D, B, 1, flags=E.FLAG_FLAT_GRADS)
        eng.set_tuning(E.TUNE_CHAIN, chain)
        _load(eng, case)
        da = E.Engine.disc_args(_t(case.x_lab[0]), _t(case.labels[0], torch.int32), _t(case.x_unl[0]))
        eng.disc_step(da, E.D_GEN, E.D_MAIN, want_outputs=False)
        gd = eng.get_slot(E.NET_D, 2)
        acts = [eng.debug_buffer(0, l).cpu().numpy()[:, :B] for l in range(5)] + [eng.debug_buffer(2, 0).cpu().numpy()[:, :B]]
        dpre = [eng.debug_buffer(1, l).cpu().numpy()[:, :B] for l in range(5)]
        out = eng.disc_step(da, E.D_ADAM, E.D_ADAM)
        res.append((gd, out, acts, dpre))
        engines.append(eng)
    (gd1, out1, acts1, dpre1), (gd0, out0, acts0, dpre0) = res
    np.testing.assert_allclose(out1, out0, rtol=1e-6, atol=1e-7)
    for l, (a, b) in enumerate(zip(acts1, acts0)):
        np.testing.assert_array_equal(a, b, err_msg="layer input / features %d" % l)
    for l, (a, b) in enumerate(zip(dpre1, dpre0)):
        d = np.abs(a - b)
        assert d.max() <= 2.0 ** -7 * np.abs(b).max() and (d.max(axis=2) > 0).mean() < 0.05, ("dpre", l, d.max(), (d.max(axis=2) > 0).mean())
    for i, (a, b) in enumerate(zip(gd1, gd0)):
        assert rel_err(a, b) < 5e-4, ("dD", i, rel_err(a, b))
    # the G sub-step from identical discriminator weights (Adam turns rounding-level gradient differences into +-lr steps)
    wd = engines[0].get_weights(E.NET_D)
    gres = []
    for eng in engines:
        eng.set_weights(E.NET_D, wd)
        ga = E.Engine.gen_args(_t(case.x_unl2[0]))
        eng.gen_step(ga, E.G_GEN, E.G_TAIL, want_outputs=False)
        gg = eng.get_slot(E.NET_G, 2)
        gres.append((gg, eng.gen_step(ga, E.G_ADAM, E.G_ADAM)))
        eng.close()
    (gg1, lg1), (gg0, lg0) = gres
    assert abs(lg1 - lg0) <= 1e-6 * abs(lg0)
    for i, (a, b) in enumerate(zip(gg1, gg0)):
        assert rel_err(a, b) < 2e-5, ("dG", i, rel_err(a, b))


@pytest.mark.parametrize("dtype,B", [(1, 200), (2, 256)])
def test_matrix_core_loss_head_equals_scalar_head(dtype, B):
    """Feature layers wider than the chain holds (here 512 columns = two chunks; BASELINE configs[4]: 4096) run the loss head of
    the D sub-step as the stand-alone MFMA kernel over 64-row blocks (gemm_chain.hip: head_wide_kernel; TUNE_HEAD_MFMA = 1, the
    default) instead of head_kernel's fmaf loops over 32-row blocks.  Same rule as the chain test above: three-addend bf16 splits
    make every product exact, only the fp32 summation order differs -- losses to 1e-6, the bf16 dL/d(pre5) within one ulp on a few
    rows (fp8 mode: the e5m2 copies are what leaves the kernel; compared through the weight gradients), D gradients to 5e-4.
    B = 200: a ragged last row block (8 valid rows); dtype 2: the fp8 engine (e5m2 row-major + transposed copies, amax slot)."""
    from mr_gan_amd import engine as E
    D, hid = 96, (256, 256, 256, 512, 512)
    case = Case(D=D, B=B, steps=1, device_z=True, d_hidden=hid)
    res = []
    for mfma in (1, 0):
        eng = _engine(D, B, dtype, flags=E.FLAG_FLAT_GRADS, d_hidden=hid)
        eng.set_tuning(E.TUNE_HEAD_MFMA, mfma)
        _load(eng, case)
        da = E.Engine.disc_args(_t(case.x_lab[0]), _t(case.labels[0], torch.int32), _t(case.x_unl[0]))
        eng.disc_step(da, E.D_GEN, E.D_MAIN, want_outputs=False)
        gd = eng.get_slot(E.NET_D, 2)
        dpre = eng.debug_buffer(1, 4).cpu().numpy()[:, :B] if dtype == 1 else None
        out = eng.disc_step(da, E.D_ADAM, E.D_ADAM)
        res.append((gd, out, dpre))
        eng.close()
    (gd1, out1, dp1), (gd0, out0, dp0) = res
    np.testing.assert_allclose(out1, out0, rtol=1e-6, atol=1e-7)
    if dtype == 1:
        d = np.abs(dp1 - dp0)
        assert d.max() <= 2.0 ** -7 * np.abs(dp0).max() and (d.max(axis=2) > 0).mean() < 0.05, (d.max(), (d.max(axis=2) > 0).mean())
    for i, (a, b) in enumerate(zip(gd1, gd0)):
        assert rel_err(a, b) < (5e-4 if dtype == 1 else 5e-3), ("dD", i, rel_err(a, b))


@pytest.mark.parametrize("D,B", [(2432, 1024), (400, 50)])
def test_substeps_are_bit_reproducible(D, B):
    """The same D and G sub-step on fresh handles, three times: every gradient and every stored activation identical bit for bit.
    No atomics and no arrival-order reductions exist on the training path, so any difference is a race --
    this is the test that caught a too-weak vmcnt wait at the tail of the chain kernel's 3-stage weight ring, which showed up as
    an intermittent 1e-2 error in one parity case and nowhere else."""
    from mr_gan_amd import engine as E
    case = Case(D=D, B=B, steps=1, device_z=True)
    ref = None
    for rep in range(3):
        eng = _engine(D, B, 1, flags=E.FLAG_FLAT_GRADS)
        _load(eng, case)
        da = E.Engine.disc_args(_t(case.x_lab[0]), _t(case.labels[0], torch.int32), _t(case.x_unl[0]))
        eng.disc_step(da, E.D_GEN, E.D_MAIN, want_outputs=False)
        cur = eng.get_slot(E.NET_D, 2)
        eng.disc_step(da, E.D_ADAM, E.D_ADAM)
        ga = E.Engine.gen_args(_t(case.x_unl2[0]))
        eng.gen_step(ga, E.G_GEN, E.G_TAIL, want_outputs=False)
        cur += eng.get_slot(E.NET_G, 2)
        cur += [eng.debug_buffer(0, l, 2).cpu().numpy() for l in range(5)] + [eng.debug_buffer(1, l, 1).cpu().numpy() for l in range(5)]
        cur.append(eng.debug_buffer(2, 0, 2).cpu().numpy())
        eng.close()
        if ref is None:
            ref = cur
        else:
            for i, (a, b) in enumerate(zip(cur, ref)):
                np.testing.assert_array_equal(a, b, err_msg="tensor %d differs between run %d and run 0" % (i, rep))


def test_narrow_stack_bf16_matches_bf16_mirror():
    """Layers narrower than two k-tiles (64 and 100 columns): the row-block chain needs two weight tiles in flight per reduction, so
    these stacks must take the per-layer launches (engine.hip: chain_ok) -- and the 64 x 64 tile path with ragged widths."""
    # (the loose direction bound against fp64 is wider than at the reference widths: with 64-column layers one bf16 ulp of an
    #  activation is a larger share of the gradient -- the mirror moves from fp64 by the same amount, and the mirror bound holds)
    _grad_parity(72, 132, 1, 'bf16', tol=3e-3, tol_loss=5e-4, d_hidden=(200, 100, 64, 100, 64), g_hidden=(64, 100), eval_first=False,
                 loose=(0.98, 0.95, 0.3))
    # 128-column tail: the shortest reductions the chain accepts (exactly two k-tiles per product, D3 .. D5 and their dX products)
    _grad_parity(72, 132, 1, 'bf16', tol=3e-3, tol_loss=5e-4, d_hidden=(200, 128, 128, 128, 128), g_hidden=(128, 128), eval_first=False,
                 loose=(0.98, 0.95, 0.3))


def test_wide_stack_bf16_matches_bf16_mirror():
    """BASELINE configs[4] geometry at one rank's share: hidden 4096 x 5 (generator 4096 x 2), D = 512, B = 8192 / 8 = 1024.
    The layer widths are literals in the reference (mr_gan.py:111-128); mrgan_config generalises them."""
    # (reductions of length 4096: the fp32 accumulation error, hence the residual against the mirror, is larger: frac 0.85)
    _grad_parity(512, 1024, 1, 'bf16', tol=3e-3, tol_loss=5e-4, d_hidden=(4096,) * 5, g_hidden=(4096,) * 2, eval_first=False, frac=0.85)


# ---------------------------------------------------------------------------------------------------------
# fp8 mode (BASELINE configs[4]): the discriminator's products on the fp8 matrix cores, against the oracle's fp8 mirror
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D,B,hidden", [(400, 256, None),          # reference widths (padded to multiples of 128), 128x128 blocks
                                        (200, 50, None),           # ragged batch: the transposed copies keep zero padding rows
                                        (512, 1024, 2048),         # wide stack, K = 2048 reductions
                                        (512, 1024, 4096)])        # BASELINE configs[4] at one rank's share (8192 / 8 rows): hidden
                                                                   # 4096 x 5, generator 4096 x 2, default tile selection, the
                                                                   # split first-layer weight gradient and the calibration as shipped
def test_fp8_gradients_match_fp8_mirror(D, B, hidden):
    """One D and one G sub-step with MRGAN_FP8 (gemm_fp8.hip: e4m3 activations / weights, e5m2 gradients, delayed power-of-two
    scales settled by the dry passes of the first sub-step) against MRGANMirror(quantize='fp8'), which rounds to fp8 exactly
    where the engine stores fp8.  Same bound as the bf16 mode: err(engine, mirror) < max(tol, frac * err(mirror, fp64)) --
    what is left between engine and mirror is fp32-vs-fp64 accumulation flipping individual fp8 / bf16 roundings, a fraction
    of what fp8 itself does to the gradients (measured ~0.25: scripts/parity_probe.py D B 2).  Loose bounds against fp64 say
    what fp8 costs: gradient direction cosine > 0.9."""
    kw = dict(d_hidden=(hidden,) * 5, g_hidden=(hidden,) * 2) if hidden else {}
    _grad_parity(D, B, 2, 'fp8', tol=5e-3, tol_loss=2e-3, eval_first=hidden is None, frac=0.6 if hidden is None else 0.85,
                 loose=(0.9, 0.8, 0.6), **kw)


def test_fp8_split_weight_gradient_of_a_narrow_first_layer():
    """a first layer with few output tiles (512 x 2048) splits its fp8 weight-gradient product over the batch rows into two
    fp32 slabs (engine.hip fp8_dw_splits); the other layers take the unsplit path"""
    _grad_parity(512, 2048, 2, 'fp8', tol=5e-3, tol_loss=2e-3, eval_first=False, frac=0.7, loose=(0.9, 0.8, 0.6),
                 d_hidden=(2048, 256, 256, 256, 256), g_hidden=(500, 500))


def test_fp8_steps_match_fp8_mirror():
    """three (D, G) pairs in fp8 mode: losses and weights against the fp8 mirror's trajectory (delayed scales included)"""
    case = Case(D=400, B=128, steps=3)
    ref = case.run_oracle()
    mir = case.run_oracle(mirror=True, quantize='fp8')
    eng = _engine(400, 128, 2)
    _load(eng, case)
    got = _run_engine(eng, case)
    rel = lambda a, b: abs(a - b) / max(abs(b), 1e-12)
    for t in range(case.steps):
        for k in range(2):
            # (after the first update the two fp8 trajectories -- 128 rows, e5m2 gradients through Adam's sign-like early steps --
            # are a few per cent apart on a batch loss; the per-step gradient tests are the tight ones)
            slack = 0.0 if t == 0 else max(5e-2, 2.0 * rel(mir['disc'][t][k], ref['disc'][t][k]))
            assert rel(got['disc'][t][k], mir['disc'][t][k]) < max(2e-3, slack), (t, k, got['disc'][t], mir['disc'][t], ref['disc'][t])
        assert abs(got['disc'][t][2] - mir['disc'][t][2]) <= 3.01 / 128
        assert rel(got['gen'][t], mir['gen'][t]) < max(5e-3 if t == 0 else 5e-2, (0.6 if t == 0 else 2.0) * rel(mir['gen'][t], ref['gen'][t])), (t, got['gen'][t], mir['gen'][t], ref['gen'][t])
    for name, ws, wm, wr_, w0 in (("D", got['d'], mir['d'], ref['d'], case.d0), ("G", got['g'], mir['g'], ref['g'], case.g0)):
        for i, (w, wm_i, wr, wi) in enumerate(zip(ws, wm, wr_, w0)):
            em, eo, emo = update_rel_err(w, wm_i, wi), update_rel_err(w, wr, wi), update_rel_err(wm_i, wr, wi)
            # after three Adam updates (early steps ~ lr * sign(g)) engine and mirror have each drifted from fp64 by about the
            # same amount; the per-step gradient tests above are the tight ones
            assert em < max(0.15, 1.2 * emo), (name, i, em, emo)
            assert eo < 0.8, (name, i, eo)
    assert rel_err(got['logits'], mir['logits']) < max(5e-3, 1.0 * rel_err(mir['logits'], ref['logits']))
    assert eng.get_iterations() == 2 * case.steps
    eng.close()


def test_fp8_large_tiles_equal_small_tiles():
    """the 256x256-block instantiations of the fp8 kernel (selected by themselves only at the full configs[4] size) against the
    128x128 ones on the same step: the reduction order per element is the same, so the gradients agree to rounding noise"""
    from mr_gan_amd import engine as E
    case = Case(D=512, B=1024, steps=1, device_z=True, d_hidden=(1024,) * 5, g_hidden=(512, 512))
    res = []
    for cfg in (1, 3):
        eng = _engine(512, 1024, 2, flags=E.FLAG_FLAT_GRADS, d_hidden=(1024,) * 5, g_hidden=(512, 512))
        eng.set_tuning(E.TUNE_KC_CFG, cfg)
        _load(eng, case)
        da = E.Engine.disc_args(_t(case.x_lab[0]), _t(case.labels[0], torch.int32), _t(case.x_unl[0]))
        eng.disc_step(da, E.D_GEN, E.D_MAIN, want_outputs=False)
        gd = eng.get_slot(E.NET_D, 2)
        out = eng.disc_step(da, E.D_ADAM, E.D_ADAM)
        ga = E.Engine.gen_args(_t(case.x_unl2[0]))
        eng.gen_step(ga, E.G_GEN, E.G_TAIL, want_outputs=False)
        res.append((gd, out, eng.get_slot(E.NET_G, 2)))
        eng.close()
    (gd1, out1, gg1), (gd3, out3, gg3) = res
    np.testing.assert_allclose(out1, out3, rtol=1e-5, atol=1e-6)
    for i, (a, b) in enumerate(zip(gd1, gd3)):
        assert rel_err(a, b) < 1e-5, ("dD", i, rel_err(a, b))
    for i, (a, b) in enumerate(zip(gg1, gg3)):
        assert rel_err(a, b) < 1e-3, ("dG", i, rel_err(a, b))       # (the bf16 generator kernels change their tile with the knob too)


def test_wide_stack_fp32_matches_oracle():
    """the same wide geometry through the fp32 MFMA path at a small batch, against the fp64 restatement"""
    _grad_parity(64, 64, 0, None, tol=5e-5, tol_loss=1e-5, d_hidden=(1024, 512, 512, 320, 320), g_hidden=(320, 576), eval_first=False)


# ---------------------------------------------------------------------------------------------------------
# data parallelism, emulated on one GPU: two rank handles whose "all-reduce" is a host-side add
# ---------------------------------------------------------------------------------------------------------
def test_two_rank_emulation_equals_full_batch():
    from mr_gan_amd import engine as E
    B, D = 64, 32
    case = Case(D=D, B=B, steps=2, device_z=True)
    ref = case.run_oracle()
    flags = E.FLAG_FLAT_GRADS | E.FLAG_SYNC_STATS
    ranks = [_engine(D, B // 2, 0, flags=flags, rank=r, world=2) for r in range(2)]
    for e in ranks:
        _load(e, case)

    def allreduce(region):
        views = [e.region(region) for e in ranks]
        tot = views[0] + views[1]
        for v in views:
            v.copy_(tot)

    h = B // 2
    for t in range(case.steps):
        da = [E.Engine.disc_args(_t(case.x_lab[t][r * h:(r + 1) * h]), _t(case.labels[t][r * h:(r + 1) * h], torch.int32),
                                 _t(case.x_unl[t][r * h:(r + 1) * h])) for r in range(2)]
        for e, a in zip(ranks, da):
            e.disc_step(a, E.D_GEN, E.D_GEN, want_outputs=False)
        allreduce(E.REGION_BN_STATS)
        for e, a in zip(ranks, da):
            e.disc_step(a, E.D_MAIN, E.D_MAIN, want_outputs=False)
        allreduce(E.REGION_GRAD_D)
        outs = [e.disc_step(a, E.D_ADAM, E.D_ADAM) for e, a in zip(ranks, da)]
        np.testing.assert_allclose(outs[0], ref['disc'][t], rtol=3e-4, atol=3e-5)
        np.testing.assert_allclose(outs[1], outs[0], rtol=0, atol=0)
        ga = [E.Engine.gen_args(_t(case.x_unl2[t][r * h:(r + 1) * h])) for r in range(2)]
        for ph, reg in ((E.G_GEN, E.REGION_BN_STATS), (E.G_FEAT, E.REGION_FM_MOMENTS), (E.G_BWD, E.REGION_BN_BWD),
                        (E.G_TAIL, E.REGION_GRAD_G)):
            for e, a in zip(ranks, ga):
                e.gen_step(a, ph, ph, want_outputs=False)
            allreduce(reg)
        outs = [e.gen_step(a, E.G_ADAM, E.G_ADAM) for e, a in zip(ranks, ga)]
        np.testing.assert_allclose(outs[0], ref['gen'][t], rtol=3e-3, atol=1e-9)
    w0, w1 = ranks[0].get_weights(E.NET_D), ranks[1].get_weights(E.NET_D)
    for a, b in zip(w0, w1):
        np.testing.assert_array_equal(a, b)                    # replicas stay bit-identical
    for i, (w, wr, wi) in enumerate(zip(w0, ref['d'], case.d0)):
        assert update_rel_err(w, wr, wi) < 0.05, ("D", i)
    for e in ranks:
        e.close()


def test_two_rank_emulation_with_bf16_gradient_payload():
    """MRGAN_FLAG_GRAD_BF16: the reduce phases write the gradients as bfloat16 regions, the host sums those in place (here: a
    host-side add of two rank handles, rounded once like a bf16 all-reduce), the Adam phases read them back -- no fp32 flat buffer
    travels.  Replicas stay bit-identical, the first sub-step's losses are the full-batch ones, and the weights are the fp32
    exchange's up to the bf16 rounding of the gradients (a labelled, different numerical path)."""
    from mr_gan_amd import engine as E
    B, D = 64, 32
    case = Case(D=D, B=B, steps=2, device_z=True)
    ref = case.run_oracle()
    flags = E.FLAG_FLAT_GRADS | E.FLAG_SYNC_STATS | E.FLAG_GRAD_BF16
    ranks = [_engine(D, B // 2, 0, flags=flags, rank=r, world=2) for r in range(2)]
    for e in ranks:
        _load(e, case)
    with pytest.raises(E.MrganError, match="bfloat16"):
        ranks[0].get_slot(E.NET_D, 2)

    def allreduce(region):
        for reg in ({E.REGION_GRAD_D: (E.REGION_GRAD_D_BF16, E.REGION_TAIL_D), E.REGION_GRAD_G: (E.REGION_GRAD_G_BF16, E.REGION_TAIL_G)}.get(region, (region,))):
            views = [e.region(reg) for e in ranks]
            tot = (views[0].float() + views[1].float()).to(views[0].dtype)
            for v in views:
                v.copy_(tot)

    h = B // 2
    for t in range(case.steps):
        da = [E.Engine.disc_args(_t(case.x_lab[t][r * h:(r + 1) * h]), _t(case.labels[t][r * h:(r + 1) * h], torch.int32),
                                 _t(case.x_unl[t][r * h:(r + 1) * h])) for r in range(2)]
        for ph, reg in ((E.D_GEN, E.REGION_BN_STATS), (E.D_MAIN, E.REGION_GRAD_D)):
            for e, a in zip(ranks, da):
                e.disc_step(a, ph, ph, want_outputs=False)
            allreduce(reg)
        outs = [e.disc_step(a, E.D_ADAM, E.D_ADAM) for e, a in zip(ranks, da)]
        if t == 0:
            np.testing.assert_allclose(outs[0], ref['disc'][t], rtol=3e-4, atol=3e-5)
        np.testing.assert_allclose(outs[1], outs[0], rtol=0, atol=0)
        ga = [E.Engine.gen_args(_t(case.x_unl2[t][r * h:(r + 1) * h])) for r in range(2)]
        for ph, reg in ((E.G_GEN, E.REGION_BN_STATS), (E.G_FEAT, E.REGION_FM_MOMENTS), (E.G_BWD, E.REGION_BN_BWD), (E.G_TAIL, E.REGION_GRAD_G)):
            for e, a in zip(ranks, ga):
                e.gen_step(a, ph, ph, want_outputs=False)
            allreduce(reg)
        for e, a in zip(ranks, ga):
            e.gen_step(a, E.G_ADAM, E.G_ADAM)
    w0, w1 = ranks[0].get_weights(E.NET_D), ranks[1].get_weights(E.NET_D)
    for a, b in zip(w0 + ranks[0].get_weights(E.NET_G), w1 + ranks[1].get_weights(E.NET_G)):
        np.testing.assert_array_equal(a, b)                    # replicas stay bit-identical
    errs = [update_rel_err(w, wr, wi) for w, wr, wi in zip(w0, ref['d'], case.d0)]
    assert 1e-6 < max(errs) < 0.3, errs                        # bf16-rounded gradients through two Adam updates
    for e in ranks:
        e.close()


@pytest.mark.parametrize("exact", [True, False])
def test_fp8_phase_protocol_equals_whole_steps(exact):
    """fp8 through the data-parallel phase protocol (mr_gan_amd/dist.py, one rank): with synchronised statistics the HOST runs
    the calibration passes (mrgan_fp8_calibration) with the statistic exchanges in between, with per-shard statistics the
    engine calibrates by itself; both must reproduce the whole-sub-step path exactly (same kernels, same order, same scales)."""
    from mr_gan_amd import engine as E
    from mr_gan_amd.dist import DataParallel, EngineBackend, dp_flags
    D, B = 200, 128
    case = Case(D=D, B=B, steps=2, device_z=True)
    plain = _engine(D, B, 2, flags=E.FLAG_FLAT_GRADS)
    phased = _engine(D, B, 2, flags=dp_flags(exact=exact))
    if exact:                       # a phase-wise first sub-step that skipped the host-side calibration is refused
        probe = _engine(D, B, 2, flags=dp_flags(exact=True))
        _load(probe, case)
        with pytest.raises(E.MrganError, match="calibration"):
            probe.disc_step(E.Engine.disc_args(_t(case.x_lab[0]), _t(case.labels[0], torch.int32), _t(case.x_unl[0])), E.D_GEN, E.D_GEN, want_outputs=False)
        probe.close()
    for e in (plain, phased):
        _load(e, case)
    dp = DataParallel(EngineBackend(phased), exact=exact)
    for t in range(case.steps):
        da = E.Engine.disc_args(_t(case.x_lab[t]), _t(case.labels[t], torch.int32), _t(case.x_unl[t]))
        ga = E.Engine.gen_args(_t(case.x_unl2[t]))
        plain.disc_step(da, want_outputs=False)
        plain.gen_step(ga, want_outputs=False)
        dp.disc_step(da)
        dp.gen_step(ga)
    for net in (E.NET_D, E.NET_G):
        for i, (a, b) in enumerate(zip(plain.get_weights(net), phased.get_weights(net))):
            # (synchronised statistics take the BatchNorm sums through the finalize kernel: another summation order)
            assert update_rel_err(b, a, (case.d0 if net == E.NET_D else case.g0)[i]) < (2e-2 if exact else 1e-6), (net, i)
    assert phased.fp8_calibration(0, E.Engine.FP8_CAL_QUERY) == 1 and phased.fp8_calibration(1, E.Engine.FP8_CAL_QUERY) == 1
    plain.close()
    phased.close()


# ---------------------------------------------------------------------------------------------------------
# host loop
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_fit_learns_planted_structure(dtype):
    """Synthetic data with recoverable structure (the reference's only sanity pattern,
    others/test_activation_map.py:10-25): a few epochs must beat chance by a wide margin."""
    from mr_gan_amd import MRGAN, select_labeled, standard_scale, synthetic_blobs
    X, y = synthetic_blobs(n=6000, d=64, seed=3)
    Xtr, Xte = standard_scale(X[:4800].astype(np.float64), X[4800:].astype(np.float64))
    ytr, yte = y[:4800], y[4800:]
    xl, yl, _ = select_labeled(Xtr, ytr, 20)
    m = MRGAN(64, batch_size=100, dtype=dtype, seed=11)
    hist = m.fit(xl, yl, Xtr, epochs=3, validation_data=(Xte, yte), rng=np.random.RandomState(5))
    assert np.isfinite(hist[-1]['loss_lab']) and np.isfinite(hist[-1]['loss_gen'])
    assert m.evaluate(Xte, yte) < 0.2
    assert m.predict(Xte[:10]).shape == (10,)
    m.engine.close()


def test_accuracy_parity_on_mreo_surrogate():
    """north_star: accuracies within +-0.5 % of the reference arithmetic on identical seeds / inputs.

    The MREO data set is not available offline (README.md:7-11), so this runs SURVEY 8(d) config 1's surrogate at the real size
    (synthetic_mreo: N = 7200 = 6 classes x 12 objects x 100 trials, D = 1200 = force + temperature layout, 6-fold split 0 ->
    6000 training / 1200 test rows, batch 50, mr_gan.py:73-82), at 50 and at 500 labeled rows per class, through
        (a) the HIP engine in fp32, (b) the HIP engine in bf16, (c) the CPU oracle in float32 numpy, (d) the HIP engine in fp8,
    all from the same initial weights, the same index streams (mr_gan.py:189-195) and the same z / GaussianNoise
    streams (the engine's generator, restated in the oracle).  The class separation (sep = 0.5) is chosen so that the final
    error is small but not zero (0.3 - 1 %), i.e. the comparison is not vacuous.
    A single evaluation of one trajectory moves by +-0.3 % from epoch to epoch (GaussianNoise is active in training, and the
    three arithmetic paths diverge chaotically through Adam), so the quantity held to north_star's 0.5 % absolute is the test
    error averaged over the last ten epochs (the reference prints it every epoch, mr_gan.py:219-228); the final whole-set
    error (mr_gan.py:230) of a single evaluation is held to 1 %.  30 epochs instead of the reference's 100 keep the numpy loop
    at ~1.5 min (two worker processes); scripts/accuracy_probe.py runs any length."""
    from sklearn.model_selection import StratifiedKFold
    from mr_gan_amd import MRGAN, select_labeled, standard_scale, synthetic_mreo
    from tests.helpers import run_oracle_fits
    import multiprocessing as mp
    import os
    epochs, seed = 30, 1234
    X, y, _ = synthetic_mreo(sep=0.5)
    tr, te = next(iter(StratifiedKFold(n_splits=6, shuffle=True, random_state=0).split(X, y)))
    Xtr, Xte = standard_scale(X[tr], X[te])
    ytr, yte = y[tr], y[te]
    perm = np.random.RandomState(1).permutation(len(ytr))
    Xtr, ytr = Xtr[perm], ytr[perm]
    probs = {n_lab: select_labeled(Xtr, ytr, n_lab)[:2] for n_lab in (50, 500)}
    # the CPU oracle loops start first, in clean worker processes, and run while the GPU trains
    m0 = MRGAN(Xtr.shape[1], batch_size=50, dtype='float32', seed=seed)
    g0, d0 = m0.get_weights('generator'), m0.get_weights('discriminator')
    m0.engine.close()
    old = {k: os.environ.get(k) for k in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS")}
    os.environ["OPENBLAS_NUM_THREADS"] = os.environ["OMP_NUM_THREADS"] = "4"
    from tests.helpers import oracle_fit_job
    pool = mp.get_context("spawn").Pool(2)
    try:
        asyncs = {n_lab: pool.apply_async(oracle_fit_job, (dict(g0=g0, d0=d0, x_labeled=xl, y_labeled=yl, x_train=Xtr, x_test=Xte, y_test=yte,
                                                                 batch=50, epochs=epochs, seed=seed, rng_seed=5),))
                  for n_lab, (xl, yl) in probs.items()}
        err, last5 = {}, {}        # (last5: mean over the last ten epochs)
        for n_lab, (xl, yl) in probs.items():
            for dt in ('float32', 'bfloat16', 'fp8'):
                m = MRGAN(Xtr.shape[1], batch_size=50, dtype=dt, seed=seed)
                hist = m.fit(xl, yl, Xtr, epochs=epochs, validation_data=(Xte, yte), rng=np.random.RandomState(5))
                err[(n_lab, dt)] = m.evaluate(Xte, yte)
                last5[(n_lab, dt)] = float(np.mean([h['test_err'] for h in hist[-10:]]))
                m.engine.close()
        for n_lab in probs:
            e, log = asyncs[n_lab].get(timeout=800)
            err[(n_lab, 'oracle')], last5[(n_lab, 'oracle')] = e, float(np.mean(log[-10:]))
    finally:
        pool.terminate()
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    print("\ntest error after %d epochs (test rows: %d)\n  final       : %s\n  last-10 mean: %s" % (
        epochs, len(yte), {k: round(v, 4) for k, v in sorted(err.items(), key=str)}, {k: round(v, 4) for k, v in sorted(last5.items(), key=str)}))
    for n_lab in probs:
        e = [err[(n_lab, k)] for k in ('float32', 'bfloat16', 'oracle')]
        a = [last5[(n_lab, k)] for k in ('float32', 'bfloat16', 'oracle')]
        assert max(e) < 0.05, (n_lab, e)                                    # everybody learned the task
        assert max(a) - min(a) <= 0.005 + 1e-9, (n_lab, "last-10 mean", a)   # north_star: +-0.5 % absolute
        assert max(e) - min(e) <= 0.01 + 1e-9, (n_lab, "final", e)
        # the fp8 mode (e4m3 / e5m2 products, a lower-precision path by design; north_star's +-0.5 % is stated for the
        # reference arithmetic in bf16 / fp32): held to 1 % absolute on the ten-epoch mean against the three paths above.
        # Measured: +0.0 .. +0.5 % (50 labeled rows per class), +0.1 .. +0.2 % (500).
        assert abs(last5[(n_lab, 'fp8')] - float(np.mean(a))) <= 0.01 + 1e-9, (n_lab, "fp8 last-10 mean", last5[(n_lab, 'fp8')], a)
        assert err[(n_lab, 'fp8')] < 0.05 and abs(err[(n_lab, 'fp8')] - float(np.mean(e))) <= 0.015 + 1e-9, (n_lab, "fp8 final", err[(n_lab, 'fp8')], e)


def test_six_fold_mean_accuracy_parity_on_mreo_surrogate():
    """The quantity the reference reports (mr_gan.py:255-260): the mean over SIX stratified folds of the final whole-test-set
    error (mr_gan.py:230), on the MREO-shaped surrogate at the real size (N 7200, D 1200, batch 50) at the function's default
    percentlabeled = 50, i.e. 500 labeled rows per class (mr_gan.py:73, :82; Table 1's 50 % column), held to north_star's
    +-0.5 % absolute between
        (a) the HIP engine in fp32, (b) the HIP engine in bf16, (c) the CPU oracle (float32 numpy, six worker processes),
    every fold from the same initial weights, index streams and z / GaussianNoise streams on all three paths.
    (d) the bf16 engine once more with z drawn on the HOST by np.random.normal, as mr_gan.py:206 / :212 do, instead of the
    engine's Irwin-Hall(32) generator (a documented deviation from N(0, 1)): the six-fold mean must not
    move by more than the same 0.5 % -- evidence that the generator's distribution is harmless.  (The GaussianNoise layers keep
    the device generator on every path: they are fused into the product epilogues.)
    20 epochs instead of the reference's 100 keep the six numpy loops at ~2.5 min of wall time.
    Why 500 rows per class and not 50: with 50 labeled rows per class a single final evaluation after 20 epochs is dominated by
    the training's own oscillation -- measured here, per fold: fp32 0.7 .. 1.1 %, bf16 0.4 .. 5.4 %, the float32 oracle 0.4 .. 1.9 %;
    six-fold means 0.86 / 1.86 / 1.22 %, i.e. the two FLOAT32 paths already differ by 0.36 % on identical streams (they diverge
    chaotically through Adam) -- so +-0.5 % on that quantity would test the noise, not the arithmetic.  That label count stays
    covered by test_accuracy_parity_on_mreo_surrogate (one fold, 30 epochs, 0.5 % on the ten-epoch mean and the labelled loose 1 %
    on the single final evaluation)."""
    from sklearn.model_selection import StratifiedKFold
    from mr_gan_amd import MRGAN, select_labeled, standard_scale, synthetic_mreo
    from tests.helpers import oracle_fit_job
    import multiprocessing as mp
    import os
    epochs, seed, n_lab = 20, 4321, 500
    X, y, _ = synthetic_mreo(sep=0.5)
    folds = []
    for k, (tr, te) in enumerate(StratifiedKFold(n_splits=6, shuffle=True, random_state=0).split(X, y)):
        Xtr, Xte = standard_scale(X[tr], X[te])
        ytr, yte = y[tr], y[te]
        perm = np.random.RandomState(100 + k).permutation(len(ytr))          # mr_gan.py:101
        Xtr, ytr = Xtr[perm], ytr[perm]
        xl, yl, _ = select_labeled(Xtr, ytr, n_lab)
        folds.append((Xtr, Xte, yte, xl, yl))
    m0 = MRGAN(X.shape[1], batch_size=50, dtype='float32', seed=seed)
    g0, d0 = m0.get_weights('generator'), m0.get_weights('discriminator')
    m0.engine.close()
    old = {k: os.environ.get(k) for k in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS")}
    os.environ["OPENBLAS_NUM_THREADS"] = os.environ["OMP_NUM_THREADS"] = "2"          # 6 workers x 2 BLAS threads
    pool = mp.get_context("spawn").Pool(6)
    err = {}
    try:
        asyncs = [pool.apply_async(oracle_fit_job, (dict(g0=g0, d0=d0, x_labeled=xl, y_labeled=yl, x_train=Xtr, x_test=Xte, y_test=yte,
                                                         batch=50, epochs=epochs, seed=seed + k, rng_seed=5 + k),))
                  for k, (Xtr, Xte, yte, xl, yl) in enumerate(folds)]
        for name, dt, zsrc in (('float32', 'float32', 'device'), ('bfloat16', 'bfloat16', 'device'), ('bfloat16_host_z', 'bfloat16', 'host')):
            for k, (Xtr, Xte, yte, xl, yl) in enumerate(folds):
                m = MRGAN(X.shape[1], batch_size=50, dtype=dt, seed=seed + k, init_weights=False)
                m.set_weights(g0, 'generator')
                m.set_weights(d0, 'discriminator')
                m.fit(xl, yl, Xtr, epochs=epochs, rng=np.random.RandomState(5 + k), z_source=zsrc)
                err[(name, k)] = m.evaluate(Xte, yte)
                m.engine.close()
        for k, a in enumerate(asyncs):
            err[('oracle', k)] = a.get(timeout=800)[0]
    finally:
        pool.terminate()
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    names = ('float32', 'bfloat16', 'oracle', 'bfloat16_host_z')
    mean = {n: float(np.mean([err[(n, k)] for k in range(6)])) for n in names}
    print("\nfinal whole-test-set error per fold after %d epochs, and the six-fold mean\n  " % epochs +
          "\n  ".join("%-16s %s  mean %.4f" % (n, " ".join("%.4f" % err[(n, k)] for k in range(6)), mean[n]) for n in names))
    assert max(mean.values()) < 0.05, mean                                       # everybody learned the task
    three = [mean[n] for n in ('float32', 'bfloat16', 'oracle')]
    assert max(three) - min(three) <= 0.005 + 1e-9, ("six-fold mean, +-0.5 %", mean)
    assert abs(mean['bfloat16_host_z'] - mean['bfloat16']) <= 0.005 + 1e-9, ("host-drawn normal z vs the device generator", mean)


def test_graph_replay_equals_eager():
    from mr_gan_amd import MRGAN, select_labeled, synthetic_blobs
    X, y = synthetic_blobs(n=1200, d=32, seed=4)
    xl, yl, _ = select_labeled(X, y, 10)
    ws = []
    for use_graph in (False, True):
        m = MRGAN(32, batch_size=64, dtype='float32', seed=21, use_graph=use_graph)
        m.fit(xl, yl, X, epochs=2, rng=np.random.RandomState(9))
        ws.append(m.get_weights('discriminator') + m.get_weights('generator'))
        m.engine.close()
    for a, b in zip(*ws):
        np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_pair_with_shared_generator_pass_equals_separate_substeps(dtype):
    """mrgan_train_pair runs the two generator forwards of a (D, G) pair as one two-segment pass inside the D
    sub-step (same generator weights; z, noise iteration and segment ids of the G sub-step).  The weights after a
    few pairs must be bit-identical to calling the two sub-steps one after the other."""
    from mr_gan_amd import MRGAN, engine as E
    rs = np.random.RandomState(3)
    B, D, n = 64, 40, 256
    X = rs.randn(n, D).astype(np.float32)
    xl = rs.randn(2 * B, D).astype(np.float32)
    yl = rs.randint(0, 6, size=2 * B).astype(np.int32)
    res = []
    for paired in (True, False, "hint"):
        m = MRGAN(D, batch_size=B, dtype=dtype, seed=77, use_graph=False)
        with m._on_stream():
            xu, xlab = m._dev(X), m._dev(xl)
            idx_lab = m._dev(np.arange(n, dtype=np.int32) % (2 * B), torch.int32)
            labs = m._dev(yl[np.arange(n) % (2 * B)], torch.int32)
            idx_u1 = m._dev(rs.permutation(n).astype(np.int32) if False else np.arange(n, dtype=np.int32)[::-1].copy(), torch.int32)
            idx_u2 = m._dev(np.roll(np.arange(n, dtype=np.int32), 7), torch.int32)
            dargs = E.Engine.disc_args(xlab, labs, xu, None, idx_lab, idx_u1, stream_mode=1)
            gargs = E.Engine.gen_args(xu, None, idx_u2, stream_mode=1)
            m.engine.set_iterations(0, 0)
            for _ in range(n // B):
                if paired is True:
                    m.engine.train_pair(dargs, gargs)
                else:
                    if paired == "hint":                      # what the data-parallel host does between its phases
                        m.engine.pair_hint(True)
                    m.engine.disc_step(dargs, want_outputs=False)
                    m.engine.gen_step(gargs, want_outputs=False)
            torch.cuda.synchronize()
            metrics = m.engine.read_metrics(reset=True)
        res.append((m.get_weights('discriminator') + m.get_weights('generator'), metrics))
        m.engine.close()
    for other in res[1:]:
        for a, b in zip(res[0][0], other[0]):
            np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(np.asarray(res[0][1]), np.asarray(other[1]))


@pytest.mark.parametrize("exact", [True, False])
def test_phase_graphs_equal_eager_phases(exact):
    """MRGAN_FLAG_GRAPH on a data-parallel handle: every phase range of the protocol (mr_gan_amd/dist.py) is captured once and
    replayed.  Same kernels, same arguments: the weights after several pairs are bit-identical to the eagerly launched phases,
    with and without the pair hint."""
    from mr_gan_amd import MRGAN, engine as E
    from mr_gan_amd.dist import DataParallel, EngineBackend, dp_flags
    rs = np.random.RandomState(3)
    B, D, n = 64, 40, 256
    X = rs.randn(n, D).astype(np.float32)
    xl = rs.randn(2 * B, D).astype(np.float32)
    yl = rs.randint(0, 6, size=2 * B).astype(np.int32)
    res = []
    for graph in (False, True):
        m = MRGAN(D, batch_size=B, dtype='bfloat16', seed=77, use_graph=False, flags=dp_flags(exact, graph=graph))
        dp = DataParallel(EngineBackend(m.engine), exact=exact)
        with m._on_stream():
            xu, xlab = m._dev(X), m._dev(xl)
            idx_lab = m._dev(np.arange(n, dtype=np.int32) % (2 * B), torch.int32)
            labs = m._dev(yl[np.arange(n) % (2 * B)], torch.int32)
            idx_u1 = m._dev(np.arange(n, dtype=np.int32)[::-1].copy(), torch.int32)
            idx_u2 = m._dev(np.roll(np.arange(n, dtype=np.int32), 7), torch.int32)
            dargs = E.Engine.disc_args(xlab, labs, xu, None, idx_lab, idx_u1, stream_mode=1)
            gargs = E.Engine.gen_args(xu, None, idx_u2, stream_mode=1)
            for epoch in range(3):                             # the batch counter is rewound: the same graphs serve every epoch
                m.engine.set_iterations(2 * epoch * (n // B), 0)
                for _ in range(n // B):
                    dp.train_pair(dargs, gargs)
            torch.cuda.synchronize()
            metrics = m.engine.read_metrics(reset=True)
        res.append((m.get_weights('discriminator') + m.get_weights('generator'), metrics))
        m.engine.close()
    for a, b in zip(res[0][0], res[1][0]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(np.asarray(res[0][1]), np.asarray(res[1][1]))


def _dp_worker(rank, world, port, exact, q):
    """One rank of a real 2-process data-parallel run (gloo moves the GPU regions through the host; RCCL needs
    one GPU per rank, which the 1-GPU test box does not have)."""
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mr_gan_amd import engine as E
    from mr_gan_amd.dist import DataParallel, EngineBackend, dp_flags
    B, D, steps = 64, 32, 2
    case = Case(D=D, B=B, steps=steps, device_z=True)
    h = B // world
    eng = _engine(D, h, 0, flags=dp_flags(exact), rank=rank, world=world)
    _load(eng, case)
    dp = DataParallel(EngineBackend(eng), exact=exact)
    sl = slice(rank * h, (rank + 1) * h)
    for t in range(steps):
        dp.train_pair(E.Engine.disc_args(_t(case.x_lab[t][sl]), _t(case.labels[t][sl], torch.int32), _t(case.x_unl[t][sl])),
                      E.Engine.gen_args(_t(case.x_unl2[t][sl])))
    torch.cuda.synchronize()
    q.put((rank, eng.get_weights(E.NET_D), eng.get_weights(E.NET_G)))
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("exact", [True, False])
def test_two_process_data_parallel_on_one_gpu(exact):
    """mr_gan_amd/dist.py end to end with two OS processes sharing the GPU: replicas end bit-identical; with the
    statistic exchanges (exact) they also reproduce the single-process full-batch step."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, exact, q)) for r in range(2)]
    for p in procs:
        p.start()
    import queue as _queue
    out, waited = [], 0.0
    while len(out) < 2:
        try:
            out.append(q.get(timeout=2.0))
        except _queue.Empty:
            waited += 2.0
            assert all(p.exitcode in (None, 0) for p in procs), "a rank died: %s" % [p.exitcode for p in procs]
            assert waited < 180.0, "data-parallel ranks did not finish"
    out.sort(key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, d0, g0), (_, d1, g1) = out
    for a, b in zip(d0 + g0, d1 + g1):
        np.testing.assert_array_equal(a, b)
    case = Case(D=32, B=64, steps=2, device_z=True)
    ref = case.run_oracle()
    errs = [update_rel_err(w, wr, wi) for w, wr, wi in zip(d0, ref['d'], case.d0)]
    if exact:
        assert max(errs) < 0.05, errs
    else:
        assert max(errs) > 1e-4            # per-shard statistics: a different (labelled) algorithm


def test_run_scheduler_trains_concurrently_on_one_gpu():
    """scheduler.RunScheduler with two worker processes on cuda:0: four seeded mr_gan() trainings come back in job order and
    equal the same trainings run one after the other in this process."""
    from mr_gan_amd import synthetic_blobs
    from mr_gan_amd.mr_gan import mr_gan
    from mr_gan_amd.scheduler import RunScheduler
    X, y = synthetic_blobs(n=720, d=24, seed=5)
    rs = np.random.RandomState(0)
    jobs = []
    for i in range(4):
        perm = rs.permutation(720)
        jobs.append(dict(train_idx=perm[:600], test_idx=perm[600:], percentlabeled=4, epochs=2, seed=100 + i))
    want = [mr_gan(None, None, trainTestSets=[X[j['train_idx']], X[j['test_idx']], y[j['train_idx']], y[j['test_idx']]],
                   percentlabeled=4, epochs=2, seed=j["seed"]) for j in jobs]
    with RunScheduler(gpus=1, jobs_per_gpu=2) as sched:
        key = sched.put_dataset(X, y)
        got = sched.run([dict(dataset=key, **j) for j in jobs])
        assert len({w for _, w in sched.assignments}) == 2
    np.testing.assert_array_equal(np.asarray(got), np.asarray(want))
