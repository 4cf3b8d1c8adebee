"""CPU tests: golden fixtures vs the oracle, host-side data logic, and the C-ABI surface (load + exported
symbols + struct layout).  No compute call is made without a GPU."""
import ctypes
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

from oracle import mrgan_oracle as O
from tests.helpers import Case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


# ---- golden fixtures pin the oracle ------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["case_d16_b50.npz", "case_d400_b50_devz.npz"])
def test_oracle_reproduces_golden(name):
    g = np.load(os.path.join(GOLD, name))
    case = Case(D=int(g['D']), B=int(g['B']), steps=int(g['steps']), seed=int(g['seed']), device_z=bool(g['device_z']))
    ref = case.run_oracle()
    np.testing.assert_allclose(np.array(ref['disc']), g['disc'], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(np.array(ref['gen']), g['gen'], rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(ref['logits'], g['logits'], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(ref['d'][10], g['d_w6'], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(ref['g'][2], g['g_gamma'], rtol=1e-8, atol=1e-10)


def test_tiling_golden_and_host_function():
    from mr_gan_amd.data import tiled_permutation
    g = np.load(os.path.join(GOLD, "tiling.npz"))
    for ntr, nlab in ((6000, 480), (6000, 960), (7100, 6000)):
        want = g['tiling_%d_%d' % (ntr, nlab)]
        got = tiled_permutation(np.random.RandomState(1234), nlab, ntr)
        np.testing.assert_array_equal(got, want)
        assert got.max() < nlab and len(got) == ntr
        tail = ntr % nlab
        if tail:
            assert got[-tail:].max() < tail              # mr_gan.py:189 tail quirk


def test_prologue_matches_restatement():
    from mr_gan_amd.data import select_labeled, standard_scale
    rng = np.random.default_rng(1)
    X = rng.standard_normal((600, 9)) * 2 + 3
    y = np.arange(600) % 6
    Xte = rng.standard_normal((60, 9))
    a, b = standard_scale(X, Xte)
    a2, b2 = O.standard_scale(X, Xte)
    np.testing.assert_allclose(a, a2, atol=1e-12)
    np.testing.assert_allclose(b, b2, atol=1e-12)
    xl, yl, xu = select_labeled(a, y, 10, 5)
    xl2, yl2 = O.select_labeled(a2, y, 10)
    np.testing.assert_array_equal(xl, xl2)
    np.testing.assert_array_equal(yl, yl2)
    assert xu.shape == (90, 9) and np.array_equal(xu[:10], xl[:10])
    with pytest.raises(ValueError):
        select_labeled(a, y, 101)


def test_synthetic_generators_shapes():
    from mr_gan_amd.data import synthetic_blobs, synthetic_mreo
    X, y = synthetic_blobs(n=600, d=32)
    assert X.shape == (600, 32) and X.dtype == np.float32 and set(y) == set(range(6))
    X, y, obj = synthetic_mreo(d=40, objects_per_class=2, trials=5)
    assert X.shape == (60, 40) and len(set(obj)) == 12


# ---- harness keeps the reference's CLI and print format (mr_gan.py:236-261) ------------------------------------
def test_tables_harness_format(capsys):
    import importlib
    M = importlib.import_module('mr_gan_amd.mr_gan')     # the package re-exports the function under the same name
    calls = []

    def fake_dataset(modalities=0, **kw):
        rng = np.random.default_rng(modalities)
        return rng.standard_normal((72, 5)), np.arange(72) % 6

    def fake_mr_gan(X, y, percentlabeled=50, trainTestSets=None, **kw):
        calls.append((percentlabeled, trainTestSets[0].shape, trainTestSets[1].shape))
        return 0.25
    M.main(['--tables', '1'], dataset_fn=fake_dataset, mr_gan_fn=fake_mr_gan)
    out = capsys.readouterr().out
    assert len(calls) == 7 * 7 * 6                          # modalities x percents x folds
    assert calls[0][1] == (60, 5) and calls[0][2] == (12, 5)
    assert 'Test error: 0.25 Test accuracy: 0.75' in out
    assert 'Average error: 0.25 Average accuracy: 0.75' in out
    assert 'Percentage of training data labeled: 16%' in out
    with pytest.raises(SystemExit):
        M.main([], dataset_fn=fake_dataset, mr_gan_fn=fake_mr_gan)      # --tables is required (mr_gan.py:240)


# ---- C ABI surface ------------------------------------------------------------------------------------------
def _declared_symbols():
    """every function declared in include/*.h (the drop-in boundary mrgan_abi.h and the diagnostics of mrgan_debug.h)"""
    src = open(os.path.join(ROOT, "include", "mrgan_abi.h")).read() + open(os.path.join(ROOT, "include", "mrgan_debug.h")).read()
    return sorted(set(re.findall(r"\b(mrgan_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from mr_gan_amd import engine as E
    lib = E.load_library()
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(E.EXPORTS) == names
    # the boundary header has no diagnostic entry point, and nothing in the library reads the environment
    assert "mrgan_debug_" not in open(os.path.join(ROOT, "include", "mrgan_abi.h")).read()
    for f in os.listdir(os.path.join(ROOT, "mr_gan_amd", "csrc")):
        if f.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(ROOT, "mr_gan_amd", "csrc", f)).read(), f


def test_struct_layout_matches_header():
    from mr_gan_amd import engine as E
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "mrgan_abi.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu\n", sizeof(mrgan_config), offsetof(mrgan_config, sigma), offsetof(mrgan_config, seed),
         offsetof(mrgan_config, flags), sizeof(mrgan_disc_args), sizeof(mrgan_gen_args));
  return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(prog)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", os.path.join(d, "t")])
        vals = [int(v) for v in subprocess.check_output([os.path.join(d, "t")]).split()]
    assert vals == [ctypes.sizeof(E.Config), E.Config.sigma.offset, E.Config.seed.offset, E.Config.flags.offset,
                    ctypes.sizeof(E.DiscArgs), ctypes.sizeof(E.GenArgs)]


def test_default_config_and_workspace_size_without_gpu():
    from mr_gan_amd import engine as E
    cfg = E.default_config(512, 4096)
    assert (cfg.noise_size, list(cfg.g_hidden), list(cfg.d_hidden), cfg.num_classes) == (100, [500, 500], [1000, 500, 250, 250, 250], 6)
    assert abs(cfg.lr - 0.0006) < 1e-9 and abs(cfg.beta1 - 0.5) < 1e-9 and abs(cfg.bn_eps - 2e-5) < 1e-12
    n = ctypes.c_size_t()
    lib = E.load_library()
    for dtype, lo, hi in ((E.F32, 200e6, 900e6), (E.BF16, 100e6, 600e6)):
        cfg.dtype = dtype
        assert lib.mrgan_workspace_bytes(ctypes.byref(cfg), ctypes.byref(n)) == 0
        assert lo < n.value < hi
    cfg.num_classes = 40
    assert lib.mrgan_workspace_bytes(ctypes.byref(cfg), ctypes.byref(n)) != 0
    assert b"num_classes" in lib.mrgan_last_error()


def test_product_path_never_imports_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline leg may use oracle/ (mentions in comments are fine)."""
    pkg = os.path.join(ROOT, "mr_gan_amd")
    pat = re.compile(r"^\s*(from\s+oracle|import\s+oracle|#\s*include\s*[\"<][^\">]*oracle)|import_module\(['\"]oracle|dlopen\([^)]*oracle", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                assert not pat.search(open(os.path.join(dirpath, f)).read()), f


def test_engine_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mr_gan_amd import engine as E
    with pytest.raises(RuntimeError):
        E.Engine(E.default_config(16, 50))


def test_bench_algorithmic_flops_match_survey():
    """D-step 2B[P_G + 3P_D + 3P_D + 3(P_D - 1000D)], G-step 2B[P_G + 2P_mid + P_mid + P_G + (P_G - 50000)];
    85.13 + 44.27 = 129.40 GFLOP per step at B = 4096, D = 512 -- the figure bench.py's roofline.step uses."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    B, D = 4096, 512
    fl = bench.algorithmic_flops(B, D)
    PD = 1000 * D + 751500
    Pmid = PD - 1500
    PG = 300000 + 500 * D
    d_step = 2.0 * B * (PG + 3 * PD + 3 * PD + 3 * (PD - 1000 * D))
    g_step = 2.0 * B * (PG + 2 * Pmid + Pmid + PG + (PG - 50000))
    assert abs(fl["total"] - (d_step + g_step)) < 1e-6 * fl["total"]
    assert abs(fl["total"] / 1e9 - 129.40) < 0.01
    assert abs(d_step / 1e9 - 85.13) < 0.01 and abs(g_step / 1e9 - 44.27) < 0.01
