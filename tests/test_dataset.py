"""dataset() / log-mel front end on a synthetic MREO-format pickle.
The real dataset is a download (README.md:7-11) and the reference's librosa is absent: parity unpinned; these tests pin
the file format handling (processdata.py:23-34, :91; mr_gan.py:32-62) and the basic properties of the mel front end."""
import os
import pickle

import numpy as np

from mr_gan_amd import dataset
from mr_gan_amd.data import MATERIALS
from oracle.melspec_oracle import log_melspectrogram, mel_filterbank


def _oracle_logmel(contacts, sr=48000, n_mels=128):
    """CPU stand-in for the GPU front end (this suite runs without a GPU; tests/test_gpu_parity.py holds the HIP kernel to it)"""
    return [log_melspectrogram(np.asarray(c, dtype=np.float64), sr=sr, n_mels=n_mels).astype(np.float32).flatten() for c in contacts]


def _write_fake_mreo(tmp, ft=4, cm=0.2, objects=2, trials=3):
    rng = np.random.default_rng(0)
    for m, material in enumerate(MATERIALS):
        allData = {}
        for o in range(objects):
            d = {k: [] for k in ('forceTime', 'force0', 'force1', 'pressureTime', 'pressure0', 'pressure1',
                                 'temperatureTime', 'temperature', 'contactTime', 'contact')}
            for t in range(trials):
                n = int(100 * ft)
                d['force0'].append((rng.standard_normal(n) + 10 * m).tolist())
                d['force1'].append((rng.standard_normal(n) + 20 * m).tolist())
                d['temperature'].append((rng.standard_normal(n) + 30 * m).tolist())
                d['contact'].append(rng.standard_normal(int(48000 * cm)).tolist())
            allData['%s_obj%d' % (material, o)] = d
        with open(os.path.join(tmp, 'processed_0.1sbefore_%s_times_%.2f_%.2f.pkl' % (material, ft, cm)), 'wb') as f:
            pickle.dump(allData, f, 2)          # protocol 2 = what Python-2 cPickle.HIGHEST_PROTOCOL wrote


def test_dataset_modalities_and_order(tmp_path):
    _write_fake_mreo(str(tmp_path))
    mel = 128 * 19
    want = {0: 800, 1: 400, 2: 1200, 3: mel, 4: 400 + mel, 5: 1200 + mel, 6: 800 + mel}
    for mod, d in want.items():
        X, y = dataset(modalities=mod, data_dir=str(tmp_path), logmel_fn=_oracle_logmel)
        assert X.shape == (6 * 2 * 3, d) and y.shape == (36,)
        assert list(np.unique(y)) == list(range(6))
    X, y = dataset(modalities=2, data_dir=str(tmp_path))      # temperature | force0 | force1 (mr_gan.py:54)
    r = X[y == 3][0]
    assert abs(r[:400].mean() - 90) < 1 and abs(r[400:800].mean() - 30) < 1 and abs(r[800:].mean() - 60) < 1
    objs = dataset(modalities=0, leaveObjectOut=True, data_dir=str(tmp_path))
    assert len(objs) == 12 and np.array(objs['glass_obj1']['x']).shape == (3, 800)
    # the contact-microphone block is the flattened [128 mels][19 frames] matrix of that trial (mr_gan.py:47, :57-62)
    X, y = dataset(modalities=5, data_dir=str(tmp_path), logmel_fn=_oracle_logmel)
    with open(os.path.join(str(tmp_path), 'processed_0.1sbefore_plastic_times_4.00_0.20.pkl'), 'rb') as f:
        first = pickle.load(f)['plastic_obj0']
    np.testing.assert_allclose(X[0][1200:], _oracle_logmel([first['contact'][0]])[0], rtol=0, atol=0)
    np.testing.assert_allclose(X[0][:400], first['temperature'][0])


def test_dataset_without_a_gpu_fails_loudly(tmp_path):
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    _write_fake_mreo(str(tmp_path), objects=1, trials=1)
    with pytest.raises(RuntimeError, match="GPU only"):
        dataset(modalities=3, data_dir=str(tmp_path))


def test_log_mel_properties():
    sr = 48000
    t = np.arange(9600) / sr
    S = log_melspectrogram(np.sin(2 * np.pi * 3000 * t), sr=sr)
    assert S.shape == (128, 19)                                 # 1 + 9600 // 512 frames
    assert S.max() == 0.0 and S.min() >= -80.0
    fb = mel_filterbank(sr, 2048)
    centre = np.argmax(fb[:, int(round(3000 / (sr / 2048)))])
    assert abs(int(np.argmax(S.mean(axis=1))) - centre) <= 1    # energy lands in the 3 kHz band
    assert np.all(fb >= 0) and fb.shape == (128, 1025)


def test_svm_baseline_and_its_tables(capsys):
    """mr_svm (mr_svm.py:77-116): RBF SVC on the labeled subset, and the --tables 2 4 loops with the reference's lines."""
    from mr_gan_amd.mr_svm import main, mr_svm
    from mr_gan_amd import synthetic_mreo
    from sklearn.svm import SVC
    from sklearn import preprocessing
    from sklearn.utils import shuffle
    X, y, _ = synthetic_mreo(d=40, trials=20, sep=3.0)
    tr = np.arange(len(y)) % 6 != 0
    sets = [X[tr], X[~tr], y[tr], y[~tr]]
    got = mr_svm(None, None, percentlabeled=2, trainTestSets=sets, seed=3)
    # literal transcription of mr_svm.py:94-110 with the same shuffle state
    sc = preprocessing.StandardScaler()
    a, b = sc.fit_transform(sets[0]), sc.transform(sets[1])
    a, ya = shuffle(a, sets[2], random_state=np.random.RandomState(3))
    xl = np.concatenate([a[ya == j][:20] for j in range(6)])
    yl = np.concatenate([[j] * 20 for j in range(6)])
    # gamma: the reference relies on the scikit-learn default of its time, 'auto' = 1 / n_features (mr_svm.py:106; the
    # installed scikit-learn would pick 'scale', another kernel width on this 120-row subset)
    import mr_gan_amd.mr_svm as S
    assert S.SVC_GAMMA == 'auto'
    want = 1.0 - SVC(kernel='rbf', C=1.0, gamma=1.0 / xl.shape[1]).fit(xl, yl).score(b, sets[3])
    assert got == want and got < 0.5

    rs = np.random.RandomState(0)

    def fake_dataset(modalities=0, leaveObjectOut=False, **kw):
        if leaveObjectOut:
            return {'m%d_o%d' % (m, o): {'x': rs.randn(4, 3).tolist(), 'y': [m] * 4} for m in range(6) for o in range(2)}
        return rs.randn(36, 3), np.arange(36) % 6
    calls = []
    main(['--tables', '2', '4'], dataset_fn=fake_dataset, fn=lambda X, y, **kw: calls.append(kw['percentlabeled']) or 0.25)
    out = capsys.readouterr().out
    assert len(calls) == 2 * 7 * 6 + 2 * 5 * 12
    assert out.count('Average error: 0.25 Average accuracy: 0.75') == 14
    assert out.count('Average leave-one-object-out error: 0.25') == 10 and 'm0_o0 Test error: 0.25 Test accuracy: 0.75' in out


def test_nn_baseline_tables_run_the_reference_loops(capsys):
    """mr_nn --tables 2 4 (mr_nn.py:128-168): the same loops and lines as mr_svm's, around the HIP-engine mr_nn()"""
    from mr_gan_amd.mr_nn import main
    rs = np.random.RandomState(0)

    def fake_dataset(modalities=0, leaveObjectOut=False, **kw):
        if leaveObjectOut:
            return {'m%d_o%d' % (m, o): {'x': rs.randn(4, 3).tolist(), 'y': [m] * 4} for m in range(6) for o in range(2)}
        return rs.randn(36, 3), np.arange(36) % 6
    calls = []
    main(['--tables', '4'], dataset_fn=fake_dataset, fn=lambda X, y, **kw: calls.append(kw['percentlabeled']) or 0.5)
    out = capsys.readouterr().out
    assert calls == [p for _ in range(2) for p in (1, 4, 16, 50, 100) for _ in range(12)]
    assert out.count('Average leave-one-object-out error: 0.5') == 10
