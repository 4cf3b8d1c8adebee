"""Host-side data logic of mr_gan(): scaling, labeled-subset selection, epoch index streams, and the
synthetic stand-ins for the (absent) MREO dataset described in SURVEY.md section 8(d)."""
import numpy as np

MATERIALS = ['plastic', 'glass', 'fabric', 'metal', 'wood', 'ceramic']     # mr_gan.py:24, :80


def standard_scale(X_train, X_test):
    """mr_gan.py:96-98 -- StandardScaler fit on train, applied to test."""
    from sklearn import preprocessing
    scaler = preprocessing.StandardScaler()
    X_train = scaler.fit_transform(X_train)
    X_test = scaler.transform(X_test)
    return X_train, X_test


def select_labeled(X_train, y_train, num_labeled, num_unlabeled=None, num_classes=len(MATERIALS)):
    """mr_gan.py:102-107 -- the first num_labeled rows of every class (class-sorted block matrix);
    with num_unlabeled also the table-6 unlabeled pool (labeled rows included)."""
    for j in range(num_classes):
        if np.sum(y_train == j) < num_labeled:
            raise ValueError("class %d has fewer than %d training rows" % (j, num_labeled))
    x_labeled = np.concatenate([X_train[y_train == j][:num_labeled] for j in range(num_classes)], axis=0)
    y_labeled = np.concatenate([[j] * num_labeled for j in range(num_classes)], axis=0)
    x_unlabeled = None
    if num_unlabeled is not None:
        x_unlabeled = np.concatenate([X_train[y_train == j][:num_labeled + num_unlabeled] for j in range(num_classes)], axis=0)
    return x_labeled, y_labeled, x_unlabeled


def tiled_permutation(rng, n_pool, n_total):
    """mr_gan.py:189 -- floor(n_total/n_pool) permutations of the pool followed by
    permutation(n_total % n_pool): the tail only ever indexes the first n_total % n_pool pool rows."""
    parts = [rng.permutation(n_pool) for _ in range(n_total // n_pool)] + [rng.permutation(n_total % n_pool)]
    return np.concatenate(parts).astype(np.int32)


def synthetic_blobs(n=65536, d=512, num_classes=6, seed=1234):
    """BASELINE config 2: y = arange(N) % K; class centres ~ N(0,1); X = C[y] + N(0,1)."""
    rng = np.random.default_rng(seed)
    y = (np.arange(n) % num_classes).astype(np.int32)
    centres = rng.standard_normal((num_classes, d)).astype(np.float32)
    X = centres[y] + rng.standard_normal((n, d), dtype=np.float32)
    return X, y


def synthetic_mreo(d=1200, objects_per_class=12, trials=100, num_classes=6, seed=20171113, sep=1.0):
    """MREO-shaped surrogate: per-class smooth template + per-object offset +
    per-trial noise; N = classes x objects x trials.  Returns X, y, object_id."""
    rng = np.random.default_rng(seed)
    X, y, obj = [], [], []
    for c in range(num_classes):
        template = np.cumsum(rng.standard_normal(d)) / np.sqrt(d) * sep
        for o in range(objects_per_class):
            offset = 0.3 * rng.standard_normal(d)
            X.append(template + offset + rng.standard_normal((trials, d)))
            y += [c] * trials
            obj += [c * objects_per_class + o] * trials
    return np.concatenate(X).astype(np.float64), np.array(y), np.array(obj)
