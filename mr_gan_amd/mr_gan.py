"""Drop-in entry points of the reference script: mr_gan(), dataset() and the --tables harness.

    mr_gan(X, y, percentlabeled=50, percentunlabeled=None, epochs=100, trainTestSets=None, verbose=False) -> test error
        same signature and return value as mr_gan.py:73 / :234
    dataset(modalities=0, forcetempTime=4, contactmicTime=0.2, leaveObjectOut=False, verbose=False)
        same signature as mr_gan.py:23
    python -m mr_gan_amd.mr_gan --tables 1 3 5 6 [-v]        (mr_gan.py:236-341)
    python -m mr_gan_amd.mr_gan --tables 1 3 5 6 --gpus 8 --jobs-per-gpu 2     run-level scheduling of the tables (scheduler.py)

Extra keyword arguments (batch_size, dtype, seed, device) default to the reference's literals.
"""
import argparse
import itertools
import os
import pickle
import sys

import numpy as np

from mr_gan_amd.data import MATERIALS, select_labeled, standard_scale

MODALITIES = ['Force', 'Temperature', 'Force and Temperature', 'Contact mic', 'Temperature and Contact Mic',
              'Force, Temperature, and Contact Mic', 'Force and Contact Mic']          # mr_gan.py:237


def mr_gan(X, y, percentlabeled=50, percentunlabeled=None, epochs=100, trainTestSets=None, verbose=False,
           batch_size=50, dtype='float32', seed=None, device='cuda:0'):
    from sklearn.model_selection import train_test_split
    from sklearn.utils import shuffle

    from mr_gan_amd.model import MRGAN

    rs = np.random.RandomState(seed if seed is not None else np.random.randint(1 << 31))   # mr_gan.py:75 (unseeded there)
    test_ratio = 200 * len(MATERIALS)                              # mr_gan.py:81
    num_labeled_examples = int(10 * percentlabeled)                # mr_gan.py:82
    num_unlabeled_examples = int(10 * percentunlabeled) if percentunlabeled is not None else None

    if trainTestSets is None:                                      # mr_gan.py:87-90
        X_train, X_test, y_train, y_test = train_test_split(X, y, test_size=test_ratio, stratify=y, random_state=rs)
    else:
        X_train, X_test, y_train, y_test = trainTestSets
    if verbose:
        print('Num of class examples in test set:', [int(np.sum(y_test == i)) for i in range(len(MATERIALS))])
        print('X_train:', np.shape(X_train), 'y_train:', np.shape(y_train), 'X_test:', np.shape(X_test), 'y_test:',
              np.shape(y_test))

    X_train, X_test = standard_scale(X_train, X_test)              # mr_gan.py:96-98
    X_train, y_train = shuffle(X_train, y_train, random_state=rs)  # mr_gan.py:101
    x_labeled, y_labeled, x_unlabeled = select_labeled(X_train, y_train, num_labeled_examples, num_unlabeled_examples)
    if verbose:
        print('x_labeled:', np.shape(x_labeled), 'y_labeled:', np.shape(y_labeled))

    model = MRGAN(X_train.shape[1], batch_size=batch_size, dtype=dtype, seed=int(rs.randint(1 << 31)), device=device)
    if verbose:
        print('Epochs:', epochs)
        print('Batch size:', batch_size)
        print('Training batches per epoch:', X_train.shape[0] // batch_size)
        print('Testing batches per epoch:', X_test.shape[0] // batch_size)
    hist = model.fit(x_labeled, y_labeled, X_train, epochs=epochs, verbose=1 if verbose else 0,
                     validation_data=(X_test, y_test), x_unlabeled_pool=x_unlabeled, rng=rs)
    testerror = model.evaluate(X_test, y_test)                     # mr_gan.py:230 -- whole test set in one call
    if verbose:
        print('Test error:', testerror, hist[-1]['test_err'] if hist else float('nan'))
        sys.stdout.flush()
    model.engine.close()
    return testerror


def _logmel_all(contacts, sr=48000, n_mels=128):
    """mr_gan.py:42-47 for every trial of the data set in one GPU launch per signal length (mr_gan_amd/melspec.py)"""
    from mr_gan_amd.melspec import log_melspectrogram_batch
    return log_melspectrogram_batch(contacts, sr=sr, n_mels=n_mels)


def dataset(modalities=0, forcetempTime=4, contactmicTime=0.2, leaveObjectOut=False, verbose=False,
            data_dir='data_processed', logmel_fn=_logmel_all):
    """mr_gan.py:23-71.  Reads the MREO pickles written by processdata.py (not shipped with the reference;
    README.md:7-11) and concatenates the modality vectors in the reference's fixed order.  The log-mel spectrograms of the
    contact-microphone modalities are computed for all trials at once on the GPU after the files are read."""
    trials = []                                                  # (object name, material, temperature, force0, force1, contact)
    for m, material in enumerate(MATERIALS):
        if verbose:
            print('Processing', material)
            sys.stdout.flush()
        fname = os.path.join(data_dir, 'processed_0.1sbefore_%s_times_%.2f_%.2f.pkl' % (material, forcetempTime, contactmicTime))
        with open(fname, 'rb') as f:
            allData = pickle.load(f, encoding='latin1')          # py2 cPickle files (others/mr_nn_activation_map_py3.py:33)
        for objName, objData in allData.items():
            for i in range(len(objData['temperature'])):
                trials.append((objName, m, list(objData['temperature'][i]), list(objData['force0'][i]), list(objData['force1'][i]),
                               objData['contact'][i] if modalities > 2 else None))
    logs = logmel_fn([t[5] for t in trials]) if modalities > 2 else None     # mr_gan.py:42-47
    X, y = [], []
    objects = dict()
    for n, (objName, m, temp, f0, f1, _) in enumerate(trials):
        if leaveObjectOut:
            if objName not in objects:
                objects[objName] = {'x': [], 'y': []}
            X, y = objects[objName]['x'], objects[objName]['y']
        y.append(m)
        log_S = logs[n] if logs is not None else None
        if modalities == 0:
            X.append(f0 + f1)
        elif modalities == 1:
            X.append(temp)
        elif modalities == 2:
            X.append(temp + f0 + f1)
        elif modalities == 3:
            X.append(log_S.flatten())
        elif modalities == 4:
            X.append(temp + log_S.flatten().tolist())
        elif modalities == 5:
            X.append(temp + f0 + f1 + log_S.flatten().tolist())
        elif modalities == 6:
            X.append(f0 + f1 + log_S.flatten().tolist())
    if leaveObjectOut:
        return objects
    X = np.array(X)
    y = np.array(y)
    if verbose:
        print('X:', np.shape(X), 'y:', np.shape(y))
    return X, y


class InProcessRuns(object):
    """The scheduler's interface (put_dataset / run) executed in this process, one training after the other: what
    `--gpus 0` (the default, the reference's own behaviour) uses, so that every table has ONE set of job builders."""

    def __init__(self, mr_gan_fn):
        self.fn, self.datasets = mr_gan_fn, {}

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def put_dataset(self, X, y):
        self.datasets[len(self.datasets)] = (X, y)
        return len(self.datasets) - 1

    def run(self, jobs):
        out = []
        for job in jobs:
            kw = {k: v for k, v in job.items() if k not in ('dataset', 'train_idx', 'test_idx')}
            X, y = self.datasets[job['dataset']]
            tr, te = job['train_idx'], job['test_idx']
            out.append(self.fn(None, None, trainTestSets=[X[tr], X[te], y[tr], y[te]], **kw))
        return out


def _kfold_jobs(X, y, key, **job_kw):
    """The six stratified folds of _kfold as scheduler jobs (row indices into the dataset already shipped as `key`)."""
    from sklearn.model_selection import StratifiedKFold
    skf = StratifiedKFold(n_splits=6, shuffle=True)              # mr_gan.py:255
    return [dict(dataset=key, train_idx=tr, test_idx=te, **job_kw) for tr, te in skf.split(X, y)]


def _print_kfold(errors):
    for e in errors:
        print('Test error:', e, 'Test accuracy:', 1.0 - e)
    print('Average error:', np.mean(errors), 'Average accuracy:', np.mean(1.0 - np.array(errors)))
    sys.stdout.flush()


def _table1_scheduled(sched, dataset_fn, kw):
    """Table 1 with the 42 trainings of a modality (7 label fractions x 6 folds) dispatched together; prints the
    reference's lines in the reference's order (mr_gan.py:244-261)."""
    print('\n', '-' * 25, 'Testing various amounts of labeled training data', '-' * 25)
    print('-' * 100)
    percents = [1, 2, 4, 8, 16, 50, 100]
    for modality in range(len(MODALITIES)):
        print('-' * 25, MODALITIES[modality], 'modality', '-' * 25)
        X, y = dataset_fn(modalities=modality)
        key = sched.put_dataset(X, y)
        jobs = []
        for percent in percents:
            jobs += _kfold_jobs(X, y, key, percentlabeled=percent, **kw)
        errors = sched.run(jobs)
        for i, percent in enumerate(percents):
            print('-' * 15, 'Percentage of training data labeled: %d%%' % percent, '-' * 15)
            _print_kfold(errors[6 * i:6 * i + 6])


def _table3_scheduled(sched, dataset_fn, kw):
    """Table 3 (mr_gan.py:263-283): leave-one-object-out, 2 modalities x 5 label fractions x one training per object; every
    training of a modality is dispatched together, the reference's lines are printed in the reference's order."""
    print('\n', '-' * 25, 'Testing generalization with leave-one-object-out validation', '-' * 25)
    print('-' * 100)
    percents = [1, 4, 16, 50, 100]
    for modality in [2, 5]:
        print('-' * 25, MODALITIES[modality], 'modality', '-' * 25)
        objects = dataset_fn(modalities=modality, leaveObjectOut=True)
        names = list(objects.keys())
        # the objects' rows once, in the reference's order (mr_gan.py:277-278 chains them in dict order): a job is two index
        # vectors into that matrix instead of its own copy of ~7000 training rows
        Xall = np.array(list(itertools.chain.from_iterable(objects[n]['x'] for n in names)))
        yall = np.array(list(itertools.chain.from_iterable(objects[n]['y'] for n in names)))
        owner = np.concatenate([np.full(len(objects[n]['y']), i) for i, n in enumerate(names)])
        key = sched.put_dataset(Xall, yall)
        jobs = []
        for percent in percents:
            for i in range(len(names)):
                jobs.append(dict(dataset=key, train_idx=np.flatnonzero(owner != i), test_idx=np.flatnonzero(owner == i),
                                 percentlabeled=percent, **kw))
        errors = sched.run(jobs)
        for i, percent in enumerate(percents):
            print('-' * 15, 'Percentage of training data labeled: %d%%' % percent, '-' * 15)
            errs = errors[i * len(names):(i + 1) * len(names)]
            for objName, e in zip(names, errs):
                print(objName, 'Test error:', e, 'Test accuracy:', 1.0 - e)
            print('Average leave-one-object-out error:', np.mean(errs), 'Average accuracy:', np.mean(1.0 - np.array(errs)))
            sys.stdout.flush()


def _table5_scheduled(sched, dataset_fn, kw):
    """Table 5 (mr_gan.py:285-318): 28 (modality, contact time) data sets x 6 folds at 100 % labeled"""
    for block, combos, fmt in ((0, [(m, dict(forcetempTime=t)) for m in range(3) for t in [4, 3, 2, 1, 0.5, 0.2, 0.1]], None),
                               (1, [(3, dict(contactmicTime=t)) for t in [1, 0.7, 0.5, 0.3, 0.2, 0.1, 0.05]], None)):
        print('\n', '-' * 25, 'Testing various lengths of contact time in training data', '-' * 25)
        print('-' * 100)
        last_mod = None
        for modality, dkw in combos:
            if modality != last_mod:
                print('-' * 25, MODALITIES[modality], 'modality', '-' * 25)
                last_mod = modality
            print('-' * 15, 'Length of training data: %.1fs' % list(dkw.values())[0], '-' * 15)
            X, y = dataset_fn(modalities=modality, **dkw)
            key = sched.put_dataset(X, y)
            _print_kfold(sched.run(_kfold_jobs(X, y, key, percentlabeled=100, **kw)))


def _table6_scheduled(sched, dataset_fn, kw):
    """Table 6 (mr_gan.py:320-341): 2 modalities x 7 amounts of unlabeled data x 6 folds at 4 % labeled"""
    print('\n', '-' * 25, 'Testing performance as quantity of unlabeled data increases', '-' * 25)
    print('-' * 100)
    for modality in [2, 5]:
        print('-' * 25, MODALITIES[modality], 'modality', '-' * 25)
        X, y = dataset_fn(modalities=modality)
        key = sched.put_dataset(X, y)
        for percentlabeled in [4]:
            print('-' * 15, 'Percentage of training data labeled: %d%%' % percentlabeled, '-' * 15)
            unl = [0, 4, 8, 16, 32, 64, 100 - percentlabeled]
            jobs = []
            for percentunlabeled in unl:
                jobs += _kfold_jobs(X, y, key, percentlabeled=percentlabeled, percentunlabeled=percentunlabeled, **kw)
            errors = sched.run(jobs)
            for i, percentunlabeled in enumerate(unl):
                print('-' * 15, 'Percentage of training data unlabeled: %d%%' % percentunlabeled, '-' * 15)
                _print_kfold(errors[6 * i:6 * i + 6])


def main(argv=None, dataset_fn=dataset, mr_gan_fn=mr_gan, scheduler_factory=None):
    parser = argparse.ArgumentParser(description='Semi-supervised learning with GANs for material recognition on haptic data.')
    parser.add_argument('-t', '--tables', nargs='+', help='[Required] Tables to recompute', required=True)
    parser.add_argument('-v', '--verbose', help='Verbose', action='store_true')
    parser.add_argument('--epochs', type=int, default=100)
    parser.add_argument('--dtype', default='float32')
    parser.add_argument('--gpus', type=int, default=0,
                        help='run-level scheduling: dispatch the independent trainings of tables 1 / 3 / 5 / 6 over this many GPUs (0 = in-process, sequential)')
    parser.add_argument('--jobs-per-gpu', type=int, default=1, help='concurrent trainings per GPU with --gpus')
    args = parser.parse_args(argv)
    kw = dict(epochs=args.epochs, dtype=args.dtype)
    if args.gpus > 0:                                              # the independent trainings dispatched over worker processes
        from mr_gan_amd.scheduler import RunScheduler
        runs = (scheduler_factory or RunScheduler)(gpus=args.gpus, jobs_per_gpu=args.jobs_per_gpu)
    else:                                                          # the reference's behaviour: one after the other, here
        runs = InProcessRuns(mr_gan_fn)
        kw['verbose'] = args.verbose
    with runs as sched:
        if '1' in args.tables:                                     # mr_gan.py:244-261
            _table1_scheduled(sched, dataset_fn, kw)
        if '3' in args.tables:                                     # mr_gan.py:263-283
            _table3_scheduled(sched, dataset_fn, kw)
        if '5' in args.tables:                                     # mr_gan.py:285-318
            _table5_scheduled(sched, dataset_fn, kw)
        if '6' in args.tables:                                     # mr_gan.py:320-341
            _table6_scheduled(sched, dataset_fn, kw)


if __name__ == '__main__':
    main()
