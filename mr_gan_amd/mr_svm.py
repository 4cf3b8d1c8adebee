"""SVM baseline of the reference on the same harness (mr_svm.py:77-116 and its --tables 2 4 loops, :119-166):
RBF-kernel SVC (C = 1) trained on the labeled subset only, stays on scikit-learn / CPU as in the reference.

    python -m mr_gan_amd.mr_svm --tables 2 4 [-v]

Shares dataset(), the data prologue and the modality names with mr_gan_amd.mr_gan; prints the reference's lines
(mr_svm.py prints only the averages for table 2 and one line per object for table 4)."""
import argparse
import itertools
import sys

import numpy as np

from mr_gan_amd.data import MATERIALS, select_labeled, standard_scale
from mr_gan_amd.mr_gan import MODALITIES, dataset


SVC_GAMMA = 'auto'


def mr_svm(X, y, percentlabeled=50, trainTestSets=None, verbose=False, seed=None):
    from sklearn.model_selection import train_test_split
    from sklearn.svm import SVC
    from sklearn.utils import shuffle
    rs = np.random.RandomState(seed if seed is not None else np.random.randint(1 << 31))     # mr_svm.py:79 is unseeded
    test_ratio = 200 * len(MATERIALS)                              # mr_svm.py:82
    num_labeled_examples = int(10 * percentlabeled)                # mr_svm.py:83
    if trainTestSets is None:                                      # mr_svm.py:86-89
        X_train, X_test, y_train, y_test = train_test_split(X, y, test_size=test_ratio, stratify=y, random_state=rs)
    else:
        X_train, X_test, y_train, y_test = trainTestSets
    if verbose:
        print('Num of class examples in test set:', [int(np.sum(y_test == i)) for i in range(len(MATERIALS))])
        print('X_train:', np.shape(X_train), 'y_train:', np.shape(y_train), 'X_test:', np.shape(X_test), 'y_test:', np.shape(y_test))
    X_train, X_test = standard_scale(X_train, X_test)              # mr_svm.py:95-97
    X_train, y_train = shuffle(X_train, y_train, random_state=rs)  # mr_svm.py:100
    x_labeled, y_labeled, _ = select_labeled(X_train, y_train, num_labeled_examples)
    if verbose:
        print('x_labeled:', np.shape(x_labeled), 'y_labeled:', np.shape(y_labeled))
    # mr_svm.py:106 is SVC(kernel='rbf', C=1.0) under the scikit-learn of 2017 (Keras 2.0.9 era, README.md:43-48), whose default
    # kernel width was gamma='auto' = 1 / n_features.  scikit-learn >= 0.22 defaults to 'scale' (1 / (n_features * X.var())),
    # which differs on the small standardised labeled subset: pass the reference's value explicitly.
    svm = SVC(kernel='rbf', C=1.0, gamma=SVC_GAMMA)
    svm.fit(x_labeled, y_labeled)
    testerror = 1.0 - svm.score(X_test, y_test)                    # mr_svm.py:110
    if verbose:
        print('Test error:', testerror, 1.0 - np.mean(svm.predict(X_test) == y_test))
        sys.stdout.flush()
    return testerror


def baseline_tables(tables, fn, dataset_fn, verbose=False):
    """The --tables 2 4 loops shared by mr_svm.py:126-166 and mr_nn.py:128-168 (identical up to the function called)."""
    from sklearn.model_selection import StratifiedKFold
    if '2' in tables:
        print('\n', '-' * 25, 'Testing various amounts of labeled training data', '-' * 25)
        print('-' * 100)
        for modality in [2, 5]:
            print('-' * 25, MODALITIES[modality], 'modality', '-' * 25)
            X, y = dataset_fn(modalities=modality)
            for percent in [1, 2, 4, 8, 16, 50, 100]:
                print('-' * 15, 'Percentage of training data labeled: %d%%' % percent, '-' * 15)
                errors = []
                skf = StratifiedKFold(n_splits=6, shuffle=True)
                for trainIdx, testIdx in skf.split(X, y):
                    errors.append(fn(None, None, percentlabeled=percent, trainTestSets=[X[trainIdx], X[testIdx], y[trainIdx], y[testIdx]],
                                     verbose=verbose))
                    sys.stdout.flush()
                print('Average error:', np.mean(errors), 'Average accuracy:', np.mean(1.0 - np.array(errors)))
                sys.stdout.flush()
    if '4' in tables:
        print('\n', '-' * 25, 'Testing generalization with leave-one-object-out validation', '-' * 25)
        print('-' * 100)
        for modality in [2, 5]:
            print('-' * 25, MODALITIES[modality], 'modality', '-' * 25)
            objects = dataset_fn(modalities=modality, leaveObjectOut=True)
            for percent in [1, 4, 16, 50, 100]:
                print('-' * 15, 'Percentage of training data labeled: %d%%' % percent, '-' * 15)
                errors = []
                for objName, objData in objects.items():
                    Xtest, ytest = np.array(objData['x']), np.array(objData['y'])
                    Xtrain = np.array(list(itertools.chain.from_iterable([d['x'] for n, d in objects.items() if n != objName])))
                    ytrain = np.array(list(itertools.chain.from_iterable([d['y'] for n, d in objects.items() if n != objName])))
                    errors.append(fn(None, None, percentlabeled=percent, trainTestSets=[Xtrain, Xtest, ytrain, ytest], verbose=verbose))
                    print(objName, 'Test error:', errors[-1], 'Test accuracy:', 1.0 - errors[-1])
                    sys.stdout.flush()
                print('Average leave-one-object-out error:', np.mean(errors), 'Average accuracy:', np.mean(1.0 - np.array(errors)))
                sys.stdout.flush()


def main(argv=None, dataset_fn=dataset, fn=mr_svm):
    parser = argparse.ArgumentParser(description='RBF-SVM baseline for material recognition on haptic data.')
    parser.add_argument('-t', '--tables', nargs='+', help='[Required] Tables to recompute', required=True)
    parser.add_argument('-v', '--verbose', help='Verbose', action='store_true')
    args = parser.parse_args(argv)
    baseline_tables(args.tables, fn, dataset_fn, args.verbose)


if __name__ == '__main__':
    main()
