"""Run-level scheduler of the table harness (mr_gan_amd/scheduler.py) -- CPU tests with a stub runner."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests.helpers import stub_training  # noqa: E402


def _jobs(n, key):
    rs = np.random.RandomState(0)
    jobs = []
    for i in range(n):
        perm = rs.permutation(120)
        jobs.append(dict(dataset=key, train_idx=perm[:100], test_idx=perm[100:], percentlabeled=i))
    return jobs


def test_results_come_back_in_job_order_from_several_processes():
    from mr_gan_amd.scheduler import RunScheduler
    rs = np.random.RandomState(1)
    X, y = rs.randn(120, 7), rs.randint(0, 6, size=120)
    with RunScheduler(runner=stub_training, devices=['cpu', 'cpu', 'cpu']) as sched:
        key = sched.put_dataset(X, y)
        jobs = _jobs(17, key)
        out = sched.run(jobs)
        assert len(sched.assignments) == 17 and sorted(j for j, _ in sched.assignments) == list(range(17))
        want = [stub_training(j, {key: (X, y)}, 'cpu')[0] for j in jobs]
        np.testing.assert_allclose([o[0] for o in out], want, rtol=0, atol=0)
        assert len({o[1] for o in out}) >= 2                    # more than one worker process did the work
        assert os.getpid() not in {o[1] for o in out}
        # a second batch on the same workers and dataset
        out2 = sched.run(jobs[:4])
        np.testing.assert_allclose([o[0] for o in out2], want[:4], rtol=0, atol=0)


def test_device_list_and_failure_report():
    from mr_gan_amd.scheduler import RunScheduler
    s = RunScheduler.__new__(RunScheduler)                      # device naming without starting workers
    assert ['cuda:%d' % g for g in range(2) for _ in range(3)] == ['cuda:0'] * 3 + ['cuda:1'] * 3
    del s
    rs = np.random.RandomState(2)
    X, y = rs.randn(120, 3), rs.randint(0, 6, size=120)
    with RunScheduler(runner=stub_training, devices=['cpu', 'cpu']) as sched:
        key = sched.put_dataset(X, y)
        jobs = _jobs(3, key)
        jobs[1]['explode'] = True
        with pytest.raises(RuntimeError, match="stub failure requested"):
            sched.run(jobs)
    with pytest.raises(ValueError):
        RunScheduler(gpus=0)


def test_scheduler_survives_a_failed_run_and_a_hung_job():
    """After a failed run() the same scheduler must give correct results again (no stale 'done' message of the failed run may
    land in the next run's result slots); a job that exceeds job_timeout has its worker replaced by a fresh process."""
    from mr_gan_amd.scheduler import RunScheduler
    rs = np.random.RandomState(3)
    X, y = rs.randn(120, 5), rs.randint(0, 6, size=120)
    with RunScheduler(runner=stub_training, devices=['cpu', 'cpu']) as sched:
        key = sched.put_dataset(X, y)
        jobs = _jobs(6, key)
        want = [stub_training(j, {key: (X, y)}, 'cpu')[0] for j in jobs]
        bad = [dict(j) for j in jobs]
        bad[0]['explode'] = True
        bad[1]['sleep'] = 1.0                                   # still in flight when job 0 fails
        with pytest.raises(RuntimeError, match="stub failure requested"):
            sched.run(bad)
        out = sched.run(jobs)                                   # same workers, same dataset
        np.testing.assert_allclose([o[0] for o in out], want, rtol=0, atol=0)
        hung = [dict(j) for j in jobs[:3]]
        hung[1]['sleep'] = 60.0
        with pytest.raises(RuntimeError, match="exceeded"):
            sched.run(hung, job_timeout=2.0)
        out = sched.run(jobs)                                   # the replaced worker got the dataset again
        np.testing.assert_allclose([o[0] for o in out], want, rtol=0, atol=0)
        # a second data set after the replacement: the fresh worker's acknowledgement of the RE-SENT first data set must not be
        # counted for the new key (each worker has its own result pipe, acknowledgements carry the key)
        X2, y2 = rs.randn(120, 5), rs.randint(0, 6, size=120)
        key2 = sched.put_dataset(X2, y2)
        jobs2 = _jobs(5, key2)
        out = sched.run(jobs2)
        np.testing.assert_allclose([o[0] for o in out], [stub_training(j, {key2: (X2, y2)}, 'cpu')[0] for j in jobs2], rtol=0, atol=0)
        # a worker process that dies (here: killed from outside) is reported, not waited for
        died = [dict(j) for j in jobs[:2]]
        died[0]['sleep'] = 30.0
        import threading
        killer = threading.Timer(1.0, lambda: [p.kill() for p in sched.procs])
        killer.start()
        with pytest.raises(RuntimeError, match="died"):
            sched.run(died)
        killer.cancel()


def test_scheduled_table1_prints_the_reference_lines(capsys):
    """--gpus N routes table 1 through the scheduler and still prints the reference's lines in the reference's order."""
    import importlib
    M = importlib.import_module('mr_gan_amd.mr_gan')     # the package re-exports the function under the same name

    class FakeScheduler(object):
        def __init__(self, gpus, jobs_per_gpu):
            self.gpus, self.jobs_per_gpu, self.batches = gpus, jobs_per_gpu, []

        def put_dataset(self, X, y):
            return 0

        def run(self, jobs):
            self.batches.append(len(jobs))
            self.jobs = getattr(self, 'jobs', []) + list(jobs)
            return [0.25 + 0.001 * j['percentlabeled'] for j in jobs]

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    made = []

    def factory(gpus, jobs_per_gpu):
        made.append(FakeScheduler(gpus, jobs_per_gpu))
        return made[-1]

    rs = np.random.RandomState(0)

    def fake_dataset(modalities=0, **kw):
        return rs.randn(60, 4), np.arange(60) % 6

    M.main(['--tables', '1', '--gpus', '2', '--jobs-per-gpu', '3'], dataset_fn=fake_dataset, scheduler_factory=factory)
    out = capsys.readouterr().out
    assert made[0].gpus == 2 and made[0].jobs_per_gpu == 3
    assert made[0].batches == [42] * 7                          # 7 modalities x (7 label fractions x 6 folds)
    assert out.count('Test error:') == 7 * 42
    assert out.count('Average error:') == 7 * 7
    i50 = out.index('Percentage of training data labeled: 50%')
    assert 'Test error: 0.3' in out[i50:i50 + 200]              # 0.25 + 0.001 * 50
    assert out.index('Force modality') < out.index('Temperature modality') < out.index('Force and Contact Mic modality')


def test_scheduled_tables_3_5_6_dispatch_every_training(capsys):
    """tables 3 (leave-one-object-out), 5 (contact time) and 6 (amount of unlabeled data) through the scheduler: the job counts
    of mr_gan.py:267-282, :289-318, :324-341 and the reference's lines in the reference's order."""
    import importlib
    M = importlib.import_module('mr_gan_amd.mr_gan')

    class FakeScheduler(object):
        def __init__(self, gpus, jobs_per_gpu):
            self.batches, self.jobs = [], []

        def put_dataset(self, X, y):
            return 0

        def run(self, jobs):
            self.batches.append(len(jobs))
            self.jobs += list(jobs)
            return [0.125 for _ in jobs]

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    made = []

    def factory(gpus, jobs_per_gpu):
        made.append(FakeScheduler(gpus, jobs_per_gpu))
        return made[-1]

    rs = np.random.RandomState(0)

    def fake_dataset(modalities=0, leaveObjectOut=False, **kw):
        if leaveObjectOut:
            return {'m%d_o%d' % (m, o): {'x': rs.randn(5, 4).tolist(), 'y': [m] * 5} for m in range(6) for o in range(3)}
        return rs.randn(60, 4), np.arange(60) % 6

    M.main(['--tables', '3', '5', '6', '--gpus', '1'], dataset_fn=fake_dataset, scheduler_factory=factory)
    out = capsys.readouterr().out
    f = made[0]
    # table 3: 2 modalities x 5 label fractions x 18 objects ; table 5: (3 x 7 + 7) x 6 folds ; table 6: 2 x 7 x 6 folds
    assert sum(f.batches) == 2 * 5 * 18 + 28 * 6 + 2 * 7 * 6
    assert out.count('Average leave-one-object-out error:') == 10
    assert out.count('Average error:') == 28 + 14
    assert 'm0_o0 Test error: 0.125 Test accuracy: 0.875' in out
    t6 = [j for j in f.jobs if j.get('percentunlabeled') is not None]
    assert sorted({j['percentunlabeled'] for j in t6}) == [0, 4, 8, 16, 32, 64, 96] and all(j['percentlabeled'] == 4 for j in t6)
    # leave-one-object-out jobs are index vectors into ONE shipped matrix (no per-job copy of the training rows): the held-out
    # object's 5 rows against the 85 others, every object held out once per label fraction
    loo = f.jobs[:180]
    assert all('trainTestSets' not in j for j in f.jobs)
    assert len(loo[0]['test_idx']) == 5 and len(loo[0]['train_idx']) == 85 and not set(loo[0]['test_idx']) & set(loo[0]['train_idx'])
    assert sorted(int(j['test_idx'][0]) for j in loo[:18]) == list(range(0, 90, 5))
