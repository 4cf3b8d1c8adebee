"""Run-level scheduler for the table harness.

Every mr_gan() call of the reference's tables (fold x percent x modality: 294 trainings for table 1, 720 for
table 3; mr_gan.py:248-260, :267-282, :324-341) is independent of the others, so the harness can hand them to
one worker process per GPU -- and to several workers per GPU: at the reference's batch of 50 a training is a
chain of launch-latency-bound kernels that leaves most of an MI355X idle.  No communication between workers.

    with RunScheduler(gpus=8, jobs_per_gpu=2) as sched:
        key = sched.put_dataset(X, y)                    # shipped to every worker once
        errors = sched.run([dict(dataset=key, train_idx=tr, test_idx=te, percentlabeled=p, epochs=100) ...])

run() returns the results in job order, so the harness prints exactly the reference's lines.  Workers are spawned
(never forked), so a script that creates a scheduler needs the usual `if __name__ == "__main__":` guard.  The default runner
trains with mr_gan_amd.mr_gan.mr_gan on the worker's device; tests substitute a CPU stand-in (`runner=`).
"""
import multiprocessing as mp
import traceback


def train_job(job, datasets, device):
    """Default runner: one mr_gan() training on `device`.  job: dataset key + row indices (or explicit
    trainTestSets) + the keyword arguments of mr_gan()."""
    from mr_gan_amd.mr_gan import mr_gan
    kw = {k: v for k, v in job.items() if k not in ('dataset', 'train_idx', 'test_idx', 'trainTestSets')}
    if 'trainTestSets' in job:
        sets = job['trainTestSets']
    else:
        X, y = datasets[job['dataset']]
        tr, te = job['train_idx'], job['test_idx']
        sets = [X[tr], X[te], y[tr], y[te]]
    return mr_gan(None, None, trainTestSets=sets, device=device, **kw)


def _worker(wid, device, runner, inbox, outbox):
    datasets = {}
    local = device
    while True:
        msg = inbox.get()
        if msg is None:
            return
        kind = msg[0]
        if kind == 'dataset':
            datasets[msg[1]] = (msg[2], msg[3])
            outbox.put(('ack', wid, msg[1]))
        else:
            _, idx, job = msg
            try:
                outbox.put(('done', wid, idx, runner(job, datasets, local)))
            except Exception:                                                    # report, keep serving
                outbox.put(('fail', wid, idx, traceback.format_exc()))


class RunScheduler(object):
    def __init__(self, gpus=1, jobs_per_gpu=1, runner=train_job, devices=None):
        if gpus < 1 or jobs_per_gpu < 1:
            raise ValueError("gpus and jobs_per_gpu must be positive")
        self.devices = list(devices) if devices is not None else ['cuda:%d' % g for g in range(gpus) for _ in range(jobs_per_gpu)]
        ctx = mp.get_context('spawn')                       # never fork a process that may have initialised HIP
        self.outbox = ctx.Queue()
        self.inboxes, self.procs = [], []
        for wid, dev in enumerate(self.devices):
            q = ctx.Queue()
            q.cancel_join_thread()                          # never block interpreter exit on bytes a dead worker will not read
            p = ctx.Process(target=_worker, args=(wid, dev, runner, q, self.outbox), daemon=True)
            p.start()
            self.inboxes.append(q)
            self.procs.append(p)
        self._nkeys = 0
        self.assignments = []                               # (job index, worker id) of the last run()

    def put_dataset(self, X, y):
        key = self._nkeys
        self._nkeys += 1
        for q in self.inboxes:
            q.put(('dataset', key, X, y))
        for _ in self.inboxes:
            self._get()
        return key

    def _get(self):
        while True:
            try:
                return self.outbox.get(timeout=5.0)
            except Exception:                               # queue.Empty
                dead = [i for i, p in enumerate(self.procs) if p.exitcode not in (None, 0)]
                if dead:
                    raise RuntimeError("scheduler worker(s) %s died" % dead)

    def run(self, jobs):
        """Greedy dispatch: every idle worker takes the next job; results come back in job order."""
        jobs = list(jobs)
        results = [None] * len(jobs)
        self.assignments = []
        nxt, inflight = 0, 0
        idle = list(range(len(self.inboxes)))
        while nxt < len(jobs) or inflight:
            while idle and nxt < len(jobs):
                w = idle.pop(0)
                self.inboxes[w].put(('job', nxt, jobs[nxt]))
                self.assignments.append((nxt, w))
                nxt += 1
                inflight += 1
            msg = self._get()
            if msg[0] == 'fail':
                raise RuntimeError("job %d failed on worker %d (%s):\n%s" % (msg[2], msg[1], self.devices[msg[1]], msg[3]))
            if msg[0] == 'done':
                results[msg[2]] = msg[3]
                idle.append(msg[1])
                inflight -= 1
        return results

    def close(self):
        for q, p in zip(self.inboxes, self.procs):
            if p.is_alive():
                q.put(None)
        for p in self.procs:
            p.join(timeout=30)
            if p.is_alive():
                p.terminate()
        self.inboxes, self.procs = [], []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
