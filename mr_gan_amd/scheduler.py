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
            _, run_id, idx, job = msg
            try:
                outbox.put(('done', wid, run_id, idx, runner(job, datasets, local)))
            except Exception:                                                    # report, keep serving
                outbox.put(('fail', wid, run_id, idx, traceback.format_exc()))


class RunScheduler(object):
    def __init__(self, gpus=1, jobs_per_gpu=1, runner=train_job, devices=None):
        if gpus < 1 or jobs_per_gpu < 1:
            raise ValueError("gpus and jobs_per_gpu must be positive")
        self.devices = list(devices) if devices is not None else ['cuda:%d' % g for g in range(gpus) for _ in range(jobs_per_gpu)]
        ctx = mp.get_context('spawn')                       # never fork a process that may have initialised HIP
        self._ctx, self._runner, self._datasets = ctx, runner, {}
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
        self._datasets[key] = (X, y)
        for q in self.inboxes:
            q.put(('dataset', key, X, y))
        acks = 0
        while acks < len(self.inboxes):
            if self._get()[0] == 'ack':                     # (late results of a failed run() may still be in the queue)
                acks += 1
        return key

    def _get(self, timeout=None):
        import time
        t0 = time.time()
        while True:
            try:
                msg = self.outbox.get(timeout=1.0 if timeout else 5.0)
                if msg[0] == 'ack' and len(msg) == 3:
                    return msg
                return msg
            except Exception:                               # queue.Empty
                dead = [i for i, p in enumerate(self.procs) if p.exitcode not in (None, 0)]
                if dead:
                    raise RuntimeError("scheduler worker(s) %s died" % dead)
                if timeout and time.time() - t0 >= timeout:
                    raise TimeoutError()

    def run(self, jobs, job_timeout=None):
        """Greedy dispatch: every idle worker takes the next job; results come back in job order.

        Every message carries the id of the run() that issued it, so results of an earlier, failed run that arrive late are
        ignored.  On a failed job the jobs still in flight are waited for (their workers stay usable) before the error is
        raised.  job_timeout (seconds, optional): a job that takes longer has its worker terminated and replaced by a fresh
        process, and counts as failed."""
        import time
        jobs = list(jobs)
        results = [None] * len(jobs)
        self.assignments = []
        self._run_id = getattr(self, '_run_id', 0) + 1
        rid = self._run_id
        nxt, failure = 0, None
        idle = list(range(len(self.inboxes)))
        started = {}                                        # worker -> (job index, start time)
        while (nxt < len(jobs) and failure is None) or started:
            while idle and nxt < len(jobs) and failure is None:
                w = idle.pop(0)
                self.inboxes[w].put(('job', rid, nxt, jobs[nxt]))
                self.assignments.append((nxt, w))
                started[w] = (nxt, time.time())
                nxt += 1
            try:
                msg = self._get(timeout=1.0 if job_timeout else None)
            except TimeoutError:
                msg = None
            if msg is not None and msg[0] in ('done', 'fail') and msg[2] == rid:
                w = msg[1]
                started.pop(w, None)
                idle.append(w)
                if msg[0] == 'done':
                    results[msg[3]] = msg[4]
                elif failure is None:
                    failure = "job %d failed on worker %d (%s):\n%s" % (msg[3], w, self.devices[w], msg[4])
            if job_timeout:
                for w, (idx, t0) in list(started.items()):
                    if time.time() - t0 > job_timeout:
                        self._respawn(w)
                        started.pop(w)
                        idle.append(w)
                        if failure is None:
                            failure = "job %d exceeded %.0f s on worker %d (%s); the worker was replaced" % (idx, job_timeout, w, self.devices[w])
        if failure is not None:
            raise RuntimeError(failure)
        return results

    def _respawn(self, w):
        """terminate worker w and start a fresh process in its place (it loses the datasets: they are re-sent)"""
        p = self.procs[w]
        if p.is_alive():
            p.terminate()
        p.join(timeout=10)
        q = self._ctx.Queue()
        q.cancel_join_thread()
        np_ = self._ctx.Process(target=_worker, args=(w, self.devices[w], self._runner, q, self.outbox), daemon=True)
        np_.start()
        self.inboxes[w], self.procs[w] = q, np_
        for key, (X, y) in self._datasets.items():
            q.put(('dataset', key, X, y))

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
