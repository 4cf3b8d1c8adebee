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


def _worker(wid, device, runner, inbox, out):
    """inbox: this worker's job queue; out: the sending end of this worker's OWN result pipe (no queue is shared between
    workers: terminating one of them can neither corrupt nor lock what the others write to)"""
    datasets = {}
    while True:
        msg = inbox.get()
        if msg is None:
            return
        kind = msg[0]
        if kind == 'dataset':
            datasets[msg[1]] = (msg[2], msg[3])
            out.send(('ack', msg[1]))
        else:
            _, run_id, idx, job = msg
            try:
                out.send(('done', run_id, idx, runner(job, datasets, device)))
            except Exception:                                                    # report, keep serving
                out.send(('fail', run_id, idx, traceback.format_exc()))


class RunScheduler(object):
    def __init__(self, gpus=1, jobs_per_gpu=1, runner=train_job, devices=None):
        if gpus < 1 or jobs_per_gpu < 1:
            raise ValueError("gpus and jobs_per_gpu must be positive")
        self.devices = list(devices) if devices is not None else ['cuda:%d' % g for g in range(gpus) for _ in range(jobs_per_gpu)]
        self._ctx = mp.get_context('spawn')                 # never fork a process that may have initialised HIP
        self._runner, self._datasets = runner, {}
        n = len(self.devices)
        self.inboxes, self.procs, self.results = [None] * n, [None] * n, [None] * n
        for wid in range(n):
            self._spawn(wid)
        self._nkeys = 0
        self._run_id = 0
        self.assignments = []                               # (job index, worker id) of the last run()

    def _spawn(self, wid):
        q = self._ctx.Queue()
        q.cancel_join_thread()                              # never block interpreter exit on bytes a dead worker will not read
        recv, send = self._ctx.Pipe(duplex=False)
        p = self._ctx.Process(target=_worker, args=(wid, self.devices[wid], self._runner, q, send), daemon=True)
        p.start()
        send.close()                                        # the parent keeps only the receiving end: EOF = the worker is gone
        self.inboxes[wid], self.procs[wid], self.results[wid] = q, p, recv

    def _poll(self, timeout):
        """messages that arrived within `timeout` seconds, as (worker id, message); raises if a worker died"""
        from multiprocessing.connection import wait
        out = []
        for conn in wait(self.results, timeout):
            wid = self.results.index(conn)
            try:
                out.append((wid, conn.recv()))
            except (EOFError, OSError):
                raise RuntimeError("scheduler worker %d (%s) died (exit code %s)" % (wid, self.devices[wid], self.procs[wid].exitcode))
        return out

    def put_dataset(self, X, y):
        key = self._nkeys
        self._nkeys += 1
        self._datasets[key] = (X, y)
        for q in self.inboxes:
            q.put(('dataset', key, X, y))
        pending = set(range(len(self.inboxes)))
        while pending:
            for wid, msg in self._poll(5.0):
                if msg[0] == 'ack' and msg[1] == key:       # (acks of datasets re-sent to a replaced worker carry older keys)
                    pending.discard(wid)
        return key

    def run(self, jobs, job_timeout=None):
        """Greedy dispatch: every idle worker takes the next job; results come back in job order.

        Every message carries the id of the run() that issued it, so results of an earlier, failed run that arrive late are
        ignored.  On a failed job the jobs still in flight are waited for (their workers stay usable) before the error is
        raised.  job_timeout (seconds, optional): a job that takes longer has its worker terminated and replaced by a fresh
        process with a fresh job queue and result pipe (nothing the old process wrote can arrive any more), and counts as
        failed."""
        import time
        jobs = list(jobs)
        results = [None] * len(jobs)
        self.assignments = []
        self._run_id += 1
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
            for w, msg in self._poll(1.0 if job_timeout else 5.0):
                if msg[0] not in ('done', 'fail') or msg[1] != rid or w not in started:
                    continue
                started.pop(w)
                idle.append(w)
                if msg[0] == 'done':
                    results[msg[2]] = msg[3]
                elif failure is None:
                    failure = "job %d failed on worker %d (%s):\n%s" % (msg[2], w, self.devices[w], msg[3])
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
        self.results[w].close()
        self._spawn(w)
        for key, (X, y) in self._datasets.items():
            self.inboxes[w].put(('dataset', key, X, y))

    def close(self):
        for q, p in zip(self.inboxes, self.procs):
            if p.is_alive():
                q.put(None)
        for p in self.procs:
            p.join(timeout=30)
            if p.is_alive():
                p.terminate()
        for c in self.results:
            c.close()
        self.inboxes, self.procs, self.results = [], [], []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
