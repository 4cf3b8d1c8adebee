"""Throughput of the run-level scheduler on ONE GPU at the reference's own sizes (batch 50, N = 7200, D = 1200:
force + temperature), as trainings per minute for 1 / 2 / 4 concurrent workers.
usage: python scripts/sched_bench.py [epochs] [jobs]"""
import sys
import time

sys.path.insert(0, '.')
import numpy as np

from mr_gan_amd import synthetic_blobs
from mr_gan_amd.scheduler import RunScheduler

def main():
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    njobs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    X, y = synthetic_blobs(n=7200, d=1200, seed=1)
    rs = np.random.RandomState(0)
    jobs = []
    for i in range(njobs):
        perm = rs.permutation(7200)
        jobs.append(dict(train_idx=perm[:6000], test_idx=perm[6000:], percentlabeled=50, epochs=epochs, seed=i))
    for jpg in (1, 2, 4):
        with RunScheduler(gpus=1, jobs_per_gpu=jpg) as sched:
            key = sched.put_dataset(X, y)
            sched.run([dict(dataset=key, **j) for j in jobs[:jpg]])            # warm-up: library load, first launches
            t0 = time.time()
            out = sched.run([dict(dataset=key, **j) for j in jobs])
            dt = time.time() - t0
        print("jobs_per_gpu %d: %d trainings x %d epochs (120 steps each) in %.2f s -> %.1f trainings/min, mean test error %.3f"
              % (jpg, njobs, epochs, dt, 60.0 * njobs / dt, float(np.mean(out))))


if __name__ == "__main__":      # the workers are spawned: they re-import this file
    main()
