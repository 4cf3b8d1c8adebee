"""world_size-2 gloo test of the data-parallel protocol (mr_gan_amd/dist.py) on CPU.

The HIP engine cannot run here, so each rank drives dist.DataParallel through a stand-in PhaseBackend whose
phases are the CPU oracle cut at the same points as include/mrgan_abi.h (MRGAN_D_* / MRGAN_G_* phases) and whose
exchange regions are CPU tensors.  Two ranks at batch B/2 must reproduce the single-process batch-B oracle step:
a missing, mis-ordered or mis-scaled exchange changes the result.  (The GPU-side equivalence of the engine's own
phases with the same protocol is tests/test_gpu_parity.py::test_two_rank_emulation_equals_full_batch.)
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import mrgan_oracle as O  # noqa: E402
from tests.helpers import Case  # noqa: E402


def _flat(ts):
    return np.concatenate([np.asarray(t, np.float64).ravel() for t in ts])


def _unflat(v, like):
    out, o = [], 0
    for t in like:
        out.append(v[o:o + t.size].reshape(t.shape))
        o += t.size
    return out


class OraclePhases(object):
    """The oracle's D / G sub-steps cut into the phases of the C ABI, for one rank's row shard."""

    def __init__(self, g, d, world, exact=True):
        from mr_gan_amd import engine as E
        self.E = E
        self.orc = O.MRGANOracle(g, d)
        self.world = world
        self.exact = exact          # False: statistics stay per shard (count = local rows, FM gradient scaled by 1/world)
        n1 = g[1].size
        F = d[8].shape[1]
        self.regions = {
            E.REGION_BN_STATS: torch.zeros(2 * n1, dtype=torch.float64),
            E.REGION_FM_MOMENTS: torch.zeros(2 * F, dtype=torch.float64),
            E.REGION_BN_BWD: torch.zeros(2 * n1, dtype=torch.float64),
            E.REGION_GRAD_D: torch.zeros(sum(p.size for p in d) + 4, dtype=torch.float64),
            E.REGION_GRAD_G: torch.zeros(sum(p.size for p in g) + 4, dtype=torch.float64),
        }
        self.bf16 = False           # FLAG_GRAD_BF16: the reduce phases write bfloat16 regions (+ fp32 tails), the Adam phases read them
        self.last = None

    def region(self, which):
        return self.regions[which]

    def _publish(self, which):
        """what the engine's reduce phase does with MRGAN_FLAG_GRAD_BF16: gradients rounded once to bfloat16, the four scalars fp32"""
        E = self.E
        if not self.bf16:
            return
        d = which == E.REGION_GRAD_D
        v = self.regions[which]
        self.regions[E.REGION_GRAD_D_BF16 if d else E.REGION_GRAD_G_BF16] = v[:-4].to(torch.bfloat16)
        self.regions[E.REGION_TAIL_D if d else E.REGION_TAIL_G] = v[-4:].float()

    def _collect(self, which):
        E = self.E
        if not self.bf16:
            return self.regions[which].numpy()
        d = which == E.REGION_GRAD_D
        body = self.regions[E.REGION_GRAD_D_BF16 if d else E.REGION_GRAD_G_BF16].double().numpy()
        return np.concatenate([body, self.regions[E.REGION_TAIL_D if d else E.REGION_TAIL_G].double().numpy()])

    # ---- generator pieces with externally supplied (global) batch statistics ----
    def _gen_head(self, z):
        W1, b1 = self.orc.g[0], self.orc.g[1]
        self.z = z
        self.h1 = O.softplus(z @ W1 + b1)
        r = self.regions[self.E.REGION_BN_STATS]
        r[:] = torch.from_numpy(np.concatenate([self.h1.sum(0), (self.h1 ** 2).sum(0)]))

    def _count(self, B):
        return B * self.world if self.exact else B

    def _gen_tail(self, Bg):
        _, _, gamma, beta, W2, b2, W3, b3 = self.orc.g
        st = self.regions[self.E.REGION_BN_STATS].numpy()
        n1 = st.size // 2
        self.mu = st[:n1] / Bg
        var = st[n1:] / Bg - self.mu ** 2
        self.rstd = 1.0 / np.sqrt(var + O.BN_EPS)
        self.xhat = (self.h1 - self.mu) * self.rstd
        self.hbn = gamma * self.xhat + beta
        self.pre2 = self.hbn @ W2 + b2
        self.h2 = O.softplus(self.pre2)
        return self.h2 @ W3 + b3

    def disc_phase(self, a, phase):
        E, orc = self.E, self.orc
        Bg = a['x_lab'].shape[0] * self.world
        if phase == E.D_GEN:
            self._gen_head(a['z'])
        elif phase == E.D_MAIN:
            x_fake = self._gen_tail(self._count(a['x_lab'].shape[0]))
            l_lab, _, c_lab = O.disc_forward(orc.d, a['x_lab'], a['n_lab'])
            l_unl, _, c_unl = O.disc_forward(orc.d, a['x_unl'], a['n_unl'])
            l_fake, _, c_fake = O.disc_forward(orc.d, x_fake, a['n_fake'])
            # local sums scaled by the GLOBAL batch: the all-reduce (sum) completes the mean
            ll, lu, err = O.disc_losses(l_lab, a['labels'], l_unl, l_fake)
            dl = O.disc_loss_grads(l_lab, a['labels'], l_unl, l_fake)
            grads = None
            for c, d_ in zip((c_lab, c_unl, c_fake), dl):
                g_, _ = O.disc_backward(orc.d, c, dlogits=d_ / self.world)
                grads = g_ if grads is None else [x + y for x, y in zip(grads, g_)]
            tail = np.array([ll, lu, err, 0.0]) / self.world
            self.regions[E.REGION_GRAD_D][:] = torch.from_numpy(np.concatenate([_flat(grads), tail]))
            self._publish(E.REGION_GRAD_D)
        elif phase == E.D_ADAM:
            v = self._collect(E.REGION_GRAD_D)
            orc.adam.apply(orc.d, _unflat(v[:-4], orc.d), 'd')
            self.last = tuple(v[-4:-1])

    def gen_phase(self, a, phase):
        E, orc = self.E, self.orc
        B = a['x_unl'].shape[0]
        Bg = self._count(B)
        gscale = 1.0 if self.exact else 1.0 / self.world
        if phase == E.G_GEN:
            self._gen_head(a['z'])
        elif phase == E.G_FEAT:
            x_fake = self._gen_tail(Bg)
            _, self.f_fake, self.c_fake = O.disc_forward(orc.d, x_fake, a['n_fake'])
            _, f_real, _ = O.disc_forward(orc.d, a['x_unl'], a['n_real'])
            self.regions[E.REGION_FM_MOMENTS][:] = torch.from_numpy(np.concatenate([self.f_fake.sum(0), f_real.sum(0)]))
        elif phase == E.G_BWD:
            m = self.regions[E.REGION_FM_MOMENTS].numpy()
            F = m.size // 2
            diff = (m[:F] - m[F:]) / Bg
            self.loss = float(np.mean(diff ** 2))
            df = np.broadcast_to(gscale * 2.0 / (F * Bg) * diff, self.f_fake.shape)
            _, dx = O.disc_backward(orc.d, self.c_fake, dfeat=df, want_param_grads=False)
            W2, W3 = orc.g[4], orc.g[6]
            self.dW3, self.db3 = self.h2.T @ dx, dx.sum(0)
            dpre2 = (dx @ W3.T) * O.sigmoid(self.pre2)
            self.dW2, self.db2 = self.hbn.T @ dpre2, dpre2.sum(0)
            self.dhbn = dpre2 @ W2.T
            self.regions[E.REGION_BN_BWD][:] = torch.from_numpy(
                np.concatenate([self.dhbn.sum(0), (self.dhbn * self.xhat).sum(0)]))
        elif phase == E.G_TAIL:
            s = self.regions[E.REGION_BN_BWD].numpy()
            n1 = s.size // 2
            dbeta, dgamma = s[:n1], s[n1:]
            gamma = orc.g[2]
            dh1 = (gamma * self.rstd / Bg) * (Bg * self.dhbn - dbeta - self.xhat * dgamma)
            dpre1 = dh1 * (1.0 - np.exp(-self.h1))
            # gamma / beta gradients enter the flat buffer as LOCAL sums (the all-reduce makes them global)
            loc_dgamma, loc_dbeta = (self.dhbn * self.xhat).sum(0), self.dhbn.sum(0)
            grads = [self.z.T @ dpre1, dpre1.sum(0), loc_dgamma, loc_dbeta, self.dW2, self.db2, self.dW3, self.db3]
            self.regions[E.REGION_GRAD_G][:] = torch.from_numpy(np.concatenate([_flat(grads), np.zeros(4)]))
            self._publish(E.REGION_GRAD_G)
        elif phase == E.G_ADAM:
            v = self._collect(E.REGION_GRAD_G)
            orc.adam.apply(orc.g, _unflat(v[:-4], orc.g), 'g')
            self.last = self.loss


def _case(sorted_labels=False):
    case = Case(D=12, B=16, steps=2)
    if sorted_labels:
        # uneven label tiling: the tail of the reference's labeled stream only touches the first classes of the class-sorted
        # pool (mr_gan.py:189), so the ranks of a sharded batch can see disjoint label sets
        case.labels = np.sort(case.labels, axis=1)
    return case


def _worker(rank, world, port, exact, q, sorted_labels=False, grad_dtype=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mr_gan_amd.dist import DataParallel
    B, D, steps = 16, 12, 2
    case = _case(sorted_labels)
    h = B // world
    backend = OraclePhases(case.g0, case.d0, world, exact)
    backend.bf16 = grad_dtype == 'bf16'
    dp = DataParallel(backend, exact=exact, grad_dtype=grad_dtype)
    res = []
    it = 0
    for t in range(steps):
        dp.disc_step(case.disc_inputs(t, it, rows=h, row0=rank * h))
        res.append(backend.last)
        it += 1
        dp.gen_step(case.gen_inputs(t, it, rows=h, row0=rank * h))
        res.append(backend.last)
        it += 1
    q.put((rank, res, [p.copy() for p in backend.orc.d], [p.copy() for p in backend.orc.g]))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(exact, world=2, sorted_labels=False, grad_dtype=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, exact, q, sorted_labels, grad_dtype)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=120) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return out


def test_two_ranks_reproduce_full_batch_step():
    out = _run(exact=True)
    ref = _case().run_oracle()
    (_, res0, d0, g0), (_, res1, d1, g1) = out
    for t in range(2):
        np.testing.assert_allclose(res0[2 * t], ref['disc'][t], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(res0[2 * t + 1], ref['gen'][t], rtol=1e-8, atol=1e-14)
        np.testing.assert_allclose(res1[2 * t], res0[2 * t], rtol=0, atol=0)      # every rank sees the global scalars
    for a, b, r in zip(d0, d1, ref['d']):
        np.testing.assert_array_equal(a, b)                                       # replicas stay identical
        np.testing.assert_allclose(a, r, rtol=1e-7, atol=1e-10)
    for a, b, r in zip(g0, g1, ref['g']):
        np.testing.assert_array_equal(a, b)
        np.testing.assert_allclose(a, r, rtol=1e-6, atol=1e-10)


def test_four_ranks_with_disjoint_label_sets_reproduce_full_batch_step():
    """world 4, four rows per rank, labels sorted so that every rank holds different classes: the labeled loss is a mean over
    the GLOBAL batch, so only the protocol's 1/B_global scaling and the gradient all-reduce make this equal the one-process step"""
    out = _run(exact=True, world=4, sorted_labels=True)
    ref = _case(True).run_oracle()
    for t in range(2):
        np.testing.assert_allclose(out[0][1][2 * t], ref['disc'][t], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(out[0][1][2 * t + 1], ref['gen'][t], rtol=1e-8, atol=1e-14)
    for r in range(1, 4):
        for a, b in zip(out[0][2] + out[0][3], out[r][2] + out[r][3]):
            np.testing.assert_array_equal(a, b)
    for a, r in zip(out[0][2], ref['d']):
        np.testing.assert_allclose(a, r, rtol=1e-7, atol=1e-10)
    for a, r in zip(out[0][3], ref['g']):
        np.testing.assert_allclose(a, r, rtol=1e-6, atol=1e-10)


def test_bf16_gradient_payload_keeps_replicas_identical():
    """grad_dtype='bf16': half the bytes per gradient all-reduce; replicas still agree bit for bit (they all receive the same
    reduced bf16 values) and the losses stay those of the full-batch step; the weights differ from the fp32 exchange at the
    bf16 level (a labelled, different numerical path)"""
    out = _run(exact=True, world=2, grad_dtype='bf16')
    ref = _case().run_oracle()
    (_, res0, d0, g0), (_, res1, d1, g1) = out
    np.testing.assert_allclose(res0[0], ref['disc'][0], rtol=1e-5, atol=1e-7)        # first sub-step: weights still identical
    for a, b in zip(d0 + g0, d1 + g1):
        np.testing.assert_array_equal(a, b)
    from tests.helpers import update_rel_err
    case = _case()
    errs = [update_rel_err(a, r, w0) for a, r, w0 in zip(d0, ref['d'], case.d0)]
    assert max(errs) < 0.5 and max(errs) > 1e-6


def test_local_statistics_mode_is_a_different_algorithm():
    """exact=False skips the statistic exchanges: replicas still agree (gradients are all-reduced) but the step is
    not the full-batch step -- which is why bench.py labels it local_stats."""
    out = _run(exact=False)
    case = Case(D=12, B=16, steps=2)
    ref = case.run_oracle()
    (_, _, d0, g0), (_, _, d1, g1) = out
    for a, b in zip(g0, g1):
        np.testing.assert_array_equal(a, b)
    assert max(np.max(np.abs(a - r)) for a, r in zip(g0, ref['g'])) > 1e-6


def test_dp_flags():
    from mr_gan_amd import engine as E
    from mr_gan_amd.dist import dp_flags
    assert dp_flags(True) == E.FLAG_FLAT_GRADS | E.FLAG_SYNC_STATS
    assert dp_flags(False) == E.FLAG_FLAT_GRADS


class _PairingStub(object):
    """Records the phase / exchange sequence dist.DataParallel issues, with the engine's pairing and fp8-calibration state
    machine restated from csrc/engine.hip: a pair hint is good for ONE D sub-step (D_MAIN hands it over as `gen_ready` and clears
    it; any later D_MAIN without a hint resets gen_ready); G_GEN re-runs the generator head -- and so rewrites the BatchNorm
    sums of its segment -- unless gen_ready is set.  `stale` counts generator tails that normalised with sums no exchange had
    completed."""

    def __init__(self, fp8):
        from mr_gan_amd import engine as E
        self.E = E
        self.cal = [not fp8, not fp8]
        self.pair_gen = self.gen_ready = 0
        self.reduced = {0: True, 1: True}          # BatchNorm sums of generator segment s have been all-reduced
        self.stale = 0
        self.log = []

    def pair_hint(self, on=True):
        self.pair_gen = 1 if on else 0

    def fp8_calibration(self, kind, action):
        E = self.E.Engine
        if action == E.FP8_CAL_QUERY:
            return 1 if self.cal[kind] else 0
        if action == E.FP8_CAL_DONE:
            self.cal[kind] = True
        return 0

    def exchanged(self, which):
        self.log.append(('allreduce', which))
        if which == self.E.REGION_BN_STATS:
            self.reduced = {0: True, 1: True}

    def disc_phase(self, args, phase):
        E = self.E
        self.log.append(('D', phase))
        if phase == E.D_GEN:
            self.reduced[0] = False
            if self.pair_gen:
                self.reduced[1] = False
        elif phase == E.D_MAIN:
            self.stale += 0 if self.reduced[0] and (not self.pair_gen or self.reduced[1]) else 1
            self.gen_ready, self.pair_gen = self.pair_gen, 0

    def gen_phase(self, args, phase):
        E = self.E
        self.log.append(('G', phase))
        if phase == E.G_GEN:
            self.view = 1 if self.gen_ready else 0
            if not self.gen_ready:
                self.reduced[0] = False
        elif phase == E.G_FEAT:
            self.stale += 0 if self.reduced[self.view] else 1
            self.gen_ready = 0


@pytest.mark.parametrize("fp8", [False, True])
def test_train_pair_never_normalises_with_unreduced_statistics(fp8):
    """exact data parallelism + train_pair (+ fp8 calibration): every generator tail runs behind a completed BatchNorm-sum
    exchange.  Regression test for the first fp8 pair, where the D sub-step's first dry pass used to consume the pair hint
    and the G sub-step's first dry pass then skipped its exchange although it had just rewritten the sums."""
    from mr_gan_amd import engine as E
    from mr_gan_amd.dist import DataParallel

    class Recording(DataParallel):
        def _allreduce(self, which):
            self.backend.exchanged(which)

    stub = _PairingStub(fp8)
    dp = Recording(stub, exact=True)
    for _ in range(3):
        dp.train_pair(object(), object())
    assert stub.stale == 0, stub.log
    # steady state: one BatchNorm exchange per pair (both segments travel in the D sub-step's)
    tail = stub.log[-13:]
    assert sum(1 for ev in tail if ev == ('allreduce', E.REGION_BN_STATS)) == 1, tail
    assert stub.cal == [True, True]
