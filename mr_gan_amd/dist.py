"""Data-parallel driver: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" in the CPU tests), gradients and batch statistics exchanged between the phases of
include/mrgan_abi.h.

The reference is single-process; this is new.  Every rank holds a replica of the weights
and Adam state and processes batch/world rows of each of the labeled / unlabeled / generated streams.
Two quantities of the path are batch-global rather than per-sample sums -- the BatchNorm statistics of
the generator (mr_gan.py:112) and the feature-matching moments (mr_gan.py:152-153) -- so with
exact=True they are all-reduced too and W ranks at batch B/W reproduce the single-GPU batch-B step up
to summation order.  exact=False keeps them per-shard (a different, cheaper algorithm; labelled
"local_stats" wherever it is reported).

Exchanges per (D, G) pair, all sums of fp32 buffers that live inside the library workspace:
    D: [BN stats, both sub-steps' segments]  ->  flat D gradients (+4 scalars)
    G: [BN stats, only if not already exchanged]  [FM moments 2xFp]  [BN-backward sums 2xN1p]  ->  flat G gradients (+4 scalars)
"""
import torch.distributed as dist

from mr_gan_amd import engine as E


class PhaseBackend(object):
    """What DataParallel needs from an engine: run phases, expose the exchange regions as tensors.
    Engine satisfies it; the CPU tests supply an oracle-backed stand-in to exercise the protocol."""

    def disc_phase(self, args, phase):
        raise NotImplementedError

    def gen_phase(self, args, phase):
        raise NotImplementedError

    def region(self, which):
        raise NotImplementedError


class EngineBackend(PhaseBackend):
    def __init__(self, engine):
        self.engine = engine

    def disc_phase(self, args, phase):
        return self.engine.disc_step(args, phase, phase, want_outputs=False)

    def gen_phase(self, args, phase):
        return self.engine.gen_step(args, phase, phase, want_outputs=False)

    def region(self, which):
        return self.engine.region(which)

    def pair_hint(self, on=True):
        self.engine.pair_hint(on)

    def fp8_calibration(self, kind, action):
        return self.engine.fp8_calibration(kind, action)


class DataParallel(object):
    """grad_dtype: None (default) all-reduces the flat gradient buffers in fp32 -- replicas then reproduce the one-GPU step up
    to summation order; 'bf16' sends them as bfloat16 (half the bytes on the xGMI links: 2.5 MB instead of 5.1 MB for the
    discriminator at D = 512; the four scalars at the tail stay fp32) at the price of an 8-bit mantissa per addend -- a
    different, labelled numerical path.  It needs a backend built with dp_flags(grad_dtype='bf16'): the engine then writes and
    reads the bfloat16 regions itself.  The small statistic regions always travel in fp32."""

    def __init__(self, backend, exact=True, group=None, grad_dtype=None):
        self.backend = backend
        self.exact = exact
        self.group = group
        self.grad_dtype = grad_dtype
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def _allreduce(self, which):
        if self.world <= 1:
            return
        if self.grad_dtype == 'bf16' and which in (E.REGION_GRAD_D, E.REGION_GRAD_G):
            # the library wrote the gradients as bfloat16 (FLAG_GRAD_BF16) and reads them back from the same region: reduced in
            # place, no cast, no allocation; the four loss scalars travel on in fp32
            d = which == E.REGION_GRAD_D
            dist.all_reduce(self.backend.region(E.REGION_GRAD_D_BF16 if d else E.REGION_GRAD_G_BF16), op=dist.ReduceOp.SUM, group=self.group)
            dist.all_reduce(self.backend.region(E.REGION_TAIL_D if d else E.REGION_TAIL_G), op=dist.ReduceOp.SUM, group=self.group)
        else:
            dist.all_reduce(self.backend.region(which), op=dist.ReduceOp.SUM, group=self.group)

    def _calibrate(self, kind, run_pass):
        """fp8 engines: the first sub-step of a kind is preceded by dry passes (forward + backward phases WITH the statistic
        exchanges, no gradient exchange, no update) that settle the delayed scales (include/mrgan_abi.h)"""
        cal = getattr(self.backend, "fp8_calibration", None)
        if cal is None or cal(kind, E.Engine.FP8_CAL_QUERY):
            return
        cal(kind, E.Engine.FP8_CAL_BEGIN)
        for _ in range(E.Engine.FP8_DRY_PASSES):
            run_pass()
            cal(kind, E.Engine.FP8_CAL_END_PASS)
        cal(kind, E.Engine.FP8_CAL_DONE)

    def _disc_fwd_bwd(self, args):
        b = self.backend
        b.disc_phase(args, E.D_GEN)
        if self.exact:
            self._allreduce(E.REGION_BN_STATS)
        b.disc_phase(args, E.D_MAIN)

    def _gen_fwd_bwd(self, args, stats_done):
        b = self.backend
        b.gen_phase(args, E.G_GEN)
        if self.exact and not stats_done:
            self._allreduce(E.REGION_BN_STATS)
        b.gen_phase(args, E.G_FEAT)
        if self.exact:
            self._allreduce(E.REGION_FM_MOMENTS)
        b.gen_phase(args, E.G_BWD)

    def disc_step(self, args):
        b = self.backend
        self._calibrate(0, lambda: self._disc_fwd_bwd(args))
        b.disc_phase(args, E.D_GEN)
        if self.exact:
            self._allreduce(E.REGION_BN_STATS)
        b.disc_phase(args, E.D_MAIN)
        self._allreduce(E.REGION_GRAD_D)
        b.disc_phase(args, E.D_ADAM)

    def gen_step(self, args, stats_done=False):
        b = self.backend
        first = [True]

        def dry():
            # the first dry pass may still consume the paired generator forward of the D sub-step; later ones (and the real
            # pass, if a dry pass ran) redo it and need their own BN exchange
            self._gen_fwd_bwd(args, stats_done and first[0])
            first[0] = False
        cal = getattr(b, "fp8_calibration", None)
        if cal is not None and not cal(1, E.Engine.FP8_CAL_QUERY):
            self._calibrate(1, dry)
            stats_done = False
        b.gen_phase(args, E.G_GEN)
        if self.exact and not stats_done:
            self._allreduce(E.REGION_BN_STATS)
        b.gen_phase(args, E.G_FEAT)
        if self.exact:
            self._allreduce(E.REGION_FM_MOMENTS)
        b.gen_phase(args, E.G_BWD)
        if self.exact:
            self._allreduce(E.REGION_BN_BWD)
        b.gen_phase(args, E.G_TAIL)
        self._allreduce(E.REGION_GRAD_G)
        b.gen_phase(args, E.G_ADAM)

    def train_pair(self, dargs, gargs):
        # Both sub-steps use the same generator weights, so the engine can run their two generator forwards as one
        # two-segment pass inside the D sub-step.  In exact mode the BatchNorm statistics of both segments then travel in
        # the D sub-step's one BN_STATS all-reduce and the G sub-step needs none.  Only when the G sub-step draws its z on
        # the device (the engine could not know a host-supplied z in advance).
        hint = getattr(self.backend, "pair_hint", None)
        paired = hint is not None and not getattr(gargs, "z", None)
        # fp8: settle the D sub-step's scales BEFORE the hint is given.  A hint is good for one D sub-step, and the first
        # dry pass would consume it: the real pass would then run unpaired while gen_step below still skipped its BatchNorm
        # exchange (stats_done), i.e. normalised with unreduced statistics in its first calibration pass.
        self._calibrate(0, lambda: self._disc_fwd_bwd(dargs))
        if paired:
            hint(True)
        self.disc_step(dargs)
        self.gen_step(gargs, stats_done=paired)


def dp_flags(exact=True, graph=False, grad_dtype=None):
    """handle flags of a data-parallel rank.  graph=True: the kernels of every phase range are replayed as captured hipGraphs
    (stream-mode arguments; the collectives stay between the ranges; measured SLOWER than eager launches on one GPU, see
    DESIGN.md section 6).  grad_dtype='bf16': the gradients travel as bfloat16 (DataParallel(grad_dtype='bf16'))."""
    return (E.FLAG_FLAT_GRADS | (E.FLAG_SYNC_STATS if exact else 0) | (E.FLAG_GRAPH if graph else 0) |
            (E.FLAG_GRAD_BF16 if grad_dtype == 'bf16' else 0))
