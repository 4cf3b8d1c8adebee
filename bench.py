#!/usr/bin/env python
"""Headline benchmark: GAN train steps/sec (labeled + unlabeled + G) at batch 4096 (BASELINE.json).

One step = one discriminator sub-step (B labeled + B unlabeled + B generated rows) + one generator
sub-step (B generated + B unlabeled rows), Adam included (mr_gan.py:204-213).  Workload = BASELINE
config 2: synthetic N=65536 x D=512, K=6, batch 4096, bf16 MFMA with fp32 accumulate/master weights.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--global-batch G]
N > 1 is launched by the driver through torch.distributed.run (one rank per GPU, RCCL).  Default: per-GPU batch stays
4096 (weak scaling) and `value` counts batch-4096 steps summed over ranks.  --global-batch G: every rank takes G / N rows
of each stream (strong scaling at a fixed global batch) and `value` counts global-batch steps.

Prints ONE JSON line on rank 0.  `roofline` is measured live: per-launch hipEvent pairs recorded by the
library on the launch stream over a separate profiled pass (never the timed region's numbers re-used),
`cpu_baseline` times the CPU oracle (a port: the reference's Theano/Keras path cannot run)
on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3, "fp8": 5000.0}     # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0                            # HBM3E spec (6.29 TB/s measured by a float4 copy), same guide


def source_sha():
    """identity of the kernel sources a PMC summary under profiles/ was measured on (the GPU box has no .git)"""
    import glob
    import hashlib
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "mr_gan_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "mr_gan_amd", "csrc", "*.h"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def algorithmic_flops(B, D, g_hidden=(500, 500), d_hidden=(1000, 500, 250, 250, 250), K=6, nz=100):
    """2 FLOPs per MAC, GEMMs only, unpadded logical shapes.  Returns per-category FLOPs of
    one (D, G) step pair: forward / input-gradient / weight-gradient products."""
    gd = (nz,) + tuple(g_hidden) + (D,)
    dd = (D,) + tuple(d_hidden) + (K,)
    PG = sum(gd[i] * gd[i + 1] for i in range(3))
    PD = sum(dd[i] * dd[i + 1] for i in range(6))
    PD_head = dd[5] * dd[6]
    PD_body = PD - PD_head                       # the five dense layers run as MFMA GEMMs; the 250x6 head is fused
    FD1, FG1 = dd[0] * dd[1], gd[0] * gd[1]
    # D step: G fwd, 3x D fwd, 3x dW, 3x dX (no dX through the first D layer)
    d_fwd = PG + 3 * PD
    d_dx = 3 * (PD - FD1)
    d_dw = 3 * PD
    # G step: G fwd, 2x feature fwd, dX through the 5 feature layers (fake rows), G dW, G dX (not through layer 1)
    Pmid = PD_body
    g_fwd = PG + 2 * Pmid
    g_dx = Pmid + (PG - FG1)
    g_dw = PG
    tot = 2.0 * B * (d_fwd + d_dx + d_dw + g_fwd + g_dx + g_dw)
    # what the three GEMM kernels execute (the fused head's 2*B*250*6-sized products are excluded from kernel shares)
    head = 2.0 * B * PD_head
    fwd = 2.0 * B * (d_fwd + g_fwd) - 3 * head
    dx = 2.0 * B * (d_dx + g_dx) - 3 * head
    dw = 2.0 * B * (d_dw + g_dw) - 3 * head
    return dict(total=tot, gemm_fwd=fwd, gemm_dx=dx, gemm_dw=dw)


def build_problem(args, rank):
    from mr_gan_amd import select_labeled, synthetic_blobs
    X, y = synthetic_blobs(n=args.rows, d=args.d, seed=1234 + rank)
    mu, sd = X.mean(0), X.std(0)
    X = ((X - mu) / sd).astype(np.float32)                           # StandardScaler (mr_gan.py:96-98)
    xl, yl, _ = select_labeled(X, y, args.labeled_per_class)
    return X, y, xl, yl


def cpu_baseline(args, X, xl, yl, budget_s=12.0):
    """The CPU path timed beside the GPU one on a bounded sample of the same workload.  The reference's Theano/Keras path
    cannot run, so both variants are ports (`kind`): (i) the numpy oracle in fp32 with its noise drawn inside
    the loop, as mr_gan.py:206 does; (ii) a PyTorch-CPU fp32 module of the same step with torch.set_num_threads(all cores)
    and pre-generated noise.  `value` is the faster of the two."""
    from oracle import mrgan_oracle as O
    rng = np.random.default_rng(0)
    B, D = args.batch, args.d
    dims = (D,) + O.D_HIDDEN

    def run_numpy():
        g, d = O.init_params(D, seed=1, dtype=np.float32)
        orc = O.MRGANOracle(g, d)
        noise = lambda: [rng.standard_normal((B, dims[l]), dtype=np.float32) for l in range(5)]

        def one_step():
            il = rng.integers(0, xl.shape[0], B)
            iu = rng.integers(0, X.shape[0], B)
            z = rng.standard_normal((B, O.NOISE_SIZE), dtype=np.float32)
            orc.disc_step(xl[il], yl[il], X[iu], z, noise(), noise(), noise())
            z = rng.standard_normal((B, O.NOISE_SIZE), dtype=np.float32)
            orc.gen_step(X[iu], z, noise(), noise())
        return one_step

    torch_threads = min(os.cpu_count(), 32)      # more threads than this only add contention on these layer sizes (measured: 256
                                                 # threads took 56 s per step on the GPU box's host, 32 are an order of magnitude faster)

    def run_torch():
        import torch
        torch.set_num_threads(torch_threads)
        F = torch.nn.functional
        g, d = O.init_params(D, seed=1, dtype=np.float32)
        gp = [torch.tensor(p, requires_grad=True) for p in g]
        dp = [torch.tensor(p, requires_grad=True) for p in d]
        st = dict(t=0, mg=[torch.zeros_like(p) for p in gp], vg=[torch.zeros_like(p) for p in gp],
                  md=[torch.zeros_like(p) for p in dp], vd=[torch.zeros_like(p) for p in dp])
        pool = [[torch.randn(B, dims[l]) for l in range(5)] for _ in range(5)]      # pre-generated layer noise, re-used
        zs = [torch.randn(B, O.NOISE_SIZE) for _ in range(2)]
        Xt, xlt, ylt = torch.from_numpy(X), torch.from_numpy(xl), torch.from_numpy(yl.astype(np.int64))

        def gen(z):
            h = F.softplus(z @ gp[0] + gp[1])
            mu, var = h.mean(0), h.var(0, unbiased=False)
            h = gp[2] * (h - mu) / torch.sqrt(var + O.BN_EPS) + gp[3]
            return F.softplus(h @ gp[4] + gp[5]) @ gp[6] + gp[7]

        def disc(x, nz, feat=False):
            a = x
            for l in range(5):
                a = torch.relu((a + O.D_SIGMAS[l] * nz[l]) @ dp[2 * l] + dp[2 * l + 1])
            return a if feat else a @ dp[10] + dp[11]

        def adam(ps, gs, ms, vs):
            t = st['t'] + 1
            lr_t = O.ADAM_LR * np.sqrt(1 - O.ADAM_B2 ** t) / (1 - O.ADAM_B1 ** t)
            with torch.no_grad():
                for p, gr, m, v in zip(ps, gs, ms, vs):
                    m.mul_(O.ADAM_B1).add_(gr, alpha=1 - O.ADAM_B1)
                    v.mul_(O.ADAM_B2).addcmul_(gr, gr, value=1 - O.ADAM_B2)
                    p.sub_(lr_t * m / (v.sqrt() + O.ADAM_EPS))
            st['t'] = t

        def one_step():
            il = torch.randint(0, xlt.shape[0], (B,))
            iu = torch.randint(0, Xt.shape[0], (B,))
            l_lab, l_unl = disc(xlt[il], pool[0]), disc(Xt[iu], pool[1])
            l_fake = disc(gen(zs[0]).detach(), pool[2])
            lse = lambda l: torch.logsumexp(l, 1)
            loss = (-l_lab[torch.arange(B), ylt[il]].mean() + lse(l_lab).mean() - 0.5 * lse(l_unl).mean()
                    + 0.5 * F.softplus(lse(l_unl)).mean() + 0.5 * F.softplus(lse(l_fake)).mean())
            adam(dp, torch.autograd.grad(loss, dp), st['md'], st['vd'])
            f_fake, f_real = disc(gen(zs[1]), pool[3], True), disc(Xt[iu], pool[4], True)
            lg = ((f_fake.mean(0) - f_real.mean(0).detach()) ** 2).mean()
            adam(gp, torch.autograd.grad(lg, gp), st['mg'], st['vg'])
        return one_step

    variants = {}
    for name, make, threads in (("numpy_fp32", run_numpy, os.cpu_count()), ("torch_cpu_fp32", run_torch, torch_threads)):
        step = make()
        t0 = time.time()
        step()                                                        # warm-up (thread pools, page-in) -- kept if it is all the budget allows
        n, dt = 1, time.time() - t0
        if dt < budget_s:
            n, t0 = 0, time.time()
            while True:
                step()
                n += 1
                if time.time() - t0 > budget_s or n >= 64:
                    break
            dt = time.time() - t0
        variants[name] = dict(value=n / dt, steps=n, seconds=round(dt, 1), threads=threads)
    best = max(variants, key=lambda k: variants[k]["value"])
    return dict(value=variants[best]["value"], unit="steps/s", cores=variants[best]["threads"], host_cpu_count=os.cpu_count(), kind="port",
                sample="the same workload (B=%d, D=%d), %d + %d steps in %.0f s: oracle/mrgan_oracle.py in numpy fp32 (noise drawn in the "
                       "loop, BLAS on all %d cores) and a PyTorch-CPU fp32 module with torch.set_num_threads(%d) and pre-generated noise; value = the faster (%s)"
                       % (B, D, variants["numpy_fp32"]["steps"], variants["torch_cpu_fp32"]["steps"],
                          variants["numpy_fp32"]["seconds"] + variants["torch_cpu_fp32"]["seconds"], os.cpu_count(), torch_threads, best),
                variants=variants)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--d", type=int, default=512)
    ap.add_argument("--rows", type=int, default=65536)
    ap.add_argument("--hidden", type=int, default=0,
                    help="0 = the reference's layer widths (mr_gan.py:111-128); W = BASELINE configs[4]'s wide stack: five "
                         "discriminator layers and two generator layers of width W")
    ap.add_argument("--g-hidden", type=int, default=0,
                    help="with --hidden: width of the generator's two hidden layers (0 = the same width as the discriminator's; "
                         "500 = BASELINE configs[4] read literally, \"wide D\": only the discriminator is widened)")
    ap.add_argument("--labeled-per-class", type=int, default=100)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "fp8"],
                    help="fp8: the discriminator's dense products on the fp8 matrix cores (BASELINE configs[4] asks for it with "
                         "--hidden 4096 --batch 8192); generator / loss head stay bf16")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling: every rank takes GLOBAL / N rows of each stream (default 0: --batch rows per rank, weak scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--local-stats", action="store_true",
                    help="N > 1: keep the BatchNorm / feature-matching statistics per shard (a different, labelled algorithm: 3 "
                         "fewer small collectives per step). Default: they are all-reduced, so that W ranks at B/W reproduce the "
                         "one-GPU batch-B step")
    ap.add_argument("--profile-steps", type=int, default=10)
    ap.add_argument("--min-seconds", type=float, default=0.5, help="the timed block of --steps steps is repeated until this much time was timed (>= 5 repeats)")
    ap.add_argument("--max-repeats", type=int, default=2000)
    ap.add_argument("--ablate", type=int, default=0, help="timing experiments only (see mrgan_debug_ablate)")
    ap.add_argument("--tune", action="append", default=[], metavar="KNOB=VALUE",
                    help="mrgan_set_tuning experiments, e.g. --tune 1=3 (MRGAN_TUNE_KC_CFG = 3); results stay within rounding")
    ap.add_argument("--grad-dtype", default="f32", choices=["f32", "bf16"],
                    help="N > 1: payload of the gradient all-reduces (bf16 halves the bytes; a labelled, different numerical path)")
    ap.add_argument("--force-dp", action="store_true", help="diagnostic: run the N>1 phase protocol (no all-reduce) on one GPU")
    ap.add_argument("--dp-graph", action="store_true", help="N > 1: replay every phase range as a captured hipGraph (measured slower than eager launches)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from mr_gan_amd import engine as E
    from mr_gan_amd.data import tiled_permutation
    from mr_gan_amd.dist import DataParallel, EngineBackend, dp_flags

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if args.global_batch:
        if args.global_batch % world:
            raise SystemExit("--global-batch must be a multiple of the number of ranks")
        args.batch = args.global_batch // world
    X, y, xl, yl = build_problem(args, rank)
    B, D = args.batch, args.d
    cfg = E.default_config(D, B)
    g_hidden, d_hidden = (500, 500), (1000, 500, 250, 250, 250)
    if args.hidden:
        g_hidden, d_hidden = (args.g_hidden or args.hidden,) * 2, (args.hidden,) * 5
        cfg.g_hidden[0], cfg.g_hidden[1] = g_hidden
        for i, w in enumerate(d_hidden):
            cfg.d_hidden[i] = w
    cfg.dtype = {"bf16": E.BF16, "f32": E.F32, "fp8": E.FP8}[args.dtype]
    cfg.seed = 1
    cfg.rank, cfg.world = rank, world
    use_dp = world > 1 or args.force_dp
    cfg.flags = (dp_flags(exact=not args.local_stats, graph=args.dp_graph, grad_dtype=None if args.grad_dtype == "f32" else args.grad_dtype)
                 if use_dp else (0 if args.no_graph else E.FLAG_GRAPH))
    stream = torch.cuda.Stream(dev)
    with torch.cuda.stream(stream):
        eng = E.Engine(cfg, dev)
        if args.ablate:
            eng.debug_ablate(args.ablate)
        for kv in args.tune:
            eng.set_tuning(*[int(v) for v in kv.split("=")])
        # identical initial weights on every rank
        rs = np.random.RandomState(7)
        for net in (E.NET_G, E.NET_D):
            ws = []
            for i in range(eng.num_tensors(net)):
                shp = eng.full_shape(net, i)
                if len(shp) == 2:
                    lim = np.sqrt(6.0 / (shp[0] + shp[1]))
                    ws.append(rs.uniform(-lim, lim, size=shp).astype(np.float32))
                else:
                    ws.append(np.ones(shp, np.float32) if (net == E.NET_G and i == 2) else np.zeros(shp, np.float32))
            eng.set_weights(net, ws)
        # inputs resident in HBM before the timed region; epoch index streams as in mr_gan.py:189-195
        Xd = torch.from_numpy(X).to(dev)
        xld = torch.from_numpy(xl).to(dev)
        total = max(args.warmup, args.steps, args.profile_steps) + 4
        n_rows = total * B
        prs = np.random.RandomState(11 + rank)
        inds = np.concatenate([tiled_permutation(prs, xl.shape[0], X.shape[0]) for _ in range(-(-n_rows // X.shape[0]))])[:n_rows]
        idx_lab = torch.from_numpy(inds.astype(np.int32)).to(dev)
        lab_stream = torch.from_numpy(yl[inds].astype(np.int32)).to(dev)
        perm = lambda: np.concatenate([prs.permutation(X.shape[0]) for _ in range(-(-n_rows // X.shape[0]))])[:n_rows].astype(np.int32)
        idx_unl = torch.from_numpy(perm()).to(dev)
        idx_unl2 = torch.from_numpy(perm()).to(dev)
        dargs = E.Engine.disc_args(xld, lab_stream, Xd, None, idx_lab, idx_unl, stream_mode=1)
        gargs = E.Engine.gen_args(Xd, None, idx_unl2, stream_mode=1)
        eng.set_iterations(0, 0)
        runner = DataParallel(EngineBackend(eng), exact=not args.local_stats, grad_dtype=None if args.grad_dtype == "f32" else args.grad_dtype) if use_dp else None

        def step():
            if runner is not None:
                runner.train_pair(dargs, gargs)
            else:
                eng.train_pair(dargs, gargs)

        # Index streams hold max(warmup, steps, profile_steps) + 4 batches; the device-side batch counter is rewound before every
        # block (outside the timed regions; the Adam iteration count runs on).
        done = [0]

        def rewind():
            eng.set_iterations(2 * done[0], 0)

        def run(n):
            for _ in range(n):
                step()
            done[0] += n

        def timed_block():
            """EXACTLY --steps steps between barrier + synchronize brackets -> seconds on this rank"""
            rewind()
            torch.cuda.synchronize(dev)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            run(args.steps)
            torch.cuda.synchronize(dev)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)
            return time.perf_counter() - t0

        def over_ranks(ts):
            if world == 1:
                return list(ts)
            t = torch.tensor(ts, dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return [float(v) for v in t.tolist()]

        run(args.warmup)
        torch.cuda.synchronize(dev)
        eng.read_metrics(reset=True)          # train_metrics below cover the timed steps only
        # median of >= 5 repeats of the timed block, and >= 0.5 s of timed work in total whatever --steps is
        blocks = over_ranks([timed_block()])
        repeats = max(5, int(np.ceil(args.min_seconds / max(blocks[0], 1e-6))))
        repeats = min(repeats, args.max_repeats)
        blocks += over_ranks([timed_block() for _ in range(repeats - 1)])
        elapsed = float(np.median(blocks))
        timed_steps = repeats * args.steps
        metrics = eng.read_metrics(reset=True)
        rewind()

        # ---- live per-kernel timing (separate pass, eager launches with hipEvent pairs) ----
        prof = None
        if rank == 0 or world > 1:
            eng.profile_begin()
            for _ in range(args.profile_steps):
                step()
            prof = eng.profile_end()

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = 1e3 * elapsed / args.steps
    strong = args.global_batch > 0
    value = (1 if strong else world) * args.steps / elapsed
    fl = algorithmic_flops(B, D, g_hidden, d_hidden)
    peak = PEAK_TFLOPS[args.dtype]
    ridge = peak * 1e12 / (PEAK_HBM_GBS * 1e9)                       # FLOP per byte at which the two roofs meet (312 for bf16)
    P = float(args.profile_steps)

    def peak_of(k):
        """dense MFMA peak of the arithmetic a kernel computes in (the fp8 mode keeps bf16 kernels for the generator)"""
        return PEAK_TFLOPS["fp8"] if k.startswith("gemm_fp8") else (PEAK_TFLOPS["bf16"] if args.dtype == "fp8" else PEAK_TFLOPS[args.dtype])

    def line(ms, launches, flops, nbytes, peak=peak):
        """roofline entry of one kernel (or family) from its live-measured time and its ALGORITHMIC work"""
        ridge = peak * 1e12 / (PEAK_HBM_GBS * 1e9)
        sec = 1e-3 * ms
        ai = flops / nbytes if nbytes > 0 else None
        d = {"ms_per_step": round(ms / P, 4), "launches_per_step": launches / P,
             "algorithmic_gflop_per_launch": round(flops / max(launches, 1) / 1e9, 3),
             "algorithmic_mb_per_launch": round(nbytes / max(launches, 1) / 1e6, 2) if nbytes > 0 else None,
             "flop_per_byte": round(ai, 1) if ai else None}
        if flops > 0 and (ai is None or ai >= ridge):
            d.update(bound="mfma", achieved=round(flops / sec / 1e12, 2), peak=peak, unit="TFLOP/s", frac=round(flops / sec / 1e12 / peak, 4))
        elif nbytes > 0:
            d.update(bound="hbm", achieved=round(nbytes / sec / 1e9, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(nbytes / sec / 1e9 / PEAK_HBM_GBS, 4))
            if flops > 0:
                d["tflops"] = round(flops / sec / 1e12, 2)
        return d

    def family(k):
        if k.startswith("gemm_fp8_kc_kernel<0"):
            return "fp8_forward"
        if k.startswith("gemm_fp8_kc_kernel<1"):
            return "fp8_input_gradient"
        if k.startswith("gemm_fp8_kc_kernel<2"):
            return "fp8_weight_gradient"
        if k.startswith("gemm_bf16_kc_kernel<0") or k.startswith("gemm_f32_kernel<0"):
            return "kc_forward"
        if k.startswith("gemm_bf16_kc_kernel<1") or k.startswith("gemm_f32_kernel<1"):
            return "kc_input_gradient"
        if k.startswith("gemm_bf16_ks") or k.startswith("gemm_f32_kernel<2"):
            return "ks_weight_gradient"
        if k.startswith("chain_kernel"):
            return "row_block_chain"
        return "elementwise"
    fams, fam_peak = {}, {}
    for k, v in prof.items():
        f = fams.setdefault(family(k), [0.0, 0, 0.0, 0.0])
        fam_peak[family(k)] = peak_of(k)
        for i in range(4):
            f[i] += v[i]
    # dominant kernel = the instantiation with the largest share of device time in the profiled pass
    work = {k: v for k, v in prof.items() if v[2] > 0}
    dom = max(work, key=lambda k: work[k][0])
    all_ms = sum(v[0] for v in prof.values()) / P
    # HBM bytes per launch from the PMC summary under profiles/ (scripts/traffic.sh: separate --pmc passes, gfx950 FETCH_SIZE
    # correction applied there), keyed by the full instantiation name.  The summary carries the hash of the kernel sources it
    # was measured on: a summary of other sources is not evidence for this build and gives null.
    traffic, step_traffic, traffic_src = None, None, None
    tfile = os.path.join(ROOT, "profiles", "r03_traffic.json")
    if world == 1 and not args.hidden and args.dtype == "bf16" and B == 4096 and D == 512 and os.path.exists(tfile):
        tj = json.load(open(tfile))
        if tj.get("source_sha") == source_sha():
            raw = tj.get("kernels", {})

            def row_of(k):
                # the library books a launch under the name it gives the instantiation; rocprofv3 prints the demangled (or, for
                # the templates on the element type, the mangled) symbol: "chain_kernel<0>" is "chain_kernel<0, 2>" there (the
                # rows-per-block parameter), "stage_kernel" is "_ZN5mrgan...12stage_kernelIDF16bEEv..."
                if k in raw:
                    return raw[k]
                stem = k[:-1] + "," if k.endswith(">") else None
                hits = [v for n, v in raw.items() if (stem and n.startswith(stem)) or (not stem and ("%d%s" % (len(k), k)) in n)]
                return hits[0] if len(hits) == 1 else None
            rows = {k: row_of(k) for k in prof}
            src = {"source": "profiles/r03_traffic.json", "source_sha": tj.get("source_sha"), "measured_at_commit": tj.get("commit")}
            traffic_src = src
            if rows.get(dom):
                traffic = round(rows[dom]["hbm_bytes_per_launch"])
            if all(rows.values()):
                step_traffic = round(sum(rows[k]["hbm_bytes_per_launch"] * v[1] / P for k, v in prof.items()))
    dom_line = line(*prof[dom], peak=peak_of(dom))
    # traffic: HBM (fabric) bytes per launch of the dominant kernel / per step for the headline, numbers or null; where they come from
    # goes into traffic_source
    dom_line.update({"kernel": dom, "traffic": traffic, "traffic_unit": "bytes per launch", "avg_launch_us": round(1e3 * prof[dom][0] / max(prof[dom][1], 1), 2)})
    step_tf = fl["total"] / (elapsed / args.steps) / 1e12
    alg_bytes = sum(v[3] for v in prof.values()) / P
    # headline: the WHOLE step against the MFMA roof (650 FLOP per algorithmic byte, MFMA side), with the dominant
    # kernel's own line beside it
    roofline = {
        "bound": "mfma", "achieved": round(step_tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(step_tf / peak, 4),
        "traffic": step_traffic, "traffic_unit": "bytes per step (all kernels of the step)", "traffic_source": traffic_src,
        "scope": "whole (D, G) step: %.2f algorithmic GFLOP (SURVEY.md 8d: 2 FLOP per MAC, GEMMs only, unpadded shapes) over the median timed step"
                 % (fl["total"] / 1e9),
        "algorithmic_mb_per_step": round(alg_bytes / 1e6, 1),
        "flop_per_byte": round(fl["total"] / alg_bytes, 1) if alg_bytes > 0 else None,
        "dominant_kernel": dom_line,
        "timing": "per-kernel figures: hipEvent (start, stop) pairs stamped at each kernel's begin and end on the launch stream "
                  "(hipExtLaunchKernelGGL inside the library), profiled pass of %d steps after the timed region; algorithmic bytes = operands "
                  "once + outputs once at logical shapes (no padding, no split-K slabs)" % args.profile_steps,
        "bound_rule": "per kernel: mfma if algorithmic FLOP per algorithmic byte >= peak TFLOP/s / %.0f GB/s of its arithmetic (312 for bf16, 625 for fp8), else hbm" % PEAK_HBM_GBS,
        "families": {k: line(*v, peak=fam_peak[k]) for k, v in sorted(fams.items(), key=lambda kv: -kv[1][0])},
        "step": {"algorithmic_gflop": round(fl["total"] / 1e9, 2), "launches": sum(v[1] for v in prof.values()) / P,
                 "kernel_ms": {k: round(v[0] / P, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])},
                 "kernel_launches": {k: v[1] / P for k, v in prof.items()},
                 "all_kernels_ms": round(all_ms, 4)},
    }
    out = {
        "metric": "GAN train steps/sec (labeled+unlabeled+G) at batch 4096", "value": round(value, 2), "unit": "steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "repeats": repeats, "timed_steps_total": timed_steps,
        "timing": "median over %d repeats of a block of exactly %d steps, each block between barrier + synchronize brackets "
                  "(max over ranks per block); ms per step of the blocks: min %.4f / median %.4f / max %.4f"
                  % (repeats, args.steps, 1e3 * min(blocks) / args.steps, ms_per_step, 1e3 * max(blocks) / args.steps),
        "config": {"workload": "%s: synthetic N=%d x D=%d, K=6, batch %d per GPU, labeled %d/class; one step = "
                               "D sub-step (3B rows) + G sub-step (2B rows) + both Adam updates"
                               % ("BASELINE configs[4] geometry (hidden %d x 5, generator %d x 2) on one GPU" % (args.hidden, args.g_hidden or args.hidden)
                                  if args.hidden else "BASELINE configs[1]", args.rows, D, B, args.labeled_per_class),
                   "global_batch": B * world, "rows_per_gpu": B, "parallelism": "dp%d" % world if world > 1 else "single",
                   "batch_statistics": ("local_stats (per shard)" if args.local_stats else "synced over ranks") if world > 1 else "n/a",
                   "launch": ("%s phases + RCCL all-reduce (%s gradients)" % ("hipGraph-replayed" if args.dp_graph else "eager", args.grad_dtype)) if use_dp
                             else ("eager" if args.no_graph else "hipGraph replay")},
        "roofline": roofline,
        "train_metrics": {"mean_loss_lab": metrics[0] / timed_steps, "mean_loss_unl": metrics[1] / timed_steps,
                          "mean_train_err": metrics[2] / timed_steps, "mean_loss_gen": metrics[3] / timed_steps},
    }
    if world == 1 and not args.no_cpu_baseline and not args.hidden:
        out["cpu_baseline"] = cpu_baseline(args, X, xl, yl)
    print(json.dumps(out))
    sys.stdout.flush()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
