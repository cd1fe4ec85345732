#!/usr/bin/env python3
"""bench.py -- SpareNet loss/render hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either that very command -- it re-launches itself as N ranks -- or
     python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one synthetic batch (BASELINE.json configs[1] + configs[2]):
    Chamfer distance            fwd + bwd   [B,16384,3] <-> [B,16384,3]
    EMD (auction)               fwd + bwd   eps 0.005, 50 iterations
    expansion penalty           fwd + bwd   primitive_size 512, alpha 1.5
    ComputeDepthMaps render     fwd + bwd   8 views x radius_list, 256 x 256
    scalar losses               all-reduce (RCCL) when N > 1
Scaling (whole clouds are independent, ranks own contiguous slices, no data-path collective):
    --scaling strong (default)  ONE global batch of 32 clouds (seed 1234) split with dist_utils.shard:
                                32 / N clouds per rank -- SURVEY 8(e)'s split, the north star's >= 6x question
    --scaling weak              B = 32 clouds PER RANK, seed 1234 + rank (the same batch as strong at N = 1)
  At N > 1 the other mode is timed after the headline region with the same K / W and reported in
  `other_scaling`, so one driver run per N yields both curves.
  `python bench.py --gpus N` with N > 1 and no launcher around it starts the N ranks itself (torch.distributed.run,
  one process per GPU, RCCL); under a launcher (WORLD_SIZE set) it is one of the ranks.
The four parts are independent given the predicted cloud; by default the renderer runs on a second HIP
stream next to the distance losses, and at <= 16 clouds per rank the expansion penalty on a third
(config.streams = 2 / 3; --no-overlap times the one-stream step).
After the timed region the same steps run once more one stream at a time, untimed for `value`, to
report per-part times and the kernels' uncontended durations (roofline.isolated).
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line:
  value             point pairs per second, whole job (CD pairs 2*B*N*M + EMD effective pairs
                    sum_it sum_b unassigned*n, counted on the device) / wall time of the K steps
  ms_per_step, step_ms_percentiles_rank0   wall / K, and median / p10 / p90 of the per-step durations (HIP
                    events on the main stream at the step boundaries)
  depthmaps_per_sec single-radius 256x256 maps per second over the same wall time
  roofline          the dominant kernel (emd_auction_kernel, one launch per EMD call): `achieved` / `frac` =
                    the work the kernel really ISSUED (matrix-core flops + vector lane operations from committed
                    PMC counters of the SAME build) / its launch time measured live with HIP events on the launch
                    stream, against the fp32 peak; `algorithmic_*` = SURVEY 8d's 14 flop x effective pairs over
                    the same time (the pruned search skips most algorithmic pairs, so that rate can exceed the
                    peak); `wait_frac`, `valu_busy`, `mfma_busy`, `traffic`; the same block for nn_search_kernel,
                    p2i_gather_max_kernel and (other_ops) mds_dense_team_kernel
  cpu_baseline      the CPU oracle (port of the reference algorithm; the Chamfer single-thread leg is the
                    reference's own CPU build when oracle/_ref is present) on a bounded sample, two legs
"""
import argparse
import ctypes
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

if os.environ.get("AB_LIB"):   # A/B a saved build of the library (tools/build_variant.sh); never set by the driver
    import sparenet_amd._lib as _ab
    _ab.LIB_PATH = os.path.abspath(os.environ["AB_LIB"])

from sparenet_amd.dist_utils import reduce_mean_of_means, shard  # noqa: E402

B, N = 32, 16384
EMD_EPS, EMD_ITERS = 0.005, 50
PRIM, ALPHA = 512, 1.5
IMG = 256
N_VIEWS = 8
FLOP_PER_PAIR = {"chamfer_fwd": 9.0, "emd_auction": 14.0}
PEAK_F32_TFLOPS = 157.3                                       # MI355X_MICROARCH.md (vector == f32 MFMA peak)
PEAK_HBM_GBS = 8000.0                                         # MI355X_MICROARCH.md: HBM3E ~8 TB/s
N_SIMD = 256 * 4


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--scaling", choices=("auto", "weak", "strong"), default="auto",
                    help="auto = strong: ONE global batch of 32 clouds split over the ranks (SURVEY 8e; at N = 1 "
                         "both modes are the same 32-cloud workload); the other mode is reported in other_scaling")
    ap.add_argument("--radius-list", type=str, default="5,7,10",
                    help="p2i radii in pixels (reference default, configs/base_config.py:56-60); "
                         "BASELINE.json's literal 0.02,0.05 is near-empty in pixel units")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true",
                    help="time the sequential one-stream step instead of the two-stream one")
    ap.add_argument("--no-network-steps", action="store_true",
                    help="skip the BASELINE config 4 / 5 step timings (sparenet_amd/networks.py), N = 1 only")
    ap.add_argument("--no-emd-regimes", action="store_true",
                    help="skip the auction's ms per call on the four kinds of data (uniform / surface / scatter / untrained)")
    ap.add_argument("--no-other-ops", action="store_true",
                    help="skip the untimed-for-the-headline MDS/gather/gridding/cubic measurements")
    ap.add_argument("--per-view-render", action="store_true",
                    help="render view by view (the reference's loop) instead of all 8 views in one pass")
    ap.add_argument("--no-literal-radii", action="store_true",
                    help="skip the second timed region with BASELINE.json's literal radius_list [0.02, 0.05]")
    ap.add_argument("--no-other-scaling", action="store_true",
                    help="N > 1: skip the second timed region with the other scaling mode")
    args = ap.parse_args()
    if args.scaling == "auto":
        args.scaling = "strong"
    return args


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves, one process
    per GPU (the reference spreads a batch over its GPUs from one command too: runners/base_runner.py:100-104,
    runners/sparenet_runner.py:32-34 -- nn.DataParallel threads there, torch.distributed.run + RCCL here)."""
    import socket
    import subprocess

    shared = os.environ.get("BENCH_DEBUG_SHARED_GPU") == "1"
    have = torch.cuda.device_count()
    if have < args.gpus and not shared:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible; refusing to run fewer ranks "
                         "than asked (BENCH_DEBUG_SHARED_GPU=1 puts all ranks on one GPU over gloo, for debugging)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    raise SystemExit(subprocess.call(cmd, env=env))


def make_inputs(dev, rank, world, scaling):
    """weak: 32 clouds per rank (seed 1234 + rank).  strong: rank's contiguous share of ONE 32-cloud batch."""
    if scaling == "weak":
        g = torch.Generator().manual_seed(1234 + rank)
        pred = torch.rand(B, N, 3, generator=g)
        gt = torch.rand(B, N, 3, generator=g)
    else:
        g = torch.Generator().manual_seed(1234)
        pred = shard(torch.rand(B, N, 3, generator=g), rank, world).contiguous()
        gt = shard(torch.rand(B, N, 3, generator=g), rank, world).contiguous()
    return pred.to(dev), gt.to(dev)


class HotPath:
    """One training-step worth of loss/render ops, composed like the reference runners
    (runners/sparenet_runner.py:83-108, runners/sparenet_gan_runner.py:212-225)."""

    def __init__(self, dev, radius_list, per_view=False):
        from sparenet_amd.cuda.chamfer_distance import ChamferDistance
        from sparenet_amd.cuda.emd.emd_module import emd_forward_raw, emdFunction
        from sparenet_amd.cuda.expansion_penalty.expansion_penalty_module import expansionPenaltyModule
        from sparenet_amd.utils.p2i_utils import ComputeDepthMaps
        from sparenet_amd import _lib

        self.lib = _lib
        self.dev = dev
        self.cd = ChamferDistance()
        self.emd_raw = emd_forward_raw
        self.emd_fn = emdFunction
        self.expansion = expansionPenaltyModule()
        self.render = ComputeDepthMaps("orthorgonal", 1.0, IMG).to(dev)
        self.radius_list = radius_list
        self.per_view = per_view
        self.stats = torch.zeros(2, dtype=torch.int64, device=dev)
        self.last_mean_mst = None
        self.side = None
        self.side2 = None
        self.hi = None
        self.order = os.environ.get("BENCH_ORDER", "auto")   # "auction_first" / "chain" / "one_stream" force an order (A/B); auto: MEASURED
        self.three_streams_env = os.environ.get("BENCH_THREE_STREAMS")   # "0" / "1" force it (A/B); default: measured with the order
        # batch size -> (schedule name, {schedule: ms}) chosen by choose_schedule() during the untimed warm-up
        self.schedule = {}

    def _emd(self, pred, gt):
        """emdFunction with the effective-pair counter attached."""
        fn = self.emd_fn
        raw = self.emd_raw
        stats = self.stats

        class _Counted(torch.autograd.Function):
            @staticmethod
            def forward(ctx, a, b):
                d, asg = raw(a.contiguous(), b.contiguous(), EMD_EPS, EMD_ITERS, stats)
                ctx.save_for_backward(a, b, asg)
                ctx.mark_non_differentiable(asg)
                return d, asg

            @staticmethod
            def backward(ctx, gd, _):
                return fn.backward(ctx, gd, None)[:2]

        return _Counted.apply(pred, gt)

    def _render_all(self, pred):
        """All 8 views x radii, forward + backward; loss = sum over the views of the mean map.  By default the
        views are rendered in ONE pass (ComputeDepthMaps.forward_views: the views join the batch; maps bit-equal
        to the per-view calls); --per-view-render issues the reference's view-by-view loop instead."""
        p4 = (pred.detach() - 0.5).requires_grad_(True)
        if self.per_view:
            acc = None
            for v in range(N_VIEWS):
                maps = self.render(p4, view_id=v, radius_list=self.radius_list)
                s = maps.mean()
                acc = s if acc is None else acc + s
        else:
            maps = self.render.forward_views(p4, range(N_VIEWS), self.radius_list)       # [V,B,R,S,S]
            acc = maps.mean() * N_VIEWS   # = the sum over the views of the mean map (one full reduction)
        acc.backward()
        return acc

    def step_overlapped(self, pred, gt):
        """The same step on two or three HIP streams: the four parts are independent given the predicted cloud.  Main:
        (expansion penalty,) Chamfer, then the auction; second: the renderer; third, at <= 16 clouds per rank: the
        expansion penalty (see three_streams).  The renderer overlaps Chamfer, the expansion penalty and the auction's
        preparation; the persistent auction itself owns every CU it runs on.  Same kernels, same results; per-kernel
        durations stretch under contention.  (Measured and not kept, r03: at <= 8 clouds per rank the auction's XCD-local
        teams leave half of the chip idle, but running renderer + expansion + Chamfer beside it made the step
        SLOWER whichever side was enqueued first: 2.52-2.54 vs 2.26 ms at 4 clouds, 2.86 vs 2.54 at 8.)"""
        main = torch.cuda.current_stream()
        if self.one_stream(pred.size(0)):   # measured: no overlapped order beats the plain sequence at this batch size
            return self.step(pred, gt)
        if self.auction_first(pred.size(0)):
            return self._step_auction_first(pred, gt, main)
        if self.side is None:
            self.side = torch.cuda.Stream()
        # tensors that cross streams are registered with the caching allocator: a block freed on its own stream
        # could otherwise be handed out again while the other stream still reads it
        pred.record_stream(self.side)
        self.side.wait_stream(main)
        with torch.cuda.stream(self.side):
            acc = self._render_all(pred)
        acc.record_stream(main)
        if self.three_streams(pred.size(0)):
            # the expansion penalty is one lone wave per 512-point patch for ~0.45 ms (latency bound, the SIMDs nearly
            # idle): on a third stream it runs BESIDE Chamfer instead of in front of it
            if self.side2 is None:
                self.side2 = torch.cuda.Stream()
            pred.record_stream(self.side2)
            self.side2.wait_stream(main)
            with torch.cuda.stream(self.side2):
                loss_exp = self._loss_expansion(pred)
            loss_exp.record_stream(main)
            loss_cd = self._loss_cd(pred, gt)
            loss_emd = self._loss_emd(pred, gt)
            main.wait_stream(self.side2)
        else:   # (the expansion penalty AFTER the auction instead of in front of Chamfer: 5.71 against 5.51 ms)
            loss_cd, loss_emd, loss_exp = self._distance_losses(pred, gt)
        main.wait_stream(self.side)
        losses = torch.stack([loss_cd.detach(), loss_emd.detach(), loss_exp.detach(), acc.detach()])
        return reduce_mean_of_means(losses)   # RCCL all-reduce over xGMI when N > 1

    def auction_first(self, clouds):
        """Which order?  The persistent auction needs every CU's registers (16 waves x 128 VGPRs), so nothing runs
        beside it; what can overlap is everything ELSE.  From 24 clouds per rank on, the auction is enqueued first, on a
        high-priority stream, and the renderer, Chamfer and the expansion penalty share the chip after it on two more
        streams (the lone waves of the expansion penalty and Chamfer's search fill the renderer's gaps): 4.90 -> 4.79
        ms per step at 32 clouds, 4.62 -> 4.46 with round 4's gather (profiles/r04_c_share_by_order.txt).  Below that the chain
        expansion | Chamfer -> auction with the renderer beside it stays (16 / 8 / 4 clouds: 3.18 / 1.96 / 1.54-1.65 ms
        against 3.16-3.45 / 2.01 / 1.78: the auction's teams leave XCDs idle there, and the chain's head overlaps the
        renderer)."""
        if self.order != "auto":
            return self.order == "auction_first"
        if clouds in self.schedule:
            return self.schedule[clouds][0] == "auction_first"
        return clouds >= 24   # before / without choose_schedule(): round 4's table

    def one_stream(self, clouds):
        if self.order != "auto":
            return self.order == "one_stream"
        return clouds in self.schedule and self.schedule[clouds][0] == "one_stream"

    SCHEDULES = ("one_stream", "chain_2", "chain_3", "auction_first")

    def choose_schedule(self, pred, gt, reps=6, reduce_max=None):
        """Time every schedule on THIS batch during the untimed warm-up and keep the fastest (round 5 shipped a constant
        threshold -- auction first from 24 clouds on, a third stream up to 16 -- under which the 16-cloud share of the
        strong split ran 4.07 ms against 3.32 for its own one-stream sequence).  Per schedule: 2 untimed + `reps` timed
        steps between synchronisations, wall clock (what the timed region measures).  The plain one-stream sequence is a
        candidate, so the step is never slower than it by choice.  reduce_max: callable that max-reduces a list of
        floats over the ranks (every rank then picks the same schedule); None on one rank.  BENCH_ORDER /
        BENCH_THREE_STREAMS still force an order (A/B)."""
        clouds = pred.size(0)
        if self.order != "auto" or self.three_streams_env in ("0", "1") or clouds in self.schedule:
            return self.schedule.get(clouds)
        table = {}
        for name in self.SCHEDULES:
            self.schedule[clouds] = (name, None)
            for _ in range(2):
                self.step_overlapped(pred, gt)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                self.step_overlapped(pred, gt)
            torch.cuda.synchronize()
            table[name] = (time.perf_counter() - t0) / reps * 1e3
        if reduce_max is not None:
            vals = reduce_max([table[k] for k in self.SCHEDULES])
            table = dict(zip(self.SCHEDULES, vals))
        best = min(self.SCHEDULES, key=lambda k: table[k])
        self.schedule[clouds] = (best, table)
        return self.schedule[clouds]

    def _step_auction_first(self, pred, gt, main):
        """The auction on a HIGH-PRIORITY stream, enqueued first; the renderer and Chamfer + expansion penalty on two
        more streams.  The auction's launch duration measured LIVE includes its wait for the previous step's tail to
        leave the CUs (`roofline.isolated` is the kernel's own time)."""
        if self.hi is None:
            self.hi = torch.cuda.Stream(priority=-1)
            self.side = self.side or torch.cuda.Stream()
        for st in (self.hi, self.side):
            pred.record_stream(st)
            gt.record_stream(st)
            st.wait_stream(main)
        with torch.cuda.stream(self.hi):
            loss_emd = self._loss_emd(pred, gt)
        loss_emd.record_stream(main)
        # (Round 6 measured "nothing else before the auction is DONE" -- the other streams wait for the high-priority
        # stream -- on the theory that the gather and the expansion penalty's lone waves reach the compute units during
        # the auction's preparation kernels and make the persistent grid wait for them: 4.59 ms against 4.32 at 32
        # clouds, slower at every share (profiles/r06_g_strong_share.txt); what overlaps the auction's preparation and
        # tail is worth more than the clean start.  Not kept.)
        # (Round 6 also timed two variants of this order as candidates -- Chamfer before the expansion penalty, and
        # Chamfer + expansion only after the auction: within 1 % of this one or slower at every share
        # (profiles/r06_m_strong_share.txt), and one more candidate each for the warm-up to choose between: removed.)
        with torch.cuda.stream(self.side):
            acc = self._render_all(pred)
        acc.record_stream(main)
        loss_exp = self._loss_expansion(pred)
        loss_cd = self._loss_cd(pred, gt)
        main.wait_stream(self.hi)
        main.wait_stream(self.side)
        losses = torch.stack([loss_cd.detach(), loss_emd.detach(), loss_exp.detach(), acc.detach()])
        return reduce_mean_of_means(losses)

    def three_streams(self, clouds):
        """Third stream for the expansion penalty?  At <= 16 clouds per rank (the strong-scaling shares) it takes the
        penalty's 0.36-0.40 ms off the critical path: 3.44 -> 3.34, 2.52 -> 2.36, 2.25 -> 1.98 ms per step at 16 / 8 /
        4 clouds.  At 32 clouds the step gains 2 % (5.52 -> 5.40 ms) but the persistent auction then waits ~0.3 ms
        for compute units that the 1024 lone waves -- slower beside renderer and Chamfer -- still occupy, which
        stretches its LIVE launch duration (2.83 ms against 2.52 uncontended and in the rocprofv3 summary): there the
        two-stream step is kept, so that the roofline's live duration stays the kernel's."""
        if self.three_streams_env in ("0", "1"):
            return self.three_streams_env == "1"
        if clouds in self.schedule:
            return self.schedule[clouds][0] == "chain_3"
        return clouds <= 16   # before / without choose_schedule(): round 3's table

    def _loss_expansion(self, pred):
        # one wave per 512-point patch = lone waves for 0.35-0.45 ms, latency bound: next to another stream's work
        # it costs nothing, at the end of a chain it runs alone
        p3 = pred.detach().requires_grad_(True)
        pen, _, mml = self.expansion(p3, PRIM, ALPHA)
        loss_exp = pen.mean()
        loss_exp.backward()
        self.last_mean_mst = mml.detach()
        return loss_exp

    def _loss_cd(self, pred, gt):
        p = pred.detach().requires_grad_(True)
        g = gt.detach().requires_grad_(True)
        d1, d2 = self.cd(p, g)
        loss_cd = d1.mean() + d2.mean()
        loss_cd.backward()
        return loss_cd

    def _loss_emd(self, pred, gt):
        p2 = pred.detach().requires_grad_(True)
        dist_, _ = self._emd(p2, gt)
        loss_emd = torch.sqrt(dist_).mean(1).mean()
        loss_emd.backward()
        return loss_emd

    def _distance_losses(self, pred, gt, mark=lambda name: None):
        loss_exp = self._loss_expansion(pred)
        mark("expansion")
        loss_cd = self._loss_cd(pred, gt)
        mark("cd")
        loss_emd = self._loss_emd(pred, gt)
        mark("emd")
        return loss_cd, loss_emd, loss_exp

    def step(self, pred, gt, timers=None):
        """Sequential step on the current stream, with per-part event marks."""
        def mark(name):
            if timers is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                timers.append((name, e))

        mark("start")
        loss_cd, loss_emd, loss_exp = self._distance_losses(pred, gt, mark)
        acc = self._render_all(pred)
        mark("render")
        losses = torch.stack([loss_cd.detach(), loss_emd.detach(), loss_exp.detach(), acc.detach()])
        losses = reduce_mean_of_means(losses)   # RCCL all-reduce over xGMI when N > 1
        mark("allreduce")
        return losses


def timed_region(hp, pred, gt, steps, warmup, overlap, barrier, before_timed=lambda: None, reduce_max=None):
    """W untimed + exactly K timed steps between barrier + synchronize; returns (seconds, per-step ms, losses)."""
    run_step = hp.step_overlapped if overlap else hp.step
    if overlap:   # untimed: which stream order is fastest on this batch (kept per batch size)
        hp.choose_schedule(pred, gt, reduce_max=reduce_max)
    for _ in range(warmup):
        run_step(pred, gt)
    barrier()
    hp.stats.zero_()
    before_timed()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    barrier()
    t0 = time.perf_counter()
    marks[0].record()
    losses = None
    for k in range(steps):
        losses = run_step(pred, gt)
        marks[k + 1].record()      # main stream: after the renderer's stream has been joined
    barrier()
    elapsed = time.perf_counter() - t0
    per_step = [marks[k].elapsed_time(marks[k + 1]) for k in range(steps)]
    return elapsed, per_step, losses


def percentiles(v):
    s = sorted(v)
    pick = lambda q: s[min(len(s) - 1, max(0, int(round(q * (len(s) - 1)))))]
    return {"median": pick(0.5), "p10": pick(0.1), "p90": pick(0.9), "min": s[0], "max": s[-1]}


def cpu_baseline():
    """The CPU restatement of the reference (oracle/, validated against the reference's own CPU build and its
    golden vectors) timed on this box's host cores, on a bounded sample of the benched workload, two legs:
    (i) one thread -- what the reference's single-threaded CPU code does (Chamfer: the reference's OWN build
    when oracle/_ref is present); (ii) OpenMP over clouds on all cores -- best-effort CPU."""
    import numpy as np
    import oracle

    cores = os.cpu_count() or 1
    g = torch.Generator().manual_seed(1234)
    pred_t = torch.rand(B, N, 3, generator=g)
    gt_t = torch.rand(B, N, 3, generator=g)
    pred, gt = pred_t.numpy(), gt_t.numpy()
    radii = [5.0, 7.0, 10.0]

    from sparenet_amd.utils.p2i_utils import ComputeDepthMaps
    cdm = ComputeDepthMaps("orthorgonal", 1.0, IMG)

    def render_inputs(nb):
        ij, feat = cdm.project(torch.from_numpy(pred[:nb]) - 0.5, 0)
        px = ((ij + 1) / 2 * (IMG - 1)).numpy()
        return px, feat.numpy(), np.repeat(np.arange(nb, dtype=np.int32), N), np.zeros((nb, 1, IMG, IMG), np.float32)

    def clock(fn):
        t0 = time.perf_counter()
        r = fn()
        return time.perf_counter() - t0, r

    # ---- leg (i): one thread
    kind_cd = "port"
    t_cd1 = None
    try:
        from oracle import ref as oref
        if oref.available():
            t_cd1, _ = clock(lambda: oref.chamfer_forward(pred_t[:1], gt_t[:1]))
            kind_cd = "reference"
    except Exception:
        t_cd1 = None
    if t_cd1 is None:
        t_cd1, _ = clock(lambda: oracle.chamfer_forward(pred[:1], gt[:1]))
    t_emd1, (_, _, aux1) = clock(lambda: oracle.emd_forward(pred[:1], gt[:1], EMD_EPS, EMD_ITERS, return_aux=True))
    px, ft, bi, bg = render_inputs(2)
    t_p2i1, _ = clock(lambda: [oracle.p2i_max_forward(px, ft, bi, bg, r) for r in radii])
    single = {
        "value": (2.0 * N * N + float(aux1["pairs_eff"])) / (t_cd1 + t_emd1),
        "chamfer_pairs_per_sec": 2.0 * N * N / t_cd1, "chamfer_kind": kind_cd,
        "emd_pairs_per_sec": float(aux1["pairs_eff"]) / t_emd1,
        "depthmaps_per_sec": 2 * len(radii) / t_p2i1,
        "sample": (f"1 thread: Chamfer fwd on 1 of 32 clouds ({kind_cd}: "
                   + ("the reference's chamfer_distance.cpp compiled unmodified" if kind_cd == "reference" else "oracle")
                   + f", {t_cd1:.2f} s) + EMD fwd 50 it on 1 cloud (oracle, {t_emd1:.2f} s) + "
                   f"p2i max fwd on 2 clouds x 1 view x 3 radii (oracle, {t_p2i1:.2f} s)"),
        # one step's forward work on one thread, extrapolated from the sample
        "step_forward_equivalent_s": B * t_cd1 + B * t_emd1 + (B / 2.0) * N_VIEWS * t_p2i1,
    }

    # ---- leg (ii): OpenMP, all cores, all 32 clouds
    t_cd, _ = clock(lambda: oracle.chamfer_forward(pred, gt, mt=True))
    t_emd, (_, _, aux) = clock(lambda: oracle.emd_forward(pred, gt, EMD_EPS, EMD_ITERS, mt=True, return_aux=True))
    px, ft, bi, bg = render_inputs(B)
    t_p2i, _ = clock(lambda: [oracle.p2i_max_forward(px, ft, bi, bg, r, mt=True) for r in radii])
    pairs_cd, pairs_emd = 2.0 * B * N * N, float(aux["pairs_eff"])
    return {
        "value": (pairs_cd + pairs_emd) / (t_cd + t_emd),
        "unit": "point-pairs/s",
        "cores": cores,
        "kind": "port",
        "sample": (f"oracle (C, OpenMP x{cores} threads) on the benched batch: Chamfer fwd 32 clouds "
                   f"({pairs_cd:.3g} pairs, {t_cd:.2f} s) + EMD fwd eps {EMD_EPS} iters {EMD_ITERS} 32 clouds "
                   f"({pairs_emd:.3g} effective pairs, {t_emd:.2f} s) + p2i max fwd 32 clouds x 1 of 8 views x 3 "
                   f"radii, clouds in parallel ({t_p2i:.2f} s)"),
        "chamfer_pairs_per_sec": pairs_cd / t_cd,
        "emd_pairs_per_sec": pairs_emd / t_emd,
        "depthmaps_per_sec": B * len(radii) / t_p2i,
        "step_forward_equivalent_s": t_cd + t_emd + N_VIEWS * t_p2i,
        "single_thread": single,
    }


def other_ops(dev, pred, mean_mst):
    """SURVEY 8(a) rows that are not part of the loss step (they run inside the generator's
    forward): timed once each, outside the headline region, for the record."""
    from sparenet_amd.cuda.MDS.MDS_module import minimum_density_sample, gather_operation
    from sparenet_amd.cuda.gridding import Gridding, GriddingReverse
    from sparenet_amd.cuda.cubic_feature_sampling import CubicFeatureSampling

    def ms(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    g = torch.Generator(device="cpu").manual_seed(99)
    extra = torch.rand(B, 3000, 3, generator=g).to(dev)
    cloud = torch.cat([pred.detach(), extra], dim=1).contiguous()          # [32,19384,3]
    out = {}
    idx = minimum_density_sample(cloud, N, mean_mst)
    out["mds_19384_to_16384"] = ms(lambda: minimum_density_sample(cloud, N, mean_mst))
    # the same op on surface-like data: 32 compact patches of 512 points on a sphere of radius 0.5
    # (what a decoder's primitives look like) + 3000 points of the partial input; the expansion's
    # own mean_mst_length is then ~0.01 instead of ~0.05 and the cut radius covers ~5 % of the cloud
    from sparenet_amd.cuda.expansion_penalty.expansion_penalty_module import expansionPenaltyModule
    surf = surface_like(B, N, g).to(dev)
    _, _, mml_s = expansionPenaltyModule()(surf, PRIM, ALPHA)
    cloud_s = torch.cat([surf, surf[:, :3000] + 0.01 * torch.randn(B, 3000, 3, generator=g).to(dev)],
                        dim=1).contiguous()
    out["mds_19384_to_16384_surface"] = ms(lambda: minimum_density_sample(cloud_s, N, mml_s))
    out["mds_surface_mean_mst_length"] = float(mml_s.mean())
    out["mds_uniform_mean_mst_length"] = float(mean_mst.mean())
    feat = torch.cat([cloud, cloud[:, :, :1]], dim=2).transpose(1, 2).contiguous()   # [32,4,19384]
    out["gather_c4"] = ms(lambda: gather_operation(feat, idx))
    pts = ((pred.detach() - 0.5) * 1.9).requires_grad_(True)              # inside (-1,1)
    grid_op, rev_op, cubic = Gridding(64), GriddingReverse(64), CubicFeatureSampling()

    def gridding_fb():
        grid = grid_op(pts)
        grid.sum().backward()
    out["gridding64_fwd_bwd"] = ms(gridding_fb)
    vol = torch.rand(B, 64, 64, 64, generator=g).to(dev).requires_grad_(True)

    def reverse_fb():
        rev_op(vol).sum().backward()
    out["gridding_reverse64_fwd_bwd"] = ms(reverse_fb)
    feats = torch.rand(B, 32, 32, 32, 32, generator=g).to(dev).requires_grad_(True)
    q = (pred.detach()[:, :2048] * 30.0 + 0.5).contiguous()

    def cubic_fb():
        cubic(q, feats).sum().backward()
    out["cubic_sampling_2048x32c_fwd_bwd"] = ms(cubic_fb)
    # EdgeConv graph of the generator (models/sparenet_generator.py:192-209): 3000 input points, k = 8
    from sparenet_amd.cuda.knn import get_graph_feature, knn, knn_unfused
    xf = torch.rand(B, 256, 3000, generator=g).to(dev).requires_grad_(True)
    from sparenet_amd.cuda.knn import knn_fused
    out["knn_k8_c256_n3000"] = ms(lambda: knn(xf.detach(), 8))            # the default routing
    out["knn_k8_c256_n3000_fused_mfma"] = ms(lambda: knn_fused(xf.detach(), 8))
    out["knn_k8_c256_n3000_gemm_plus_rank"] = ms(lambda: knn_unfused(xf.detach(), 8))
    x3 = torch.rand(B, 3, 3000, generator=g).to(dev)
    out["knn_k8_c3_n3000_fused_mfma"] = ms(lambda: knn_fused(x3, 8))
    out["knn_k8_c3_n3000_gemm_plus_rank"] = ms(lambda: knn_unfused(x3, 8))
    nbr = knn(xf.detach(), 8)

    def graph_fb():
        get_graph_feature(xf, k=8, idx=nbr).sum().backward()
    out["graph_feature_c256_fwd_bwd"] = ms(graph_fb)
    return out


def surface_like(b, n, g):
    """[b, n, 3] points on a sphere of radius 0.5, ordered so that every run of 512 consecutive points is a compact
    patch -- what a trained decoder's 32 primitives look like (mean MST length ~0.01 instead of ~0.08 on a uniform
    cube: the sampler's surface regime)."""
    v = torch.randn(b, n, 3, generator=g)
    v = 0.5 * v / v.norm(dim=2, keepdim=True)
    key = (torch.atan2(v[..., 1], v[..., 0]) * 4).floor() * 100 + (v[..., 2] * 8).floor()
    return torch.gather(v, 1, key.argsort(dim=1).unsqueeze(-1).expand(-1, -1, 3)).contiguous()


EMD_REGIMES = ("uniform", "surface", "scatter", "untrained")
_UNTRAINED_CACHE = {}


def emd_regime_clouds(name, b, dev, seed=1234):
    """(prediction, ground truth), [b, N, 3] each: the four kinds of data a training run hands the auction.
    uniform   -- two independent uniform cubes (the benchmark's clouds; make_inputs draws the same numbers);
    surface   -- ground truth on a sphere in 512-point patches, prediction = ground truth + 1 % noise (a trained generator);
    scatter   -- prediction = ground truth + uniform offsets up to +-0.3 (early training);
    untrained -- the `refine` output of networks.Generator at RANDOM INIT on a partial view of that ground truth
                 (residual offsets up to +-1: what configs 4-5 feed the auction in the first steps of every run)."""
    g = torch.Generator().manual_seed(seed)
    if name == "uniform":
        x = torch.rand(b, N, 3, generator=g)
        y = torch.rand(b, N, 3, generator=g)
        return x.to(dev), y.to(dev)
    gt = surface_like(b, N, g)
    if name == "surface":
        return (gt + 0.01 * torch.randn(b, N, 3, generator=g)).contiguous().to(dev), gt.to(dev)
    if name == "scatter":
        return (gt + 0.3 * (2 * torch.rand(b, N, 3, generator=g) - 1)).contiguous().to(dev), gt.to(dev)
    if name != "untrained":
        raise ValueError(name)
    key = (b, seed, str(dev))
    if key not in _UNTRAINED_CACHE:
        from sparenet_amd import networks as nw
        gtd = gt.to(dev)
        partial = (gtd[:, torch.randperm(N, generator=g)[:3000]] + 1e-3 * torch.randn(b, 3000, 3, generator=g).to(dev)).contiguous()
        torch.manual_seed(0)
        gen = nw.Generator(num_points=N, n_primitives=32).to(dev).train()   # batch statistics, as in a training step
        outs = []
        with torch.no_grad():
            for i in range(0, b, 8):
                outs.append(gen(partial[i:i + 8])[2].float())
        del gen
        _UNTRAINED_CACHE[key] = (torch.cat(outs).contiguous(), gtd)
    return _UNTRAINED_CACHE[key]


NETWORK_STATES = ("random_init", "scattered_stand_in", "trained_stand_in_damped")


def make_network_step(dev, cfg, state, batch_terms=True):
    """One rank's share of BASELINE config 4 (reconstruction step, 4 clouds) or config 5 (GAN step, 8 clouds) at the
    stated sizes, as a callable: forward + backward + optimiser step(s).  Three generator states:
      `random_init`              what every run STARTS with: the decoder's output fills the cube, the refine stages'
                                 untrained PointNet residual moves every point by up to +-1 (the sampler's dense
                                 regime, a contested auction that does not converge in its 50 iterations);
      `scattered_stand_in`       early training: the decoder's output replaced by the ground truth scattered +-0.3
                                 (its own computation kept in the graph with weight 0), the residual offsets damped to 10 %;
      `trained_stand_in_damped`  a trained generator: decoder output = ground truth + 1 % noise on a surface in
                                 512-point patches, residual offsets damped to 1 % (round 4's `trained_stand_in`:
                                 the key was renamed with the damping, round 3's stand-in left the residual alone)."""
    from sparenet_amd import networks as nw
    from sparenet_amd.harness import Completion, GanStep

    class _StandIn(torch.nn.Module):
        def __init__(self, dec, surf):
            super().__init__()
            self.dec, self.surf = dec, surf

        def forward(self, style):
            return self.surf + 0.0 * self.dec(style)

    class _Damped(torch.nn.Module):
        def __init__(self, net, scale):
            super().__init__()
            self.net, self.scale = net, scale

        def forward(self, x):
            return self.scale * self.net(x)

    b = {"config4": 4, "config5": 8}[cfg]
    g = torch.Generator().manual_seed(4 if cfg == "config4" else 5)
    gt = surface_like(b, N, g).to(dev)
    partial = (gt[:, torch.randperm(N, generator=g)[:3000]] + 1e-3 * torch.randn(b, 3000, 3, generator=g).to(dev)).contiguous()
    torch.manual_seed(0)
    gen = nw.Generator(num_points=N, n_primitives=32).to(dev)
    if state in ("trained_stand_in_damped", "scattered_stand_in"):
        if state == "trained_stand_in_damped":
            surf, damp = gt + 0.01 * torch.randn(b, N, 3, generator=g).to(dev), 0.01
        else:
            surf, damp = gt + 0.3 * (2 * torch.rand(b, N, 3, generator=g).to(dev) - 1), 0.1
        gen.decoder = _StandIn(gen.decoder, surf.transpose(1, 2).contiguous())
        if gen.refine is not None:
            gen.refine.residual = _Damped(gen.refine.residual, damp)
    elif state != "random_init":
        raise ValueError(state)
    opt_g = torch.optim.Adam(gen.parameters(), lr=1e-4)
    comp = Completion("emd", batch_terms=batch_terms).to(dev)
    if cfg == "config4":
        def step():
            loss, *_ = comp(gen, partial, gt)
            opt_g.zero_grad(set_to_none=True)
            loss.backward()
            opt_g.step()
            return loss
        return step
    disc = nw.PatchDiscriminator((16, IMG, IMG)).to(dev)
    gan = GanStep(gen, disc, comp, opt_g, torch.optim.Adam(disc.parameters(), lr=1e-4))
    return lambda: gan(partial, gt)


def network_steps(dev):
    """BASELINE configs 4-5 at their stated sizes (16384 output / 3000 input points, 32 primitives, hide 4096), one
    rank's share of the 8-GPU job (4 resp. 8 clouds), with the reference's networks restated in
    sparenet_amd/networks.py (bf16 autocast around the fp32 HIP ops) -- outside the headline region, for the
    record.  Three generator states each (make_network_step)."""
    spread = {}

    def clock(fn, key, reps=5):
        """median of `reps` individually timed steps after two warm-up steps; min / max go to `spread`"""
        fn()
        fn()
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        spread[key] = {"min": ts[0], "median": ts[len(ts) // 2], "max": ts[-1], "reps": reps}
        return ts[len(ts) // 2]

    out = {"schema": 5,
           "headline_state": "random_init"}
    for cfg in ("config4", "config5"):
        for state in NETWORK_STATES:
            step = make_network_step(dev, cfg, state)
            out[f"step_ms_{cfg}_{state}"] = clock(step, f"{cfg}_{state}")
            del step
    # the three EMD terms as three auction calls (round 4's form), for the record
    step = make_network_step(dev, "config4", "random_init", batch_terms=False)
    out["step_ms_config4_random_init_three_auction_calls"] = clock(step, "config4_random_init_three_auction_calls")
    del step
    out["spread_ms"] = spread
    out["note"] = ("median of 5 individually timed steps (spread_ms: min / median / max); one rank's share of the 8-GPU "
                   "job: config 4 = 4 clouds (global batch 32), config 5 = 8 clouds (global batch 64); EMD metric, its three "
                   "terms through one auction call (harness.Completion._metrics); forward + backward + optimiser step(s); "
                   "the GAN step renders all 8 views of a cloud set in one pass.  States: make_network_step; the "
                   "headline is random_init -- what a run starts with.  schema 5: `trained_stand_in` of rounds 3-4 is "
                   "`trained_stand_in_damped` (round 4 added the 1 % damping of the refine residual under the old key)")
    return out


def emd_regimes(dev):
    """The auction alone, forward, ms per call, on the four kinds of data of emd_regime_clouds at 32 and 4 clouds (the
    1- and 8-GPU shares): the headline's uniform cubes are its EASIEST input."""
    from sparenet_amd.cuda.emd.emd_module import emd_forward_raw

    out = {}
    for name in EMD_REGIMES:
        for b in (32, 4):
            x, y = emd_regime_clouds(name, b, dev)
            for _ in range(2):
                emd_forward_raw(x, y, EMD_EPS, EMD_ITERS)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                emd_forward_raw(x, y, EMD_EPS, EMD_ITERS)
            e1.record()
            torch.cuda.synchronize()
            out[f"{name}_b{b}_ms_per_call"] = e0.elapsed_time(e1) / 5
    _UNTRAINED_CACHE.clear()
    out["note"] = ("sn_emd_forward, eps 0.005, 50 iterations, [b, 16384, 3]; uniform = the benchmark's cubes; surface = "
                   "prediction = ground truth + 1 % noise (a trained generator); scatter = +-0.3 around the surface (early "
                   "training); untrained = the refine output of networks.Generator at random init (6000-8000 of 16384 "
                   "bidders unassigned in every iteration).  Round 4's library: 2.05 / 1.04, 1.33 / 0.88, 4.85 / 2.57, "
                   "79 / 14.8 ms (profiles/r05_b_emd_regimes.txt)")
    return out


def committed_counters(build_id, kernel):
    """Hardware counters of `kernel` from the newest profiles/*pmc*.json taken on THIS build of the library
    (tools/pmc_all.sh stamps sn_build_id into the file); None if there is none -- counters of other code are
    never quoted."""
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*pmc*.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("_build_id") == build_id and kernel in d:
            best = (os.path.basename(f), d[kernel])
    return best


def device_identity(dev):
    """Something that tells two GPUs apart: the UUID where torch exposes it, else the PCI location."""
    pr = torch.cuda.get_device_properties(dev)
    u = getattr(pr, "uuid", None)
    if u is not None:
        return str(u)
    return "pci-%s:%s:%s" % tuple(getattr(pr, k, "?") for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))


def main():
    args = parse()
    radius_list = [float(r) for r in args.radius_list.split(",")]
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args)           # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    shared = os.environ.get("BENCH_DEBUG_SHARED_GPU") == "1"   # debugging aid: N ranks on ONE GPU over gloo
    if shared:
        local = 0
    elif local >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local)   # before the process group: RCCL binds to the current device
    dev = torch.device("cuda", local)
    rccl_ranks = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("gloo" if shared else "nccl", rank=rank, world_size=world)
        ids = [None] * world
        dist.all_gather_object(ids, device_identity(dev))
        rccl_ranks = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "devices": ids,
                      "distinct_devices": len(set(ids))}
        if not shared and len(set(ids)) != world:
            raise SystemExit(f"bench.py: {world} ranks but only {len(set(ids))} distinct GPUs: {ids}")
    if B % world:
        raise SystemExit(f"the strong split shards {B} clouds: world size {world} must divide it")

    pred, gt = make_inputs(dev, rank, world, args.scaling)
    b_local = pred.size(0)
    hp = HotPath(dev, radius_list, args.per_view_render)
    lib = hp.lib.lib()
    build_id = lib.sn_build_id().decode()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    overlap = not args.no_overlap

    def reduce_max(vals):   # every rank keeps the same schedule: the slowest rank's time decides
        if world == 1:
            return vals
        t = torch.tensor(vals, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t.tolist()]

    def prof_on():   # the in-library launch timers cover exactly the timed steps
        if not args.no_roofline:
            lib.sn_prof_reset()
            lib.sn_emd_prof_exec(None, 1)
            lib.sn_prof_enable(1)

    elapsed, per_step, losses = timed_region(hp, pred, gt, args.steps, args.warmup, overlap, barrier, prof_on, reduce_max)
    lib.sn_prof_enable(0)
    hp.lib.check(lib.sn_device_status(), "timed region")   # a team barrier that gave up inside it (NaN losses) is an error
    stats_timed = hp.stats.clone()

    def read_kernels():
        ks = {}
        for kname in ("chamfer_fwd", "nn_search", "emd_auction", "expansion_fwd", "p2i_max_splat", "mds"):
            ms = ctypes.c_double(0.0)
            cnt = lib.sn_prof_read(kname.encode(), ctypes.byref(ms))
            ks[kname] = {"launches": int(cnt), "total_ms": ms.value,
                         "avg_us": (ms.value / cnt * 1e3) if cnt else None}
        # the auction's own execution window (in-kernel clock: first working workgroup's start to the last one's end);
        # the HIP-event bracket above also contains the launch's wait for every compute unit to be empty
        ms = ctypes.c_double(0.0)
        cnt = lib.sn_emd_prof_exec(ctypes.byref(ms), 1)
        ks["emd_auction_exec"] = {"launches": int(cnt), "total_ms": ms.value,
                                  "avg_us": (ms.value / cnt * 1e3) if cnt else None}
        return ks

    kernels = read_kernels() if not args.no_roofline else {}
    # outside the timed region: the same steps one stream at a time, for the per-part times and for
    # the kernels' uncontended durations
    kernels_isolated, isolated_ms, timers = {}, None, []
    iso_steps = min(args.steps, 10)
    if not args.no_roofline:
        lib.sn_prof_reset()
        lib.sn_emd_prof_exec(None, 1)
        lib.sn_prof_enable(1)
        barrier()
        t1 = time.perf_counter()
        for _ in range(iso_steps):
            hp.step(pred, gt, timers)
        barrier()
        isolated_ms = (time.perf_counter() - t1) / iso_steps * 1e3
        lib.sn_prof_enable(0)
        kernels_isolated = read_kernels()
    hp.stats.copy_(stats_timed)

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    pairs_emd = hp.stats[0:1].to(torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(pairs_emd, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    pairs_cd = 2.0 * b_local * N * N * args.steps * world
    pairs_total = pairs_cd + float(pairs_emd.item())
    maps_total = b_local * N_VIEWS * len(radius_list) * args.steps * world

    # the other scaling mode, same K / W (N > 1 only): one driver run per N gives both curves
    other = None
    if world > 1 and not args.no_other_scaling:
        mode2 = "strong" if args.scaling == "weak" else "weak"
        p2_, g2_ = make_inputs(dev, rank, world, mode2)
        e2, ps2, _ = timed_region(hp, p2_, g2_, args.steps, args.warmup, overlap, barrier, reduce_max=reduce_max)
        t2 = torch.tensor([e2], dtype=torch.float64, device=dev)
        pe2 = hp.stats[0:1].to(torch.float64)
        dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        dist.all_reduce(pe2, op=dist.ReduceOp.SUM)
        bl2 = p2_.size(0)
        other = {
            "scaling": mode2, "batch_per_gpu": bl2, "global_batch": bl2 * world,
            "value": (2.0 * bl2 * N * N * args.steps * world + float(pe2.item())) / float(t2.item()),
            "depthmaps_per_sec": bl2 * N_VIEWS * len(radius_list) * args.steps * world / float(t2.item()),
            "ms_per_step": float(t2.item()) / args.steps * 1e3,
            "step_ms_percentiles_rank0": percentiles(ps2),
        }
        del p2_, g2_

    # BASELINE config 3's radius_list exactly as written ([0.02, 0.05]: sub-pixel in the renderer's pixel units, so
    # the maps are near-empty) timed as the same whole step after the headline region (SURVEY 8d asks for both)
    literal = None
    lit_radii = [0.02, 0.05]
    if not args.no_literal_radii and radius_list != lit_radii:
        hp2 = HotPath(dev, lit_radii, args.per_view_render)
        lsteps = min(args.steps, 20)
        # >= 10 warm-up steps: a new HotPath, another template instance of the gather (2 radii) and fresh allocator
        # blocks made the first timed steps of this region 5-8x slower than the rest in round 3 (max 28 ms, median 3.4)
        e3, ps3, _ = timed_region(hp2, pred, gt, lsteps, max(args.warmup, 10), overlap, barrier, reduce_max=reduce_max)
        t3 = torch.tensor([e3], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t3, op=dist.ReduceOp.MAX)
        pc3 = percentiles(ps3)
        literal = {"radius_list": lit_radii, "steps": lsteps, "warmup": max(args.warmup, 10),
                   "ms_per_step": float(t3.item()) / lsteps * 1e3,
                   "depthmaps_per_sec": b_local * N_VIEWS * len(lit_radii) * lsteps * world / float(t3.item()),
                   # the same count over the MEDIAN step of rank 0 (HIP events at the step boundaries): what the
                   # region sustains once warm, next to the wall-clock figure above
                   "depthmaps_per_sec_median_step": b_local * N_VIEWS * len(lit_radii) * world / (pc3["median"] * 1e-3),
                   "step_ms_percentiles_rank0": pc3,
                   "note": "the same step (CD + EMD + expansion + render, fwd + bwd) with BASELINE.json's literal "
                           "radii; whole-job maps over the wall time of these steps"}
        del hp2

    # per-segment times on rank 0 (torch events on the current stream)
    seg = {}
    for i in range(1, len(timers)):
        name, ev = timers[i]
        if name == "start":
            continue
        seg[name] = seg.get(name, 0.0) + timers[i - 1][1].elapsed_time(ev)
    seg = {k: v / iso_steps for k, v in seg.items()}

    def counters_block(kname, launches, dur_s):
        """Executed-work figures of `kname` from the committed PMC counters of THIS build (None-valued if there are
        none), over the live launch time `dur_s` of `launches` launches."""
        blk = {}
        pmc = committed_counters(build_id, kname)
        if not pmc:
            blk["counters_from"] = (f"none: no profiles/*pmc*.json was taken on build {build_id} "
                                    "(tools/pmc_all.sh); counters of other code are not quoted")
            return blk
        fname, c = pmc
        m = lambda k: c.get(k, {}).get("mean")
        mops, valu, gui = m("SQ_INSTS_VALU_MFMA_MOPS_F32"), m("SQ_INSTS_VALU"), m("GRBM_GUI_ACTIVE")
        if valu is not None and launches and dur_s > 0:
            ex = (mops or 0.0) * 512.0 + valu * 64.0          # matrix-core flops + vector lane operations per launch
            ex_rate = ex * launches / dur_s / 1e12
            blk["executed_flops_per_launch"] = ex
            blk["executed_tflops"] = ex_rate
            blk["executed_frac"] = ex_rate / PEAK_F32_TFLOPS
        if gui:
            simd_cycles = gui / 8.0 * N_SIMD          # GRBM_GUI_ACTIVE is summed over the 8 XCDs
            if m("SQ_ACTIVE_INST_VALU") is not None:
                blk["valu_busy"] = m("SQ_ACTIVE_INST_VALU") * 4.0 / simd_cycles   # quad-cycles
            if m("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
                blk["mfma_busy"] = m("SQ_VALU_MFMA_BUSY_CYCLES") / simd_cycles
        if m("SQ_WAIT_ANY") is not None and m("SQ_WAVE_CYCLES"):
            blk["wait_frac"] = m("SQ_WAIT_ANY") / m("SQ_WAVE_CYCLES")      # wave cycles spent waiting on a counter
        if m("SQ_WAVES") is not None:
            blk["waves_per_launch"] = m("SQ_WAVES")
        if m("FETCH_SIZE") is not None and m("WRITE_SIZE") is not None:
            # KB per launch; x2 on FETCH_SIZE: the guide's gfx950 correction, calibrated for 16 B/lane streaming
            # reads; for 4 / 8-byte coherent accesses see traffic_calibration (tools/probe/traffic_probe.hip)
            blk["traffic"] = (2.0 * m("FETCH_SIZE") + m("WRITE_SIZE")) * 1024.0
            blk["traffic_note"] = ("2 x FETCH_SIZE + WRITE_SIZE; calibration for narrow coherent accesses: "
                                   "profiles/r03_*_traffic_calibration.json (streaming loads count half their bytes at "
                                   "any width / scope, L2 hits are not counted, agent-scope stores are written through, a "
                                   "scattered 4-byte access costs a 32-byte sector)")
            blk["traffic_fetch_bytes_raw"] = m("FETCH_SIZE") * 1024.0
            blk["traffic_write_bytes_raw"] = m("WRITE_SIZE") * 1024.0
        blk["counters_from"] = f"profiles/{fname} (build {build_id})"
        return blk

    roofline = None
    if not args.no_roofline:
        auc = kernels["emd_auction"]
        if auc["launches"]:
            pairs_rank = float(stats_timed[0].item())
            flops = FLOP_PER_PAIR["emd_auction"] * pairs_rank                # this rank, algorithmic
            # The kernel's duration = its own execution window, measured live inside the timed region with the
            # in-kernel clock; the HIP-event bracket on the launch stream (`bracket_avg_us`) additionally holds the
            # launch's wait for all compute units to be empty -- in the auction-first order the previous step's
            # renderer is still draining then -- which is the schedule's time, not the kernel's.
            exe = kernels.get("emd_auction_exec") or {}
            bracket_us = auc["avg_us"]
            if exe.get("launches") == auc["launches"] and exe.get("total_ms"):
                dur = exe["total_ms"] * 1e-3
                auc = dict(auc, avg_us=exe["avg_us"])
            else:
                dur = auc["total_ms"] * 1e-3
            achieved = flops / dur / 1e12
            cb = counters_block("emd_auction_kernel", auc["launches"], dur)
            roofline = {
                # what binds it in fact (PMC: wait_frac ~0.7, valu_busy < 0.5, mfma_busy ~0.03): the LATENCY of 50
                # dependent iterations x (bid -> team barrier -> award -> team barrier); `peak` is the fp32 vector =
                # fp32 matrix-core ceiling the executed work is priced against (`ceiling`)
                "kernel": "emd_auction_kernel", "bound": "latency", "ceiling": "mfma", "schema": 5,
                # `achieved` / `frac`: work the kernel really ISSUED (matrix-core flops + vector lane operations, PMC
                # of this build) over the kernel's OWN duration -- the isolated launch, what rocprofv3's kernel table
                # shows (filled in below; schema 5: rounds 3-4 published the live window here, which in the
                # auction-first order also holds the wait for the previous step's tail to leave the CUs -- that
                # figure is now under `live`)
                "achieved": cb.get("executed_tflops"),
                "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s", "frac": cb.get("executed_frac"), "frac_basis": "live",
                "traffic": cb.get("traffic"),
                # SURVEY 8(d)'s accounting: 14 flop x effective pairs / launch time.  NOT a roofline fraction: the
                # pruned search never evaluates most algorithmic pairs, so this rate can exceed the peak
                "algorithmic_tflops": achieved, "algorithmic_frac": achieved / PEAK_F32_TFLOPS,
                "algorithmic_bytes_per_launch": 32.0 * b_local * N,           # 24 B n in + 8 B n out
                "timing": "live, inside the timed region: `avg_launch_us` = the launch's own execution window (in-kernel "
                          "100 MHz clock: first working workgroup's start to the last one's end; what rocprofv3's kernel "
                          "table shows); `bracket_avg_us` = HIP events around the launch on its stream, which in the "
                          "auction-first order (>= 24 clouds per rank) also hold `queue_wait_avg_us`: the persistent grid "
                          "needs every compute unit empty and the previous step's renderer is still draining; "
                          "`isolated` = the same launches one stream at a time",
                "note": ("`achieved` / `frac` = executed work (SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 + SQ_INSTS_VALU x 64 "
                         "per launch, every vector instruction counted as 64 useful lanes) / live launch time / fp32 "
                         "peak (vector = f32 MFMA dense peak); null when no counters of this build are committed. "
                         "One plain vector instruction per SIMD every 4 cycles is 39 T lane-operations/s = 0.25 by "
                         "this measure: a kernel whose work is unpacked vector instructions reaches `frac` 0.25 with "
                         "every issue slot used (`valu_busy` = the share of those slots it does use). The kernel is "
                         "50 dependent iterations with two team barriers each: latency bound (wait_frac), not pipe or "
                         "HBM bound."),
                "launches": auc["launches"], "avg_launch_us": auc["avg_us"],
                "pairs_per_launch_avg": pairs_rank / auc["launches"],
                # the same launches inside the timed region (auction-first order: the grid waits for every CU to be empty)
                "live": {"exec_window_avg_us": auc["avg_us"], "bracket_avg_us": bracket_us,
                         "queue_wait_avg_us": (bracket_us - auc["avg_us"]) if (bracket_us and auc["avg_us"]) else None,
                         "achieved": cb.get("executed_tflops"), "frac": cb.get("executed_frac"),
                         "algorithmic_tflops": achieved, "algorithmic_frac": achieved / PEAK_F32_TFLOPS},
                "bracket_avg_us": bracket_us,   # (kept: the figure comparable with rounds 1-3)
            }
            roofline.update({k: v for k, v in cb.items() if k not in ("traffic",)})
            iso = kernels_isolated.get("emd_auction")
            if iso and iso["launches"]:
                ach = (FLOP_PER_PAIR["emd_auction"] * pairs_rank / args.steps * iso_steps
                       / (iso["total_ms"] * 1e-3) / 1e12)
                roofline["isolated"] = {"algorithmic_tflops": ach, "algorithmic_frac": ach / PEAK_F32_TFLOPS,
                                        "avg_launch_us": iso["avg_us"]}
                # the kernel's own duration and rates are the isolated ones (`achieved` / `frac` follow when counters exist)
                roofline.update({"avg_launch_us": iso["avg_us"], "algorithmic_tflops": ach,
                                 "algorithmic_frac": ach / PEAK_F32_TFLOPS, "frac_basis": "isolated"})
                if cb.get("executed_flops_per_launch"):
                    r = cb["executed_flops_per_launch"] / (iso["avg_us"] * 1e-6) / 1e12
                    roofline["isolated"]["achieved"] = r
                    roofline["isolated"]["frac"] = r / PEAK_F32_TFLOPS
                    # the kernel's figure = the isolated one
                    roofline.update({"achieved": r, "frac": r / PEAK_F32_TFLOPS, "frac_basis": "isolated",
                                     "avg_launch_us": iso["avg_us"], "algorithmic_tflops": ach,
                                     "algorithmic_frac": ach / PEAK_F32_TFLOPS})
            cf = kernels["chamfer_fwd"]
            if cf["launches"]:
                a = (FLOP_PER_PAIR["chamfer_fwd"] * 2.0 * b_local * N * N * cf["launches"]
                     / (cf["total_ms"] * 1e-3) / 1e12)
                blk = {"algorithmic_tflops": a, "algorithmic_frac": a / PEAK_F32_TFLOPS,
                       "avg_launch_us": cf["avg_us"],
                       "note": "sort + box-pruned search: 9 flop x ALL n*m pairs / launch time (algorithmic)"}
                # the counters are the SEARCH kernel's: priced over its own bracket inside the call (the call's bracket
                # also holds the sort and the prepare kernels)
                ns = kernels.get("nn_search") or {}
                if ns.get("launches"):
                    blk["search_kernel_avg_us"] = ns["avg_us"]
                    blk.update(counters_block("nn_search_kernel", ns["launches"], ns["total_ms"] * 1e-3))
                else:
                    blk.update(counters_block("nn_search_kernel", cf["launches"], cf["total_ms"] * 1e-3))
                if "executed_frac" in blk:
                    blk["frac"] = blk["executed_frac"]
                roofline["chamfer_fwd"] = blk
            # the renderer's dominant kernel: HBM-roofline accounting of SURVEY 8(d) (12 B N in per view + 8 B S^2
            # out per view and radius) next to what binds it in fact (vector-instruction issue: valu_busy)
            gk = kernels.get("p2i_max_splat")
            if gk and gk["launches"]:
                views = 1 if args.per_view_render else N_VIEWS
                byts = views * 12.0 * b_local * N + views * len(radius_list) * 8.0 * b_local * IMG * IMG
                rate = byts * gk["launches"] / (gk["total_ms"] * 1e-3) / 1e9
                blk = {"kernel": "p2i_gather_max_kernel", "bound": "hbm", "achieved": rate, "peak": PEAK_HBM_GBS,
                       "unit": "GB/s", "frac": rate / PEAK_HBM_GBS, "algorithmic_bytes_per_launch": byts,
                       "avg_launch_us": gk["avg_us"], "launches": gk["launches"],
                       "note": "algorithmic bytes / live launch time; the kernel is bound by vector-instruction issue "
                               "(valu_busy), not by HBM"}
                blk.update(counters_block("p2i_gather_max_kernel", gk["launches"], gk["total_ms"] * 1e-3))
                roofline["p2i_gather_max"] = blk

    if rank == 0:
        out = {
            "metric": "point-pairs/sec (CD+EMD) + depthmaps/sec, B=32 N=16384",
            "value": pairs_total / elapsed,
            "unit": "point-pairs/s",
            "n_gpus": dist.get_world_size() if world > 1 else 1,
            "rccl_ranks": rccl_ranks,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "depthmaps_per_sec": maps_total / elapsed,
            "depthmaps_per_sec_literal_radii": literal["depthmaps_per_sec"] if literal else None,
            "literal_radii": literal,
            "step_ms_percentiles_rank0": percentiles(per_step),
            "config": {
                "workload": (f"per rank: CD fwd+bwd + EMD(eps 0.005, 50 it) fwd+bwd + expansion(P=512, "
                             f"alpha 1.5) fwd+bwd on [{b_local},16384,3]; ComputeDepthMaps 8 views x radii "
                             f"{radius_list} px -> 256x256 fwd+bwd; scalar-loss all-reduce"),
                "batch_per_gpu": b_local, "global_batch": b_local * world, "points": N,
                "emd_iters": EMD_ITERS, "radius_list": radius_list,
                "image": IMG, "views": N_VIEWS,
                "streams": (1 if hp.one_stream(b_local) else
                            (3 if (hp.three_streams(b_local) or hp.auction_first(b_local)) else 2)) if overlap else 1,
                "order": ("one stream" if hp.one_stream(b_local) else
                          "auction first (high-priority stream), renderer | Chamfer + expansion beside and after it" if hp.auction_first(b_local)
                          else "expansion | Chamfer -> auction, renderer beside") if overlap else "one stream",
                # the stream orders timed on this batch during the untimed warm-up (HotPath.choose_schedule): ms per step
                "schedule_table_ms": (hp.schedule.get(b_local) or (None, None))[1],
                "library_build": build_id,
                "render": "view by view" if args.per_view_render else "8 views in one pass (forward_views)",
            },
            "other_scaling": other,
            "pairs_per_step": pairs_total / args.steps / world,
            "maps_per_step": maps_total / args.steps / world,
            "segments_ms_rank0": seg,
            "sequential_ms_per_step_rank0": isolated_ms,
            "kernels_rank0": kernels,
            "kernels_isolated_rank0": kernels_isolated,
            "losses": [float(x) for x in losses.tolist()],
            "roofline": roofline,
        }
        if world == 1 and not args.no_other_ops:
            oo = other_ops(dev, pred, hp.last_mean_mst)
            out["other_ops_ms_rank0"] = oo
            if roofline is not None:
                # the sampler (outside the headline step; 2 calls per generator forward): SURVEY 8(d)'s 12 flop per
                # point and round, 16383 dependent picks per cloud.  Since round 6 every cloud of this size is sampled by
                # a TEAM of workgroups (mds_dense_team_kernel: 8 per cloud at 32 clouds) taking several exact picks per
                # exchange; counters per workload: tools/pmc_all.sh splits the probe's three team dispatches
                fl = 12.0 * B * (N - 1) * 19384
                blk = {"kernel": "mds_dense_team_kernel", "bound": "latency", "peak": PEAK_F32_TFLOPS,
                       "unit": "TFLOP/s", "algorithmic_flops_per_launch": fl,
                       "note": "a team of workgroups per cloud, several exact picks per exchange: a chain of dependent "
                               "exchanges (update -> arg-min -> poll -> replay); the updates sit at the VALU floor, the "
                               "culled update skips most of the algorithmic point-rounds in the surface regime"}
                for tag, key in (("surface", "mds_19384_to_16384_surface"), ("dense", "mds_19384_to_16384")):
                    sub = {"ms": oo[key], "algorithmic_frac": fl / (oo[key] * 1e-3) / 1e12 / PEAK_F32_TFLOPS}
                    sub.update(counters_block(f"mds_dense_team_kernel#{tag}_b32", 1, oo[key] * 1e-3))
                    sub["frac"] = sub.get("executed_frac")
                    blk[tag] = sub
                blk["frac"] = blk["dense"].get("frac")
                blk["counters_from"] = blk["dense"].get("counters_from")
                roofline["mds_team"] = blk
        if world == 1 and not args.no_emd_regimes:
            out["emd_regimes_rank0"] = emd_regimes(dev)
        if world == 1 and not args.no_network_steps:
            out["network_steps_rank0"] = network_steps(dev)
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline()
            # the north star's combined ratio: one step's forward work, CPU over the GPU's whole step
            cb["gpu_step_ms"] = elapsed / args.steps * 1e3
            cb["combined_speedup_vs_all_cores"] = cb["step_forward_equivalent_s"] / (elapsed / args.steps)
            cb["combined_speedup_vs_one_thread"] = (cb["single_thread"]["step_forward_equivalent_s"]
                                                    / (elapsed / args.steps))
            out["cpu_baseline"] = cb
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
