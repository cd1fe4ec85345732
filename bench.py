#!/usr/bin/env python3
"""bench.py -- SpareNet loss/render hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one synthetic batch PER RANK (weak scaling,
B=32 clouds of N=16384 points each rank -- BASELINE.json configs[1] + configs[2]):
    Chamfer distance            fwd + bwd   [32,16384,3] <-> [32,16384,3]
    EMD (auction)               fwd + bwd   eps 0.005, 50 iterations
    expansion penalty           fwd + bwd   primitive_size 512, alpha 1.5
    ComputeDepthMaps render     fwd + bwd   8 views x radius_list, 256 x 256
    scalar losses               all-reduce (RCCL) when N > 1
The four parts are independent given the predicted cloud; by default the renderer runs on a second
HIP stream next to the distance losses (config.streams = 2; --no-overlap times the one-stream step).
After the timed region the same steps run once more one stream at a time, untimed for `value`, to
report per-part times and the kernels' uncontended durations (roofline.isolated).
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line:
  value             point pairs per second, whole job (CD pairs 2*B*N*M + EMD effective pairs
                    sum_it sum_b unassigned*n, counted on the device) / wall time of the steps
  depthmaps_per_sec single-radius 256x256 maps per second over the same wall time
  roofline          the dominant kernel (emd_bid): algorithmic flops / its measured launch time
  cpu_baseline      the CPU oracle (port of the reference algorithm) on a bounded sample
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

from sparenet_amd.dist_utils import reduce_mean_of_means  # noqa: E402

B, N = 32, 16384
EMD_EPS, EMD_ITERS = 0.005, 50
PRIM, ALPHA = 512, 1.5
IMG = 256
N_VIEWS = 8
FLOP_PER_PAIR = {"chamfer_fwd": 9.0, "emd_bid": 14.0}
PEAK_F32_TFLOPS = 157.3                                   # MI355X_MICROARCH.md (vector == f32 MFMA peak)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--radius-list", type=str, default="5,7,10",
                    help="p2i radii in pixels (reference default, configs/base_config.py:56-60); "
                         "BASELINE.json's literal 0.02,0.05 is near-empty in pixel units")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true",
                    help="time the sequential one-stream step instead of the two-stream one")
    ap.add_argument("--no-other-ops", action="store_true",
                    help="skip the untimed-for-the-headline MDS/gather/gridding/cubic measurements")
    return ap.parse_args()


def make_inputs(dev, rank):
    g = torch.Generator().manual_seed(1234 + rank)
    pred = torch.rand(B, N, 3, generator=g)
    gt = torch.rand(B, N, 3, generator=g)
    return pred.to(dev), gt.to(dev)


class HotPath:
    """One training-step worth of loss/render ops, composed like the reference runners
    (runners/sparenet_runner.py:83-108, runners/sparenet_gan_runner.py:212-225)."""

    def __init__(self, dev, radius_list):
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
        self.stats = torch.zeros(2, dtype=torch.int64, device=dev)
        self.ev = {}
        self.last_mean_mst = None
        self.side = None

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
        p4 = (pred.detach() - 0.5).requires_grad_(True)
        acc = None
        for v in range(N_VIEWS):
            maps = self.render(p4, view_id=v, radius_list=self.radius_list)
            s = maps.mean()
            acc = s if acc is None else acc + s
        acc.backward()
        return acc

    def step_overlapped(self, pred, gt):
        """The same step with the renderer on a second HIP stream: the four parts are independent
        given the predicted cloud, and the renderer's kernels fill the CUs the late (few-bidder) auction
        iterations leave idle.  Same kernels, same results; per-kernel durations stretch under contention."""
        main = torch.cuda.current_stream()
        if self.side is None:
            self.side = torch.cuda.Stream()
        self.side.wait_stream(main)
        with torch.cuda.stream(self.side):
            acc = self._render_all(pred)
        loss_cd, loss_emd, loss_exp = self._distance_losses(pred, gt)
        main.wait_stream(self.side)
        losses = torch.stack([loss_cd.detach(), loss_emd.detach(), loss_exp.detach(), acc.detach()])
        return reduce_mean_of_means(losses)   # RCCL all-reduce over xGMI when N > 1

    def _distance_losses(self, pred, gt, mark=lambda name: None):
        # The expansion penalty first: one wave per 512-point patch = one lone wave per SIMD for 0.45 ms,
        # latency bound -- next to the renderer's stream it costs nothing, at the end of the chain it runs alone.
        p3 = pred.detach().requires_grad_(True)
        pen, _, mml = self.expansion(p3, PRIM, ALPHA)
        loss_exp = pen.mean()
        loss_exp.backward()
        self.last_mean_mst = mml.detach()
        mark("expansion")
        p = pred.detach().requires_grad_(True)
        g = gt.detach().requires_grad_(True)
        d1, d2 = self.cd(p, g)
        loss_cd = d1.mean() + d2.mean()
        loss_cd.backward()
        mark("cd")
        p2 = pred.detach().requires_grad_(True)
        dist, _ = self._emd(p2, gt)
        loss_emd = torch.sqrt(dist).mean(1).mean()
        loss_emd.backward()
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


def cpu_baseline():
    """Time the CPU oracle (OpenMP, all host cores) on a bounded sample of the same workload."""
    import numpy as np
    import oracle

    cores = os.cpu_count() or 1
    g = torch.Generator().manual_seed(1234)
    pred = torch.rand(B, N, 3, generator=g).numpy()
    gt = torch.rand(B, N, 3, generator=g).numpy()
    nb_cd, nb_emd, nb_p2i = 32, 16, 32   # sized for ~10-30 s of CPU work on a 2-socket host
    t0 = time.perf_counter()
    oracle.chamfer_forward(pred[:nb_cd], gt[:nb_cd], mt=True)
    t_cd = time.perf_counter() - t0
    pairs_cd = 2.0 * nb_cd * N * N
    t0 = time.perf_counter()
    _, _, aux = oracle.emd_forward(pred[:nb_emd], gt[:nb_emd], EMD_EPS, EMD_ITERS, mt=True,
                                   return_aux=True)
    t_emd = time.perf_counter() - t0
    pairs_emd = float(aux["pairs_eff"])
    # render sample: one view, one radius, 4 clouds, through the oracle p2i (single thread)
    from sparenet_amd.utils.p2i_utils import ComputeDepthMaps
    cdm = ComputeDepthMaps("orthorgonal", 1.0, IMG)
    data = torch.from_numpy(pred[:nb_p2i]) - 0.5
    ij, feat = cdm.project(data, 0)
    px = ((ij + 1) / 2 * (IMG - 1)).numpy()
    bi = np.repeat(np.arange(nb_p2i, dtype=np.int32), N)
    t0 = time.perf_counter()
    oracle.p2i_max_forward(px, feat.numpy(), bi, np.zeros((nb_p2i, 1, IMG, IMG), np.float32), 5.0)
    t_p2i = time.perf_counter() - t0
    return {
        "value": (pairs_cd + pairs_emd) / (t_cd + t_emd),
        "unit": "point-pairs/s",
        "cores": cores,
        "kind": "port",
        "sample": (f"oracle (C, OpenMP x{cores} threads): Chamfer fwd on {nb_cd} of 32 clouds "
                   f"({pairs_cd:.3g} pairs, {t_cd:.2f} s) + EMD fwd eps {EMD_EPS} iters {EMD_ITERS} on "
                   f"{nb_emd} cloud ({pairs_emd:.3g} effective pairs, {t_emd:.2f} s); "
                   f"p2i max fwd R=5 on {nb_p2i} clouds x 1 view, 1 thread: {t_p2i:.2f} s"),
        "depthmaps_per_sec_1thread": nb_p2i / t_p2i,
        "chamfer_pairs_per_sec": pairs_cd / t_cd,
        "emd_pairs_per_sec": pairs_emd / t_emd,
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
    v = torch.randn(B, N, 3, generator=g)
    v = 0.5 * v / v.norm(dim=2, keepdim=True)
    key = (torch.atan2(v[..., 1], v[..., 0]) * 4).floor() * 100 + (v[..., 2] * 8).floor()
    surf = torch.gather(v, 1, key.argsort(dim=1).unsqueeze(-1).expand(-1, -1, 3)).contiguous().to(dev)
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


def main():
    args = parse()
    radius_list = [float(r) for r in args.radius_list.split(",")]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    torch.cuda.set_device(local)   # before the process group: RCCL binds to the current device
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    n_gpus = world

    pred, gt = make_inputs(dev, rank)
    hp = HotPath(dev, radius_list)
    lib = hp.lib.lib()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    overlap = not args.no_overlap
    run_step = hp.step_overlapped if overlap else hp.step
    for _ in range(args.warmup):
        run_step(pred, gt)
    barrier()
    hp.stats.zero_()
    timers = []
    if not args.no_roofline:
        lib.sn_prof_reset()
        lib.sn_prof_enable(1)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = run_step(pred, gt, timers) if not overlap else run_step(pred, gt)
    barrier()
    elapsed = time.perf_counter() - t0
    lib.sn_prof_enable(0)
    stats_timed = hp.stats.clone()

    def read_kernels():
        ks = {}
        for kname in ("chamfer_fwd", "emd_bid", "emd_auction", "expansion_fwd", "p2i_max_splat"):
            ms = ctypes.c_double(0.0)
            cnt = lib.sn_prof_read(kname.encode(), ctypes.byref(ms))
            ks[kname] = {"launches": int(cnt), "total_ms": ms.value,
                         "avg_us": (ms.value / cnt * 1e3) if cnt else None}
        return ks

    kernels = read_kernels() if not args.no_roofline else {}
    # outside the timed region: the same steps one stream at a time, for the per-part times and for
    # the kernels' uncontended durations
    kernels_isolated, isolated_ms = {}, None
    if overlap and not args.no_roofline:
        lib.sn_prof_reset()
        lib.sn_prof_enable(1)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            hp.step(pred, gt, timers)
        barrier()
        isolated_ms = (time.perf_counter() - t1) / args.steps * 1e3
        lib.sn_prof_enable(0)
        kernels_isolated = read_kernels()
    hp.stats.copy_(stats_timed)

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    pairs_emd = hp.stats[0:1].to(torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(pairs_emd, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    pairs_cd = 2.0 * B * N * N * args.steps * world
    pairs_total = pairs_cd + float(pairs_emd.item())
    maps_total = B * N_VIEWS * len(radius_list) * args.steps * world

    # per-segment times on rank 0 (torch events on the current stream)
    seg = {}
    for i in range(1, len(timers)):
        name, ev = timers[i]
        if name == "start":
            continue
        seg[name] = seg.get(name, 0.0) + timers[i - 1][1].elapsed_time(ev)
    seg = {k: v / args.steps for k, v in seg.items()}

    roofline = None
    if not args.no_roofline:
        bid = kernels["emd_auction"] if kernels["emd_auction"]["launches"] else kernels["emd_bid"]
        if bid["launches"]:
            flops = FLOP_PER_PAIR["emd_bid"] * float(hp.stats[0].item())     # this rank
            achieved = flops / (bid["total_ms"] * 1e-3) / 1e12
            traffic = None   # PMC passes cannot run inside the bench: committed measurement
            tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_emd_bid_traffic.json")))
            if tfiles:   # the newest committed PMC measurement (tools/traffic_emd.sh)
                traffic = json.load(open(tfiles[-1])).get("bytes_per_launch_corrected")
            roofline = {
                "kernel": "emd_bid_kernel", "bound": "mfma", "achieved": achieved,
                "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F32_TFLOPS,
                "traffic": traffic,
                "note": ("pairwise search, 14 flop/pair x effective pairs (SURVEY 8d) over the bid "
                         "launches; peak = f32 vector = f32 MFMA dense peak 157.3 TFLOP/s. The "
                         "kernel filters pairs on the fp32 matrix cores and skips superblocks of "
                         "targets by bounding box, so part of the algorithmic pairs is never "
                         "evaluated"),
                "launches": bid["launches"], "avg_launch_us": bid["avg_us"],
                "pairs_per_launch_avg": float(hp.stats[0].item()) / bid["launches"],
            }
            iso = kernels_isolated.get("emd_auction") if kernels_isolated.get("emd_auction", {}).get("launches") else kernels_isolated.get("emd_bid")
            if iso and iso["launches"]:
                ach = flops / (iso["total_ms"] * 1e-3) / 1e12
                roofline["isolated"] = {
                    "achieved": ach, "frac": ach / PEAK_F32_TFLOPS, "avg_launch_us": iso["avg_us"],
                    "note": ("the same steps run one stream at a time after the timed region; in the "
                             "timed region the renderer runs on a second stream next to the auction "
                             "and every kernel's duration includes that contention")}
            cf = kernels["chamfer_fwd"]
            if cf["launches"]:
                roofline["chamfer_fwd"] = {
                    "achieved": FLOP_PER_PAIR["chamfer_fwd"] * 2.0 * B * N * N * cf["launches"]
                    / (cf["total_ms"] * 1e-3) / 1e12,
                    "avg_launch_us": cf["avg_us"]}
                roofline["chamfer_fwd"]["frac"] = roofline["chamfer_fwd"]["achieved"] / PEAK_F32_TFLOPS
                roofline["chamfer_fwd"]["note"] = (
                    "sort + box-pruned search: 9 flop x ALL n*m pairs / launch time; most pairs are "
                    "never evaluated, so this algorithmic rate may exceed the all-pairs roofline")

    if rank == 0:
        out = {
            "metric": "point-pairs/sec (CD+EMD) + depthmaps/sec, B=32 N=16384",
            "value": pairs_total / elapsed,
            "unit": "point-pairs/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "depthmaps_per_sec": maps_total / elapsed,
            "config": {
                "workload": ("per rank: CD fwd+bwd + EMD(eps 0.005, 50 it) fwd+bwd + expansion(P=512, "
                             "alpha 1.5) fwd+bwd on [32,16384,3]; ComputeDepthMaps 8 views x radii "
                             f"{radius_list} px -> 256x256 fwd+bwd; scalar-loss all-reduce"),
                "batch_per_gpu": B, "points": N, "emd_iters": EMD_ITERS, "radius_list": radius_list,
                "image": IMG, "views": N_VIEWS, "streams": 2 if overlap else 1,
            },
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
            out["other_ops_ms_rank0"] = other_ops(dev, pred, hp.last_mean_mst)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
