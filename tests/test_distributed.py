"""N > 1 path on CPU: world size 2 over gloo (127.0.0.1).  Each rank owns a contiguous
slice of whole clouds, computes its scalar loss means (here through the CPU oracle, the ops
themselves need no communication) and the all-reduce reproduces the DataParallel-style mean
of replica means == the global mean for equal shards."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sparenet_amd.dist_utils import reduce_mean_of_means, shard, shard_bounds


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    import oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(7)
    pred = torch.rand(4, 256, 3, generator=g)
    gt = torch.rand(4, 256, 3, generator=g)
    p, q = shard(pred, rank, world), shard(gt, rank, world)
    d1, d2, _, _ = oracle.chamfer_forward(p.numpy(), q.numpy())
    local = torch.tensor([d1.mean() + d2.mean(), float(p.shape[0])], dtype=torch.float64)
    red = reduce_mean_of_means(local)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), red.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_batch():
    for batch, world in ((32, 8), (24, 8), (5, 2), (3, 4)):
        spans = [shard_bounds(batch, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == batch
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))


def test_world2_gloo_loss_allreduce(tmp_path):
    import oracle

    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "r0.npy")
    r1 = np.load(tmp_path / "r1.npy")
    assert np.array_equal(r0, r1)
    g = torch.Generator().manual_seed(7)
    pred = torch.rand(4, 256, 3, generator=g)
    gt = torch.rand(4, 256, 3, generator=g)
    d1, d2, _, _ = oracle.chamfer_forward(pred.numpy(), gt.numpy())
    np.testing.assert_allclose(r0[0], d1.mean() + d2.mean(), rtol=1e-6)
    assert r0[1] == 2.0  # each rank owned 2 of the 4 clouds


@pytest.mark.gpu
def test_sharded_step_equals_unsharded_on_the_product_path(dev):
    """SURVEY 8(e) on the HIP ops themselves (one process standing in for the ranks, one after the other):
    the losses of shard(pred, r, world) averaged over the ranks (what the all-reduce produces) equal the
    unsharded batch's -- every op is independent per cloud.  ComputeDepthMaps is the documented exception:
    its depth feature is normalised by the z range of the LOCAL tensor (utils/p2i_utils.py:226), so a
    shard's maps differ from the corresponding slice of the whole batch's maps, exactly as under the
    reference's DataParallel."""
    from sparenet_amd.cuda.chamfer_distance import ChamferDistance
    from sparenet_amd.cuda.emd.emd_module import emdModule
    from sparenet_amd.cuda.expansion_penalty.expansion_penalty_module import expansionPenaltyModule
    from sparenet_amd.utils.p2i_utils import ComputeDepthMaps

    world = 4
    g = torch.Generator().manual_seed(1234)
    pred = torch.rand(8, 2048, 3, generator=g).to(dev)
    gt = torch.rand(8, 2048, 3, generator=g).to(dev)
    render = ComputeDepthMaps("orthorgonal", 1.0, 64).to(dev)

    def losses(p, q):
        d1, d2 = ChamferDistance()(p, q)
        dist, _ = emdModule()(p, q, eps=0.005, iters=20)
        pen, _, _ = expansionPenaltyModule()(p, 256, 1.5)
        return torch.stack([d1.mean() + d2.mean(), torch.sqrt(dist).mean(1).mean(), pen.mean()]).double()

    whole = losses(pred, gt)
    parts = torch.stack([losses(shard(pred, r, world).contiguous(), shard(gt, r, world).contiguous())
                         for r in range(world)])
    np.testing.assert_allclose(parts.mean(0).cpu().numpy(), whole.cpu().numpy(), rtol=2e-6)
    # per-cloud results themselves are identical, not just their means
    d_whole, a_whole = emdModule()(pred, gt, eps=0.005, iters=20)
    lo, hi = shard_bounds(8, 2, world)
    d_part, a_part = emdModule()(pred[lo:hi].contiguous(), gt[lo:hi].contiguous(), eps=0.005, iters=20)
    assert torch.equal(a_whole[lo:hi], a_part) and torch.equal(d_whole[lo:hi], d_part)
    maps_whole = render(pred - 0.5, view_id=1, radius_list=[5.0])
    maps_part = render((pred - 0.5)[lo:hi].contiguous(), view_id=1, radius_list=[5.0])
    assert not torch.equal(maps_whole[lo:hi], maps_part)          # local z-range normalisation
    assert (maps_whole[lo:hi] > 0).eq(maps_part > 0).all()        # same footprints, different depth scale


def test_bench_launches_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` (N > 1, no launcher around it) re-executes itself under torch.distributed.run with
    one process per GPU on 127.0.0.1 (the reference spreads a batch over its GPUs from one command too,
    runners/base_runner.py:100-104), and refuses to run fewer ranks than asked."""
    import subprocess
    import sys

    import bench

    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2"])
    args = bench.parse()
    assert args.gpus == 8 and args.scaling == "strong"          # auto = the strong split of SURVEY 8(e)
    monkeypatch.delenv("BENCH_DEBUG_SHARED_GPU", raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit) as e:
        bench.launch_ranks(args)
    assert "only 1 GPU(s) visible" in str(e.value) and "cmd" not in seen
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    with pytest.raises(SystemExit) as e:
        bench.launch_ranks(args)
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"     # dmabuf IPC: RCCL needs it on this driver


def test_bench_step_schedule_by_batch():
    """bench.HotPath's stream order: the table of rounds 3-4 before anything was measured (third stream for the expansion
    penalty at <= 16 clouds per rank, auction first from 24 on), the MEASURED choice afterwards (choose_schedule keeps
    the fastest of one stream / two / three / auction first per batch size -- never slower than one stream), and the
    environment overrides for A/B runs."""
    import bench

    hp = bench.HotPath.__new__(bench.HotPath)
    hp.three_streams_env, hp.order, hp.schedule = None, "auto", {}
    assert [hp.three_streams(b) for b in (32, 16, 8, 4)] == [False, True, True, True]
    assert [hp.auction_first(b) for b in (32, 24, 16, 4)] == [True, True, False, False]
    assert not any(hp.one_stream(b) for b in (32, 16, 8, 4))
    hp.schedule = {16: ("one_stream", {}), 32: ("auction_first", {}), 8: ("chain_3", {}), 4: ("chain_2", {})}
    assert [hp.one_stream(b) for b in (32, 16, 8, 4)] == [False, True, False, False]
    assert [hp.auction_first(b) for b in (32, 16, 8, 4)] == [True, False, False, False]
    assert [hp.three_streams(b) for b in (32, 16, 8, 4)] == [False, False, True, False]
    hp.three_streams_env = "0"
    assert not hp.three_streams(8)
    hp.order = "chain"
    assert not hp.auction_first(32) and not hp.one_stream(16)
    hp.order = "one_stream"
    assert hp.one_stream(32) and not hp.auction_first(32)
