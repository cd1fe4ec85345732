"""BASELINE configs 4-5 on synthetic ShapeNet-shaped batches: the reconstruction step and the GAN step with the
reference's networks (sparenet_amd/networks.py: EdgeConv encoder, 32-primitive style decoder, refine x2,
PatchDiscriminator; bf16 autocast around the fp32 HIP ops), plus the op-level step on surrogates.

    python tools/step_harness.py                              one GPU, per-rank shares of the global batches
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/step_harness.py
                                                              DistributedDataParallel over RCCL (bucketed
                                                              gradient all-reduce overlapped with backward)
Sizes: 16384 output points, 3000 input points, 32 primitives.  Config 4: global batch 32 (and 24), config 5:
global batch 64; a rank processes global / world clouds (world = 8 unless launched otherwise: on one GPU the
per-rank share of an 8-GPU job is timed, --full-batch runs the whole global batch on this GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from sparenet_amd.harness import Completion, GanStep, SurrogateGenerator
from sparenet_amd import networks as nw

world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
if world > 1:
    dist.init_process_group("nccl", rank=rank, world_size=world)
share_of = world if world > 1 else (1 if "--full-batch" in sys.argv else 8)
N, M = 16384, 3000
say = (lambda *a: print(*a, flush=True)) if rank == 0 else (lambda *a: None)


def batch(b, seed):
    g = torch.Generator().manual_seed(seed + rank)
    v = torch.randn(b, N, 3, generator=g)
    gt = 0.5 * v / v.norm(dim=2, keepdim=True)            # surface-like ground truth
    key = (torch.atan2(gt[..., 1], gt[..., 0]) * 4).floor() * 100 + (gt[..., 2] * 8).floor()
    gt = torch.gather(gt, 1, key.argsort(dim=1).unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    partial = (gt[:, torch.randperm(N, generator=g)[:M]] + 1e-3 * torch.randn(b, M, 3, generator=g)).contiguous()
    return partial.to(dev), gt.to(dev)


def clock(fn, reps=3):
    fn(); fn(); torch.cuda.synchronize()
    if world > 1: dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


if "--ops-only" in sys.argv or world == 1:
    part_d, gt_d = batch(32, 0)
    init = gt_d.cpu() + 0.01 * torch.randn(32, N, 3)
    for metric in ("chamfer", "emd"):
        gen = SurrogateGenerator(32, N, 32, init=init).to(dev)
        comp = Completion(metric).to(dev)
        opt = torch.optim.SGD(gen.parameters(), lr=0.1)
        def step():
            loss, *_ = comp(gen, part_d, gt_d)
            opt.zero_grad(); loss.backward(); opt.step()
            return loss
        ms, _ = clock(step)
        say(f"op-level step on surrogates, B=32, metric={metric}: {ms:.1f} ms")

for name, global_b, metric in (("config 4 (reconstruction)", 32, "emd"), ("config 4 (reconstruction)", 24, "emd"),
                               ("config 4 (reconstruction)", 32, "chamfer")):
    b = max(1, global_b // share_of)
    torch.manual_seed(0)
    gen = nw.Generator(num_points=N, n_primitives=32).to(dev)
    model = nw.data_parallel(gen, dev)
    comp = Completion(metric, overlap=False).to(dev)
    opt = torch.optim.Adam(gen.parameters(), lr=1e-4)
    part_d, gt_d = batch(b, 1)
    def rstep():
        loss, *_ = comp(model, part_d, gt_d)
        opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
        return loss
    torch.cuda.reset_peak_memory_stats()
    ms, loss = clock(rstep)
    say(f"{name}: global batch {global_b}, {b} clouds on this rank (1/{share_of}), metric {metric}, "
        f"{sum(p.numel() for p in gen.parameters()) / 1e6:.1f} M parameters, world {world}: {ms:.1f} ms per step = "
        f"{1e3 / ms:.2f} steps/s, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB, loss {float(loss):.4f}")
    del gen, model, opt

b = max(1, 64 // share_of)
torch.manual_seed(0)
gen = nw.Generator(num_points=N, n_primitives=32).to(dev)
disc = nw.PatchDiscriminator((16, 256, 256)).to(dev)
g_model, d_model = nw.data_parallel(gen, dev), nw.data_parallel(disc, dev, bucket_cap_mb=16)
gan = GanStep(g_model, d_model, Completion("emd", overlap=False).to(dev), torch.optim.Adam(gen.parameters(), lr=1e-4),
              torch.optim.Adam(disc.parameters(), lr=1e-4))
part_d, gt_d = batch(b, 2)
torch.cuda.reset_peak_memory_stats()
ms, out = clock(lambda: gan(part_d, gt_d))
say(f"config 5 (GAN step): global batch 64, {b} clouds on this rank (1/{share_of}), world {world}: {ms:.1f} ms per step = "
    f"{1e3 / ms:.2f} steps/s, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB "
    f"(errG {float(out['errG']):.4f} errD {float(out['errD_real'] + out['errD_fake']):.4f})")
if world > 1:
    dist.destroy_process_group()
