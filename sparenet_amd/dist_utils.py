"""Data-parallel helpers: one process per GPU, batches sharded by whole clouds.

Replaces the reference's nn.DataParallel scatter/gather (runners/base_runner.py:100-104,
runners/sparenet_runner.py:32-34) for the loss/render path: every op is independent per
batch element, so ranks own contiguous slices of dim 0 and the only exchange is one
all-reduce (RCCL on MI355X, backend "nccl"; gloo in the CPU tests) of the small vector of
scalar losses.  DataParallel gathers per-replica scalar means and the runner averages them
(sparenet_runner.py:86 `.mean()`): mean of shard means == all_reduce(SUM) / world for equal shards.
"""
import torch
import torch.distributed as dist


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def shard_bounds(batch: int, rank: int, world: int):
    """Contiguous slice [lo, hi) of dim 0 owned by `rank` (remainder goes to the low ranks)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard(t: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    lo, hi = shard_bounds(t.size(0), rank, world)
    return t[lo:hi]


def reduce_mean_of_means(local_means: torch.Tensor) -> torch.Tensor:
    """All-reduce a small vector of per-rank scalar means into the DataParallel-style
    mean of replica means.  No-op without an initialised process group."""
    if not is_distributed():
        return local_means
    out = local_means.clone()
    dist.all_reduce(out, op=dist.ReduceOp.SUM)
    return out / dist.get_world_size()
