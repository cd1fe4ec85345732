"""Do the kernels stay inside their INPUT buffers?  Every input tensor of an op is placed at the very END of its own
device allocation (a multiple of 2 MiB, allocator caching off), so a read or write past its last byte leaves the
mapping and the process dies with a memory access fault instead of quietly touching a neighbour -- which is what an
out-of-bounds access does inside the caching allocator's large segments.  One subprocess per op:
    PYTORCH_NO_CUDA_MEMORY_CACHING=1 python tools/oob_probe.py            (driver: runs every op, prints a verdict)
    ... tools/oob_probe.py <op>                                          (one op, in this process)"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OPS = ["chamfer", "chamfer_small", "emd", "expansion", "mds", "mds_dense", "gather", "p2i", "render", "gridding", "cubic", "knn"]

if len(sys.argv) == 1:
    env = dict(os.environ, PYTORCH_NO_CUDA_MEMORY_CACHING="1", PYTORCH_NO_HIP_MEMORY_CACHING="1")
    for op in OPS:
        r = subprocess.run([sys.executable, __file__, op], env=env, capture_output=True, text=True, timeout=300)
        ok = "DONE" in r.stdout
        print(f"{op:14s} {'ok' if ok else 'FAULT / error: ' + (r.stderr.strip().splitlines() or ['?'])[-1][:160]}", flush=True)
    sys.exit(0)

import torch
dev = torch.device("cuda:0")
MB2 = 2 << 20


def tail(t):
    """A copy of t whose last byte is the last byte of its own allocation."""
    nbytes = t.numel() * t.element_size()
    big = torch.empty((nbytes + MB2 - 1) // MB2 * MB2, dtype=torch.uint8, device=dev)
    keep.append(big)
    v = big[big.numel() - nbytes:].view(t.dtype).view(t.shape)
    v.copy_(t)
    return v


keep = []
op = sys.argv[1]
g = torch.Generator().manual_seed(3)
if op in ("chamfer", "chamfer_small"):
    from sparenet_amd.cuda.chamfer_distance import ChamferDistance
    n, m = (16384, 16384) if op == "chamfer" else (1777, 1300)
    x = tail(torch.rand(4, n, 3, generator=g).to(dev)).requires_grad_(True)
    y = tail(torch.rand(4, m, 3, generator=g).to(dev)).requires_grad_(True)
    d1, d2 = ChamferDistance()(x, y)
    (d1.mean() + d2.mean()).backward()
elif op == "emd":
    from sparenet_amd.cuda.emd.emd_module import emdModule
    x = tail(torch.rand(4, 16384, 3, generator=g).to(dev)).requires_grad_(True)
    y = tail(torch.rand(4, 16384, 3, generator=g).to(dev))
    d, _ = emdModule()(x, y, 0.005, 50)
    torch.sqrt(d).mean().backward()
elif op == "expansion":
    from sparenet_amd.cuda.expansion_penalty.expansion_penalty_module import expansionPenaltyModule
    x = tail(torch.rand(4, 16384, 3, generator=g).to(dev)).requires_grad_(True)
    pen, _, _ = expansionPenaltyModule()(x, 512, 1.5)
    pen.mean().backward()
elif op in ("mds", "mds_dense"):
    from sparenet_amd.cuda.MDS.MDS_module import minimum_density_sample
    x = tail(torch.rand(4, 19384, 3, generator=g).to(dev))
    mml = tail(torch.full((4,), 0.0085 if op == "mds" else 0.09, device=dev))
    minimum_density_sample(x, 4000, mml)
elif op == "gather":
    from sparenet_amd.cuda.MDS.MDS_module import gather_operation
    f = tail(torch.rand(4, 4, 19384, generator=g).to(dev)).requires_grad_(True)
    idx = tail(torch.randint(0, 19384, (4, 16384), generator=g).to(torch.int32).to(dev))
    gather_operation(f, idx).sum().backward()
elif op == "p2i":
    from sparenet_amd.cuda.p2i_op import p2i
    pts = tail((torch.rand(4 * 16384, 2, generator=g) * 2 - 1).to(dev)).requires_grad_(True)
    ft = tail(torch.rand(4 * 16384, 1, generator=g).to(dev)).requires_grad_(True)
    bi = tail(torch.arange(4, dtype=torch.int32).repeat_interleave(16384).to(dev))
    bg = tail(torch.zeros(4, 1, 256, 256, device=dev)).requires_grad_(True)
    for red in ("max", "sum"):
        p2i(pts, ft, bi, bg, 5.0, "cos", red).sum().backward()
elif op == "render":
    from sparenet_amd.utils.p2i_utils import ComputeDepthMaps
    cdm = ComputeDepthMaps("orthorgonal", 1.0, 256).to(dev)
    p = tail((torch.rand(4, 16384, 3, generator=g) - 0.5).to(dev)).requires_grad_(True)
    cdm.forward_views(p, range(8), [5.0, 7.0, 10.0]).mean().backward()
    cdm(p, view_id=3, radius_list=[10.0]).mean().backward()
elif op == "gridding":
    from sparenet_amd.cuda.gridding import Gridding, GriddingReverse
    p = tail(((torch.rand(4, 2048, 3, generator=g) - 0.5) * 1.9).to(dev)).requires_grad_(True)
    Gridding(64)(p).sum().backward()
    v = tail(torch.rand(2, 32, 32, 32, generator=g).to(dev)).requires_grad_(True)
    GriddingReverse(32)(v).sum().backward()
elif op == "cubic":
    from sparenet_amd.cuda.cubic_feature_sampling import CubicFeatureSampling
    q = tail((torch.rand(2, 2048, 3, generator=g) * 30 + 0.5).to(dev))
    f = tail(torch.rand(2, 32, 32, 32, 32, generator=g).to(dev)).requires_grad_(True)
    CubicFeatureSampling()(q, f).sum().backward()
elif op == "knn":
    from sparenet_amd.cuda.knn import get_graph_feature, knn
    x = tail(torch.rand(4, 64, 3000, generator=g).to(dev)).requires_grad_(True)
    idx = knn(x.detach(), 8)
    get_graph_feature(x, k=8, idx=idx).sum().backward()
torch.cuda.synchronize()
print("DONE")
