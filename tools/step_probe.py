"""One pass of every hot op at the benched sizes, for profiler passes (tools/pmc_all.sh, tools/kstats.sh).
Env: PROBE_MDS=0 skips the sampler, PROBE_VIEWS=<n> renders n views one by one instead of 8 in one pass."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import sparenet_amd._lib as _L
if os.environ.get("AB_LIB"):
    _L.LIB_PATH = os.path.abspath(os.environ["AB_LIB"])
from sparenet_amd.cuda.chamfer_distance import ChamferDistance
from sparenet_amd.cuda.emd.emd_module import emdModule
from sparenet_amd.cuda.expansion_penalty.expansion_penalty_module import expansionPenaltyModule
from sparenet_amd.cuda.MDS.MDS_module import minimum_density_sample
from sparenet_amd.utils.p2i_utils import ComputeDepthMaps

dev = torch.device("cuda:0")
B, N = 32, 16384
g = torch.Generator().manual_seed(1234)
pred = torch.rand(B, N, 3, generator=g).to(dev)
gt = torch.rand(B, N, 3, generator=g).to(dev)
render = ComputeDepthMaps("orthorgonal", 1.0, 256).to(dev)
for rep in range(int(os.environ.get("PROBE_REPS", "2"))):
    p = pred.clone().requires_grad_(True)
    pen, _, mml = expansionPenaltyModule()(p, 512, 1.5)
    pen.mean().backward()
    p = pred.clone().requires_grad_(True)
    q = gt.clone().requires_grad_(True)
    d1, d2 = ChamferDistance()(p, q)
    (d1.mean() + d2.mean()).backward()
    p = pred.clone().requires_grad_(True)
    dist, _ = emdModule()(p, gt, 0.005, 50)
    torch.sqrt(dist).mean().backward()
    p4 = (pred - 0.5).requires_grad_(True)
    # all 8 views in one pass, as bench.py renders them: the counters of a gather dispatch then describe the launch
    # bench.py times (PROBE_VIEWS=<n>: the reference's view-by-view loop over n views instead)
    if os.environ.get("PROBE_VIEWS"):
        acc = 0
        for v in range(int(os.environ["PROBE_VIEWS"])):
            acc = acc + render(p4, view_id=v, radius_list=[5.0, 7.0, 10.0]).mean()
    else:
        acc = render.forward_views(p4, range(8), [5.0, 7.0, 10.0]).mean() * 8
    acc.backward()
    torch.cuda.synchronize()
if os.environ.get("PROBE_MDS", "1") != "0":
    v = torch.randn(B, N, 3, generator=g)
    v = 0.5 * v / v.norm(dim=2, keepdim=True)
    key = (torch.atan2(v[..., 1], v[..., 0]) * 4).floor() * 100 + (v[..., 2] * 8).floor()
    surf = torch.gather(v, 1, key.argsort(dim=1).unsqueeze(-1).expand(-1, -1, 3)).contiguous().to(dev)
    _, _, mml_s = expansionPenaltyModule()(surf, 512, 1.5)
    cloud_s = torch.cat([surf, surf[:, :3000] + 0.01 * torch.randn(B, 3000, 3, generator=g).to(dev)], 1).contiguous()
    minimum_density_sample(cloud_s, N, mml_s)
    torch.cuda.synchronize()
if os.environ.get("PROBE_MDS_DENSE", "1") != "0":
    # the DENSE regime (the cut ball covers the cloud: a team of workgroups per cloud, mds_dense_team_kernel) -- what
    # the first sampler call of an untrained generator is: uniform cloud, mean MST length 0.05 / 0.0853
    dense = torch.rand(B, N + 3000, 3, generator=g).to(dev)
    for bsz in (32, 4):
        minimum_density_sample(dense[:bsz].contiguous(), N, torch.full((bsz,), 0.05, device=dev))
    torch.cuda.synchronize()
print("probe done")
