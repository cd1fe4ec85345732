"""Non-finite inputs must not take the process down: every op has to stay inside its buffers when
coordinates are NaN / inf (the values it returns for such rows are unspecified, as in the reference)."""
import numpy as np
import pytest
import torch


def _poison(t, g, frac=0.01):
    t = t.clone()
    n = t.numel() // t.shape[-1]
    rows = torch.randperm(n, generator=g)[: max(1, int(n * frac))]
    flat = t.view(n, t.shape[-1])
    flat[rows[::3], 0] = float("nan")
    flat[rows[1::3], 1] = float("inf")
    flat[rows[2::3], 2] = -float("inf")
    return t


@pytest.mark.gpu
def test_ops_survive_non_finite_inputs(dev):
    from sparenet_amd.cuda.chamfer_distance import ChamferDistanceFunction
    from sparenet_amd.cuda.emd.emd_module import emdModule
    from sparenet_amd.cuda.expansion_penalty.expansion_penalty_module import expansionPenaltyModule
    from sparenet_amd.cuda.MDS.MDS_module import gather_operation, minimum_density_sample
    from sparenet_amd.utils.p2i_utils import ComputeDepthMaps

    g = torch.Generator().manual_seed(77)
    x = _poison(torch.rand(2, 2048, 3, generator=g), g).to(dev).requires_grad_(True)
    y = _poison(torch.rand(2, 2048, 3, generator=g), g).to(dev)
    d1, d2 = ChamferDistanceFunction.apply(x, y)                       # sorted search (2^22 pairs)
    (d1[torch.isfinite(d1)].sum() + d2[torch.isfinite(d2)].sum()).backward()
    dist, assign = emdModule()(x, y, 0.005, 8)
    assert assign.shape == (2, 2048) and int(assign.max()) < 2048 and int(assign.min()) >= -1
    dist[torch.isfinite(dist)].sum().backward()
    pen, _, mml = expansionPenaltyModule()(x, 64, 1.5)
    idx = minimum_density_sample(torch.cat([x.detach(), y[:, :300]], 1).contiguous(), 2048, mml)
    assert int(idx.min()) >= 0 and int(idx.max()) < 2348
    feat = torch.rand(2, 4, 2348, generator=g).to(dev)
    assert gather_operation(feat, idx).shape == (2, 4, 2048)
    maps = ComputeDepthMaps("orthorgonal", 1.0, 64).to(dev)(x - 0.5, view_id=1, radius_list=[3.0, 5.0])
    assert maps.shape == (2, 2, 64, 64)
    torch.cuda.synchronize()
    # and the library is still usable afterwards
    a, b = ChamferDistanceFunction.apply(torch.rand(1, 512, 3, generator=g).to(dev),
                                         torch.rand(1, 512, 3, generator=g).to(dev))
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
