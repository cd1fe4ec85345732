"""The drop-in boundary as the reference uses it: after sparenet_amd.alias_reference_modules()
 the reference's own import lines -- runners/sparenet_runner.py:9-10,
models/sparenet_generator.py:8-9, utils/p2i_utils.py:7, models/grnet_generator.py:5-6 -- resolve to the
MI355X implementation, and the ops called through those names agree with the oracle."""
import sys
import types

import numpy as np
import pytest
import torch

import oracle


@pytest.fixture()
def aliased():
    import sparenet_amd

    saved = {k: v for k, v in sys.modules.items() if k == "cuda" or k.startswith("cuda.") or k.startswith("utils")}
    if "utils" not in sys.modules:            # stands in for the reference checkout's own `utils` package
        sys.modules["utils"] = types.ModuleType("utils")
        sys.modules["utils"].__path__ = []
    sparenet_amd.alias_reference_modules()
    yield
    for k in [k for k in sys.modules if k == "cuda" or k.startswith("cuda.") or k.startswith("utils")]:
        del sys.modules[k]
    sys.modules.update(saved)


def test_reference_import_lines_resolve(aliased):
    from cuda.chamfer_distance import ChamferDistance, ChamferDistanceMean          # sparenet_runner.py:10
    from cuda.emd.emd_module import emdModule                                       # sparenet_runner.py:9
    from cuda.expansion_penalty.expansion_penalty_module import expansionPenaltyModule  # generator:9
    import cuda.MDS.MDS_module as MDS_module                                         # generator:8
    from cuda.p2i_op import p2i, P2IMaxFunction, P2ISumFunction                      # utils/p2i_utils.py:7
    from cuda.gridding import Gridding, GriddingReverse                              # grnet_generator.py:5
    from cuda.cubic_feature_sampling import CubicFeatureSampling                     # grnet_generator.py:6
    from cuda.chamfer_dist import ChamferFunction, ChamferDistanceSeperate           # cuda/chamfer_dist:6-52
    from cuda.gridding_loss import GriddingLoss
    from utils.p2i_utils import ComputeDepthMaps, N_VIEWS_PREDEFINED                 # utils/model_init.py:9
    import sparenet_amd.cuda.emd.emd_module as real

    assert emdModule is real.emdModule and N_VIEWS_PREDEFINED == 8
    assert hasattr(MDS_module, "minimum_density_sample") and hasattr(MDS_module, "gather_operation")
    assert all(callable(f) for f in (ChamferDistance, ChamferDistanceMean, expansionPenaltyModule, p2i,
                                     P2IMaxFunction, P2ISumFunction, Gridding, GriddingReverse,
                                     CubicFeatureSampling, ChamferFunction, ChamferDistanceSeperate,
                                     GriddingLoss, ComputeDepthMaps))


@pytest.mark.gpu
def test_ops_through_the_reference_names(aliased, dev):
    from cuda.chamfer_distance import ChamferDistance
    from cuda.emd.emd_module import emdModule
    from cuda.expansion_penalty.expansion_penalty_module import expansionPenaltyModule
    import cuda.MDS.MDS_module as MDS_module
    from cuda.p2i_op import p2i
    from utils.p2i_utils import ComputeDepthMaps

    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 1024, 3, generator=g)
    y = torch.rand(2, 1024, 3, generator=g)
    xd, yd = x.to(dev), y.to(dev)
    # the call shapes of runners/sparenet_runner.py:83-108 / models/sparenet_generator.py:558-579
    dist, assign = emdModule()(xd, yd, eps=0.005, iters=10)
    od, oa = oracle.emd_forward(x.numpy(), y.numpy(), 0.005, 10)
    assert np.array_equal(assign.cpu().numpy(), oa) and np.array_equal(dist.cpu().numpy(), od)
    d1, d2 = ChamferDistance()(xd, yd)
    o1, o2, _, _ = oracle.chamfer_forward(x.numpy(), y.numpy())
    assert np.array_equal(d1.cpu().numpy(), o1) and np.array_equal(d2.cpu().numpy(), o2)
    pen, pa, mml = expansionPenaltyModule()(xd, 256, 1.5)
    opd, opa, opm = oracle.expansion_forward(x.numpy(), 256, 1.5)
    assert np.array_equal(pen.cpu().numpy(), opd) and np.array_equal(pa.cpu().numpy(), opa)
    idx = MDS_module.minimum_density_sample(xd, 300, mml)
    assert np.array_equal(idx.cpu().numpy(), oracle.mds(x.numpy(), 300, mml.cpu().numpy(), exp_mode=1))
    feats = xd.transpose(1, 2).contiguous()
    out = MDS_module.gather_operation(feats, idx)
    assert np.array_equal(out.cpu().numpy(), oracle.gather_forward(feats.cpu().numpy(), idx.cpu().numpy()))
    pts = torch.rand(2 * 400, 2, generator=g) * 2 - 1
    feat = torch.rand(2 * 400, 1, generator=g)
    bi = torch.arange(2, dtype=torch.int32).repeat_interleave(400)
    img = p2i(pts.to(dev), feat.to(dev), bi.to(dev), torch.zeros(2, 1, 32, 32, device=dev), 3.0, "cos", "max")
    oo, _ = oracle.p2i_max_forward(((pts + 1) / 2 * 31).numpy(), feat.numpy(), bi.numpy(),
                                   np.zeros((2, 1, 32, 32), np.float32), 3.0)
    np.testing.assert_allclose(img.cpu().numpy(), oo, rtol=2e-6, atol=1e-7)
    maps = ComputeDepthMaps("orthorgonal", 1.0, 64).to(dev)(xd - 0.5, view_id=2, radius_list=[5.0, 7.0])
    assert maps.shape == (2, 2, 64, 64) and float(maps.max()) <= 1.0
