"""Every ctypes call site of the host-side mirror, exercised on the CPU against a stub of the library that only
CONVERTS the arguments with the argtypes parsed from include/sparenet_hip.h (sparenet_amd/_lib.py): an argument of
the wrong kind or count at any call site (a numpy integer where a C int is declared, a missing workspace size, a
float passed positionally for a pointer) fails here, without a GPU, instead of corrupting memory on one.
No result is looked at: the stub computes nothing."""
import ctypes

import pytest
import torch

import sparenet_amd._lib as L


class _Stub:
    def __init__(self, real):
        self._real = real
        self.calls = {}

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if not name.startswith("sn_") or name in ("sn_last_error", "sn_build_id", "sn_abi_version"):
            return fn
        argtypes, restype = fn.argtypes, fn.restype
        assert argtypes is not None, f"{name}: no prototype parsed from the header"

        def call(*args):
            assert len(args) == len(argtypes), f"{name}: {len(args)} arguments, prototype has {len(argtypes)}"
            for i, (t, a) in enumerate(zip(argtypes, args)):
                try:
                    t.from_param(a)
                except (TypeError, ctypes.ArgumentError) as e:
                    raise AssertionError(f"{name}: argument {i} ({a!r}) does not convert to {t.__name__}: {e}")
            self.calls[name] = self.calls.get(name, 0) + 1
            return 4096 if restype is ctypes.c_size_t else 0

        return call


@pytest.fixture()
def stub(monkeypatch):
    real = L.lib()
    st = _Stub(real)
    monkeypatch.setattr(L, "_lib", st)
    want = {"fptr": torch.float32, "iptr": torch.int32, "dptr": torch.float64}

    def fake_ptr(t, dtype, name):
        assert isinstance(t, torch.Tensor), name
        assert t.dtype == dtype, f"{name}: expected {dtype}, got {t.dtype}"
        assert t.is_contiguous(), name
        return ctypes.c_void_p(t.data_ptr() or 8)

    monkeypatch.setattr(L, "ptr", fake_ptr)
    for k, dt in want.items():
        monkeypatch.setattr(L, k, lambda t, name, _dt=dt: fake_ptr(t, _dt, name))
    monkeypatch.setattr(L, "stream_of", lambda t: ctypes.c_void_p(0))
    return st


def test_every_wrapper_passes_convertible_arguments(stub):
    from sparenet_amd.cuda.chamfer_distance import ChamferDistance, cd
    from sparenet_amd.cuda.chamfer_dist import ChamferDistance as CD2
    from sparenet_amd.cuda.cubic_feature_sampling import CubicFeatureSampling
    from sparenet_amd.cuda.emd.emd_module import emdModule
    from sparenet_amd.cuda.expansion_penalty.expansion_penalty_module import expansionPenaltyModule
    from sparenet_amd.cuda.gridding import Gridding, GriddingReverse
    from sparenet_amd.cuda.gridding_loss import GriddingLoss
    from sparenet_amd.cuda.knn import get_graph_feature, knn_fused, knn_unfused
    from sparenet_amd.cuda.MDS.MDS_module import gather_operation, minimum_density_sample
    from sparenet_amd.cuda.p2i_op import p2i
    from sparenet_amd.utils.p2i_utils import ComputeDepthMaps

    x = torch.rand(2, 1024, 3, requires_grad=True)
    y = torch.rand(2, 1024, 3, requires_grad=True)
    d1, d2 = ChamferDistance()(x, y)          # CPU tensors: the host entry points (sn_chamfer_*_host)
    (d1.sum() + d2.sum()).backward()
    # the device entry points, as the wrapper calls them for CUDA tensors (the stub's fake pointers stand in)
    xs, ys = x.detach(), y.detach()
    e, ei = torch.empty(2, 1024), torch.empty(2, 1024, dtype=torch.int)
    cd.forward_cuda(xs, ys, e, e.clone(), ei, ei.clone())
    cd.backward_cuda(xs, ys, torch.empty_like(xs), torch.empty_like(ys), e, e.clone(), ei, ei.clone())
    big = torch.rand(1, 4096, 3)
    cd.forward_sorted_cuda(big, big, torch.empty(1, 4096), torch.empty(1, 4096),
                           torch.empty(1, 4096, dtype=torch.int), torch.empty(1, 4096, dtype=torch.int))
    CD2()(x, y)
    dist, _ = emdModule()(x, y, eps=0.005, iters=3)
    dist.sum().backward()
    pen, _, mml = expansionPenaltyModule()(x, 512, 1.5)
    pen.sum().backward()
    idx = minimum_density_sample(x.detach(), 256, mml.detach())
    feat = torch.rand(2, 4, 1024, requires_grad=True)
    gather_operation(feat, torch.zeros(2, 256, dtype=torch.int32)).sum().backward()
    for dt in (torch.float32, torch.float64):
        pts = torch.rand(64, 2, dtype=dt, requires_grad=True)
        ft = torch.rand(64, 1, dtype=dt, requires_grad=True)
        bi = torch.zeros(64, dtype=torch.int32)
        bg = torch.zeros(1, 1, 16, 16, dtype=dt, requires_grad=True)
        for red in ("max", "sum"):
            p2i(pts, ft, bi, bg, 2.0, "cos", red).sum().backward()
    from sparenet_amd.cuda.p2i_op import P2IMaxMultiFunction, ext
    pts, ft = torch.rand(64, 2), torch.rand(64, 1)
    bi, bg = torch.zeros(64, dtype=torch.int32), torch.zeros(1, 1, 16, 16)
    out, ids = ext.p2i_max_forward_gpu(pts, ft, bi, bg, 0, 2.0)
    ext.p2i_max_backward_gpu(torch.ones_like(out), ids, pts, ft, 0, 2.0)      # the fp32 single-radius entry
    P2IMaxMultiFunction.apply(pts.requires_grad_(True), ft, bi, (1, 1, 16, 16), 0, [2.0, 3.0], True).sum().backward()
    # the fused projection kernels (ComputeDepthMaps routes CUDA fp32 tensors to them)
    from sparenet_amd.utils.p2i_utils import DepthProjectFunction, DepthProjectViewsFunction
    cdm = ComputeDepthMaps("orthorgonal", 1.0, 32)
    p4 = (torch.rand(2, 256, 3) - 0.5).requires_grad_(True)
    ij, f = DepthProjectFunction.apply(p4, cdm._host_mats[1], 32)
    (ij.sum() + f.sum()).backward()
    ij, f = DepthProjectViewsFunction.apply(p4, [cdm._host_mats[v] for v in range(8)], 32)
    (ij.sum() + f.sum()).backward()
    pc = (torch.rand(2, 64, 3) - 0.5).requires_grad_(True)
    Gridding(8)(pc).sum().backward()
    GriddingReverse(4)(torch.rand(1, 4, 4, 4, requires_grad=True)).sum().backward()
    GriddingLoss([8], [1.0])(pc, pc.detach().clone())
    CubicFeatureSampling()(torch.rand(1, 8, 3), torch.rand(1, 2, 4, 4, 4, requires_grad=True)).sum().backward()
    xf = torch.rand(1, 16, 128, requires_grad=True)
    nbr = knn_fused(xf.detach(), 8)
    knn_unfused(xf.detach(), 8)
    get_graph_feature(xf, k=8, idx=torch.zeros(1, 128, 8, dtype=torch.int64)).sum().backward()
    called = set(stub.calls)
    for must in ("sn_chamfer_forward", "sn_chamfer_backward", "sn_chamfer_forward_sorted", "sn_chamfer_forward_host",
                 "sn_chamfer_backward_host", "sn_emd_forward",
                 "sn_emd_backward", "sn_expansion_forward", "sn_expansion_backward", "sn_mds", "sn_gather_forward",
                 "sn_gather_backward", "sn_p2i_max_forward", "sn_p2i_max_backward", "sn_p2i_sum_forward",
                 "sn_p2i_sum_backward", "sn_p2i_max_forward_f64", "sn_p2i_sum_backward_f64",
                 "sn_depth_project_forward", "sn_depth_project_backward", "sn_depth_project_forward_views",
                 "sn_depth_project_backward_views", "sn_p2i_max_forward_multi", "sn_p2i_max_backward_multi",
                 "sn_gridding_forward_padded", "sn_gridding_backward", "sn_gridding_reverse_forward",
                 "sn_gridding_reverse_backward", "sn_cubic_forward", "sn_cubic_backward", "sn_knn",
                 "sn_graph_feature_forward", "sn_graph_feature_backward"):
        assert must in called, f"{must} was never reached: {sorted(called)}"
