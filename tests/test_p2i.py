"""p2i splat and ComputeDepthMaps.

CPU: oracle vs golden vectors produced by the reference's own functors
(tests/golden/gen_p2i.py) incl. the 8x8 known answer of cuda/p2i_op/p2i_test.py:10-20;
the host-side ComputeDepthMaps camera matrices / projected coordinates vs the imported
reference (tests/golden/gen_depthmaps.py).
GPU: HIP vs oracle and vs the golden vectors.  Parity bar: values 1e-6 relative (the
cosine weight goes through libm on the CPU and OCML on the GPU), winner ids exact
except where two candidates are within that tolerance.
"""
import glob
import os

import numpy as np
import pytest
import torch

import oracle


def _golden(golden_dir, pat):
    files = sorted(glob.glob(os.path.join(golden_dir, pat)))
    assert files, pat
    return files


# ------------------------------------------------------------------ CPU side
def test_oracle_matches_reference_functor_golden(golden_dir):
    for f in _golden(golden_dir, "p2i_*.npz"):
        z = np.load(f)
        R = float(z["radius"])
        out, ids = oracle.p2i_max_forward(z["points"], z["feat"], z["batch_inds"], z["background"], R)
        assert np.array_equal(out, z["max_out"]) and np.array_equal(ids, z["max_ids"]), f
        gp, gf, gb = oracle.p2i_max_backward(z["out_grad"], z["max_ids"], z["points"], z["feat"], R)
        assert np.array_equal(gp, z["max_points_grad"]), f
        assert np.array_equal(gf, z["max_feat_grad"]) and np.array_equal(gb, z["max_background_grad"]), f
        so = oracle.p2i_sum_forward(z["points"], z["feat"], z["batch_inds"], z["background"], R)
        assert np.array_equal(so, z["sum_out"]), f
        sgp, sgf = oracle.p2i_sum_backward(z["out_grad"], z["points"], z["feat"], z["batch_inds"], R)
        assert np.array_equal(sgp, z["sum_points_grad"]) and np.array_equal(sgf, z["sum_feat_grad"]), f


def test_known_answer_8x8(golden_dir):
    """One point at the centre of an 8x8 map, radius 2, feature 1 (p2i_test.py test1):
    4 centre pixels at r = sqrt(0.5), 8 ring pixels at r = sqrt(2.5), nothing else."""
    z = np.load(os.path.join(golden_dir, "p2i_known_8x8_r2.npz"))
    out, ids = oracle.p2i_max_forward(z["points"], z["feat"], z["batch_inds"], z["background"], 2.0)
    img = out[0, 0]
    c = np.cos(np.sqrt(0.5) * np.pi / 2) * 0.5 + 0.5
    r = np.cos(np.sqrt(2.5) * np.pi / 2) * 0.5 + 0.5
    assert np.allclose(img[3:5, 3:5], c, rtol=1e-6)
    ring = [(2, 3), (2, 4), (5, 3), (5, 4), (3, 2), (4, 2), (3, 5), (4, 5)]
    assert all(np.isclose(img[y, x], r, rtol=1e-6) for y, x in ring)
    assert np.count_nonzero(img) == 12
    assert ((ids[0, 0] == 0) == (img > 0)).all()


def test_depthmaps_host_glue_vs_reference(golden_dir, monkeypatch):
    """ComputeDepthMaps (host mirror) against the imported reference: the eight P@V
    matrices bit for bit, projected coordinates / features to 1e-6."""
    from sparenet_amd.utils import p2i_utils

    for f in _golden(golden_dir, "depthmaps_*.npz"):
        z = np.load(f)
        proj = "orthorgonal" if "ortho" in f else "perspective"
        cdm = p2i_utils.ComputeDepthMaps(proj, float(z["eyepos_scale"]), int(z["image_size"]))
        assert np.array_equal(cdm.pre_matrices.numpy(), z["pre_matrices"]), f
        data = torch.from_numpy(z["data"])
        for v in range(8):
            ij, feat = cdm.project(data, v)
            assert np.array_equal(ij.numpy(), z[f"ij_{v}"]), (f, v)        # same operation order: bit-equal
            assert np.array_equal(feat.numpy(), z[f"feat_{v}"]), (f, v)
        assert cdm(data, view_id=8) is None
    assert p2i_utils.N_VIEWS_PREDEFINED == 8


def test_oracle_mt_splat_equals_sequential():
    """bench.py's multi-threaded CPU baseline (clouds painted in parallel) is the sequential splat bit for bit."""
    rng = np.random.default_rng(0)
    B, n, S = 5, 3000, 80
    pts = ((rng.random((B * n, 2), dtype=np.float32) * 1.1 - 0.05) * (S - 1)).astype(np.float32)
    feat = rng.random((B * n, 1), dtype=np.float32)
    bi = rng.integers(-1, B + 1, B * n).astype(np.int32)
    bg = np.full((B, 1, S, S), 0.01, np.float32)
    a = oracle.p2i_max_forward(pts, feat, bi, bg, 6.0)
    b = oracle.p2i_max_forward(pts, feat, bi, bg, 6.0, mt=True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


# ------------------------------------------------------------------ GPU side
def _close_maps(out, ids, ref_out, ref_ids, what, pts, feat, bg, R):
    """values at 2e-6 (series vs glibc cosine); winner ids EXACT except verified one-ulp ties (tests/p2i_check.py)"""
    from p2i_check import assert_ids_exact_up_to_ulp_ties

    np.testing.assert_allclose(out, ref_out, rtol=2e-6, atol=1e-7, err_msg=what)
    as_np = lambda t: t.cpu().numpy() if hasattr(t, "cpu") else np.asarray(t)
    ties = assert_ids_exact_up_to_ulp_ties(ids, ref_ids, as_np(pts), as_np(feat), as_np(bg), R, what)
    assert ties <= max(2, ids.size // 10000), (what, ties)   # and they are rare


def _assert_exact_accumulation(gp, gf, out_grad, ids, points, feat, R, what, scale=1.0):
    """The fixed-point backward against the oracle's EXACT sum of the same fp32 terms: what is left is the one
    rounding of the result, the 2^-45 fixed-point step and the last bit of a term (the HIP path evaluates
    sin(pi r / R) / r with its own fp64 sequence): 2e-6 relative + 1e-7 of the largest gradient."""
    ep, ef = oracle.p2i_max_backward_exact(out_grad, ids, points, feat, R)
    ep = ep * scale
    np.testing.assert_allclose(gp, ep, rtol=2e-6, atol=1e-7 * max(1.0, float(np.abs(ep).max())), err_msg=str(what))
    np.testing.assert_allclose(gf, ef, rtol=2e-6, atol=1e-7 * max(1.0, float(np.abs(ef).max())), err_msg=str(what))


@pytest.mark.gpu
def test_hip_matches_functor_golden(golden_dir, dev):
    from sparenet_amd.cuda.p2i_op import ext

    for f in _golden(golden_dir, "p2i_*.npz"):
        z = np.load(f)
        R = float(z["radius"])
        t = {k: torch.from_numpy(z[k]).to(dev) for k in
             ("points", "feat", "batch_inds", "background", "out_grad", "max_ids")}
        out, ids = ext.p2i_max_forward_gpu(t["points"], t["feat"], t["batch_inds"], t["background"], 0, R)
        _close_maps(out.cpu().numpy(), ids.cpu().numpy(), z["max_out"], z["max_ids"], f,
                    z["points"], z["feat"], z["background"], R)
        gp, gf, gb = ext.p2i_max_backward_gpu(t["out_grad"], t["max_ids"], t["points"], t["feat"], 0, R)
        # Tolerances of the gradient comparisons in this file.  A point's gradient is a SUM of up to ~pi R^2 signed
        # fp32 terms (one per pixel it won).  The golden (the reference functor, run sequentially) and the oracle add
        # them in fp32 in pixel order, the reference's GPU with fp32 atomics in arrival order: each of those sums is
        # only defined up to (#terms) 2^-24 sum|terms| (80 ... 314 terms at R = 5 ... 10: 0.5 ... 2e-5 of the
        # magnitude sum, more relative to a sum that cancels) -- the rtol 2e-5 ... 5e-5 / atol 2e-6 ... 5e-6 below are
        # that bound, not slack of the HIP path.  The HIP path adds the same fp32 terms EXACTLY (64-bit fixed point)
        # and is pinned much tighter against the oracle's exact accumulation of those terms
        # (_assert_exact_accumulation: 2e-6 relative to the result, i.e. north_star's 1e-5 with room).
        np.testing.assert_allclose(gp.cpu().numpy(), z["max_points_grad"], rtol=2e-5, atol=2e-6, err_msg=f)
        np.testing.assert_allclose(gf.cpu().numpy(), z["max_feat_grad"], rtol=2e-5, atol=2e-6, err_msg=f)
        assert np.array_equal(gb.cpu().numpy(), z["max_background_grad"]), f
        # the pixel-centric backward with exact fixed-point accumulation (what autograd uses)
        m1 = ext.p2i_max_backward_multi_gpu(t["out_grad"][None], t["max_ids"][None], t["points"], t["feat"], 0, [R])
        m2 = ext.p2i_max_backward_multi_gpu(t["out_grad"][None], t["max_ids"][None], t["points"], t["feat"], 0, [R])
        np.testing.assert_allclose(m1[0].cpu().numpy(), z["max_points_grad"], rtol=2e-5, atol=2e-6, err_msg=f)
        np.testing.assert_allclose(m1[1].cpu().numpy(), z["max_feat_grad"], rtol=2e-5, atol=2e-6, err_msg=f)
        assert np.array_equal(m1[2].cpu().numpy(), z["max_background_grad"]), f
        assert all(torch.equal(a, b) for a, b in zip(m1, m2)), "integer accumulation is order independent"
        _assert_exact_accumulation(m1[0].cpu().numpy(), m1[1].cpu().numpy(), z["out_grad"], z["max_ids"], z["points"],
                                   z["feat"], R, f)
        so = ext.p2i_sum_forward_gpu(t["points"], t["feat"], t["batch_inds"], t["background"], 0, R)
        np.testing.assert_allclose(so.cpu().numpy(), z["sum_out"], rtol=2e-5, atol=2e-6, err_msg=f)
        sgp, sgf = ext.p2i_sum_backward_gpu(t["out_grad"], t["points"], t["feat"], t["batch_inds"], 0, R)
        np.testing.assert_allclose(sgp.cpu().numpy(), z["sum_points_grad"], rtol=2e-5, atol=2e-6, err_msg=f)
        np.testing.assert_allclose(sgf.cpu().numpy(), z["sum_feat_grad"], rtol=2e-5, atol=2e-6, err_msg=f)


@pytest.mark.gpu
@pytest.mark.parametrize("B,n,C,S,R", [(2, 3000, 1, 64, 10.0), (1, 5000, 2, 96, 7.0),
                                        (3, 100, 1, 17, 1.0), (1, 1, 1, 4, 40.0), (2, 4096, 1, 128, 16.5)])
def test_hip_max_matches_oracle(B, n, C, S, R, dev):
    from sparenet_amd.cuda.p2i_op import ext

    g = torch.Generator().manual_seed(B * 100 + n)
    pts = (torch.rand(B * n, 2, generator=g) * 1.2 - 0.1) * (S - 1)
    feat = torch.rand(B * n, C, generator=g)
    bi = torch.arange(B, dtype=torch.int32).repeat_interleave(n)
    bg = torch.full((B, C, S, S), 0.05)
    o, i = oracle.p2i_max_forward(pts.numpy(), feat.numpy(), bi.numpy(), bg.numpy(), R)
    out, ids = ext.p2i_max_forward_gpu(pts.to(dev), feat.to(dev), bi.to(dev), bg.to(dev), 0, R)
    _close_maps(out.cpu().numpy(), ids.cpu().numpy(), o, i, "max fwd", pts, feat, bg, R)


@pytest.mark.gpu
@pytest.mark.parametrize("B,n,C,H,W,radii", [(2, 3000, 1, 64, 64, [5.0, 7.0, 10.0]),
                                             (3, 2000, 2, 50, 70, [10.0, 3.0]),
                                             (1, 500, 1, 33, 31, [16.0, 0.7, 2.5, 9.0]),
                                             (2, 1500, 1, 48, 48, [20.0, 5.0])])
def test_hip_multi_radius_matches_oracle_and_single(B, n, C, H, W, radii, dev):
    """sn_p2i_max_forward_multi (tile-binned, all radii in one pass; radii > 16 px fall back to
    the global splat) == one single-radius call per radius == the oracle; shuffled batch ids
    and points outside the image included."""
    from sparenet_amd.cuda.p2i_op import ext

    g = torch.Generator().manual_seed(B * 1000 + n)
    pts = (torch.rand(B * n, 2, generator=g) * 1.3 - 0.15) * torch.tensor([H - 1.0, W - 1.0])
    feat = torch.rand(B * n, C, generator=g) - 0.2
    bi = torch.randint(-1, B + 1, (B * n,), generator=g).to(torch.int32)   # some ids out of range
    bg = torch.rand(B, C, H, W, generator=g) * 0.1
    out, ids = ext.p2i_max_forward_multi_gpu(pts.to(dev), feat.to(dev), bi.to(dev), bg.to(dev), 0, radii)
    assert out.shape == (len(radii), B, C, H, W)
    if max(radii) <= 16.0:   # the image-major layout holds the same maps, [B, R, ...] instead of [R, B, ...]
        out_im, ids_im = ext.p2i_max_forward_multi_gpu(pts.to(dev), feat.to(dev), bi.to(dev), bg.to(dev), 0, radii,
                                                       image_major=True)
        assert torch.equal(out_im.transpose(0, 1), out) and torch.equal(ids_im.transpose(0, 1), ids)
        og = torch.rand(out.shape, generator=torch.Generator().manual_seed(3)).to(dev)
        g_a = ext.p2i_max_backward_multi_gpu(og, ids, pts.to(dev), feat.to(dev), 0, radii)
        g_b = ext.p2i_max_backward_multi_gpu(og.transpose(0, 1).contiguous(), ids_im, pts.to(dev), feat.to(dev), 0,
                                             radii, image_major=True)
        assert all(torch.equal(a, b) for a, b in zip(g_a, g_b))
    for r, R in enumerate(radii):
        o1, i1 = ext.p2i_max_forward_gpu(pts.to(dev), feat.to(dev), bi.to(dev), bg.to(dev), 0, R)
        assert torch.equal(out[r], o1) and torch.equal(ids[r], i1), R
        o, i = oracle.p2i_max_forward(pts.numpy(), feat.numpy(), bi.numpy(), bg.numpy(), R)
        _close_maps(out[r].cpu().numpy(), ids[r].cpu().numpy(), o, i, f"multi R={R}", pts, feat, bg, R)


@pytest.mark.gpu
def test_hip_multi_radius_autograd_is_sum_of_singles(dev):
    from sparenet_amd.cuda.p2i_op import P2IMaxFunction, P2IMaxMultiFunction

    g = torch.Generator().manual_seed(5)
    B, n, S, radii = 2, 800, 40, [3.0, 6.0, 9.0]
    pts0 = (torch.rand(B * n, 2, generator=g) * (S - 1)).to(dev)
    feat0 = torch.rand(B * n, 1, generator=g).to(dev)
    bi = torch.arange(B, dtype=torch.int32).repeat_interleave(n).to(dev)
    bg = torch.zeros(B, 1, S, S, device=dev)
    wgt = torch.rand(len(radii), B, 1, S, S, generator=g).to(dev)
    p1, f1 = pts0.clone().requires_grad_(True), feat0.clone().requires_grad_(True)
    (P2IMaxMultiFunction.apply(p1, f1, bi, bg, 0, radii) * wgt).sum().backward()
    p2, f2 = pts0.clone().requires_grad_(True), feat0.clone().requires_grad_(True)
    sum((P2IMaxFunction.apply(p2, f2, bi, bg, 0, R) * wgt[r]).sum() for r, R in enumerate(radii)).backward()
    np.testing.assert_allclose(p1.grad.cpu().numpy(), p2.grad.cpu().numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(f1.grad.cpu().numpy(), f2.grad.cpu().numpy(), rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
def test_hip_ties_pick_lowest_point_id(dev):
    """Duplicated points produce bit-equal splat values: the lowest id must win; a point
    whose value merely equals the background must not replace it (id stays -1)."""
    from sparenet_amd.cuda.p2i_op import ext

    pts = torch.tensor([[5.0, 5.0]] * 4 + [[10.0, 10.0]])
    feat = torch.tensor([[0.7], [0.7], [0.9], [0.9], [0.0]])
    bi = torch.zeros(5, dtype=torch.int32)
    bg = torch.zeros(1, 1, 16, 16)
    out, ids = ext.p2i_max_forward_gpu(pts.to(dev), feat.to(dev), bi.to(dev), bg.to(dev), 0, 2.0)
    o, i = oracle.p2i_max_forward(pts.numpy(), feat.numpy(), bi.numpy(), bg.numpy(), 2.0)
    ids = ids.cpu().numpy()
    assert np.array_equal(ids, i)
    assert ids[0, 0, 5, 5] == 2 and ids[0, 0, 10, 10] == -1
    np.testing.assert_allclose(out.cpu().numpy(), o, rtol=2e-6)


@pytest.mark.gpu
def test_hip_p2i_autograd_vs_oracle(dev):
    from sparenet_amd.cuda.p2i_op import p2i

    g = torch.Generator().manual_seed(5)
    B, n, S, R = 2, 600, 32, 3.0
    pts = (torch.rand(B * n, 2, generator=g) * 2 - 1)
    feat = torch.rand(B * n, 1, generator=g)
    bi = torch.arange(B, dtype=torch.int32).repeat_interleave(n)
    bg = torch.zeros(B, 1, S, S)
    og = torch.rand(B, 1, S, S, generator=g)
    for reduce in ("max", "sum"):
        p = pts.to(dev).requires_grad_(True)
        f = feat.to(dev).requires_grad_(True)
        b = bg.to(dev).requires_grad_(True)
        out = p2i(p, f, bi.to(dev), b, R, "cos", reduce)
        (out * og.to(dev)).sum().backward()
        px = ((pts + 1) / 2 * (S - 1)).numpy()
        if reduce == "max":
            o, ids = oracle.p2i_max_forward(px, feat.numpy(), bi.numpy(), bg.numpy(), R)
            gp, gf, gb = oracle.p2i_max_backward(og.numpy(), ids, px, feat.numpy(), R)
        else:
            o = oracle.p2i_sum_forward(px, feat.numpy(), bi.numpy(), bg.numpy(), R)
            gp, gf = oracle.p2i_sum_backward(og.numpy(), px, feat.numpy(), bi.numpy(), R)
            gb = og.numpy()
        np.testing.assert_allclose(out.detach().cpu().numpy(), o, rtol=2e-5, atol=2e-6)
        # autograd chains d pixel / d ndc = (S-1)/2
        np.testing.assert_allclose(p.grad.cpu().numpy(), gp * (S - 1) / 2, rtol=5e-5, atol=5e-6)
        np.testing.assert_allclose(f.grad.cpu().numpy(), gf, rtol=5e-5, atol=5e-6)
        np.testing.assert_allclose(b.grad.cpu().numpy(), gb, rtol=1e-6)
        if reduce == "max":   # the fp32-order bound above is the oracle's; against the exact sum of the terms:
            _assert_exact_accumulation(p.grad.cpu().numpy(), f.grad.cpu().numpy(), og.numpy(), ids, px, feat.numpy(), R,
                                       "autograd", scale=(S - 1) / 2)


@pytest.mark.gpu
@pytest.mark.parametrize("projection", ["orthorgonal", "perspective"])
def test_hip_fused_projection_matches_torch_glue(projection, dev):
    """DepthProjectFunction (sn_depth_project_*) against the torch mirror of the reference glue
    (ComputeDepthMaps.project + the NDC -> pixel rescale), values and gradients, including the
    gradient paths through the global zmin / zmax with several points attaining them."""
    from sparenet_amd.utils.p2i_utils import ComputeDepthMaps, DepthProjectFunction

    g = torch.Generator().manual_seed(21)
    cdm = ComputeDepthMaps(projection, 1.0, 64).to(dev)
    base = (torch.rand(3, 500, 3, generator=g) - 0.5)
    base[1, 7] = base[0, 3]          # duplicated points: ties at whatever extreme they reach
    base[2, 9] = base[0, 3]
    wp = torch.rand(1500, 2, generator=g).to(dev)
    wf = torch.rand(1500, 1, generator=g).to(dev)
    for v in (0, 3, 6):
        d1 = base.clone().to(dev).requires_grad_(True)
        pos_ijs, feat = cdm.project(d1, v)
        pix = (pos_ijs + 1) / 2 * 63.0
        ((pix * wp).sum() + (feat * wf).sum()).backward()
        d2 = base.clone().to(dev).requires_grad_(True)
        pix2, feat2 = DepthProjectFunction.apply(d2, cdm._host_mats[v], 64)
        ((pix2 * wp).sum() + (feat2 * wf).sum()).backward()
        np.testing.assert_allclose(pix2.detach().cpu().numpy(), pix.detach().cpu().numpy(), rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(feat2.detach().cpu().numpy(), feat.detach().cpu().numpy(), rtol=1e-5, atol=3e-6)
        # fp32 against fp32: the points that attain zmin / zmax collect a sum over ALL points, which torch's reduction
        # and the fused kernel add in different orders -- two fp32 evaluations of one quantity only agree this far:
        np.testing.assert_allclose(d2.grad.cpu().numpy(), d1.grad.cpu().numpy(), rtol=2e-4, atol=2e-4)
        # the contract (north_star: 1e-5) is against the VALUE: the same glue evaluated in float64
        d3 = base.clone().double().to(dev).requires_grad_(True)
        pos64, feat64 = cdm.project(d3, v)
        ((((pos64 + 1) / 2 * 63.0) * wp.double()).sum() + (feat64 * wf.double()).sum()).backward()
        ref = d3.grad.cpu().numpy()
        scale = float(np.abs(ref).max())
        err_hip = float(np.abs(d2.grad.cpu().numpy() - ref).max()) / scale
        err_torch = float(np.abs(d1.grad.cpu().numpy() - ref).max()) / scale
        assert err_hip <= 1e-5, (projection, v, "fused backward vs float64", err_hip, "torch fp32 mirror:", err_torch)


@pytest.mark.gpu
def test_hip_depthmaps_vs_reference_golden(golden_dir, dev):
    """End to end ComputeDepthMaps on the GPU against maps rendered by the imported
    reference (CPU torch glue + reference functor semantics)."""
    from sparenet_amd.utils.p2i_utils import ComputeDepthMaps

    for f in _golden(golden_dir, "depthmaps_*.npz"):
        z = np.load(f)
        proj = "orthorgonal" if "ortho" in f else "perspective"
        cdm = ComputeDepthMaps(proj, float(z["eyepos_scale"]), int(z["image_size"])).to(dev)
        data = torch.from_numpy(z["data"]).to(dev)
        radii = [float(r) for r in z["radius_list"]]
        from sparenet_amd.utils.p2i_utils import DepthProjectFunction
        S = int(z["image_size"])
        for v in range(8):
            # the fused projection evaluates the transform in the reference's order: pixel coordinates and
            # depth features are bit-equal to the imported reference's
            pix, feat = DepthProjectFunction.apply(data, cdm._host_mats[v], S)
            ref_pix = (torch.from_numpy(z[f"ij_{v}"]) + 1) / 2 * torch.tensor([[S - 1.0, S - 1.0]])
            assert np.array_equal(pix.cpu().numpy(), ref_pix.numpy()), (f, v)
            assert np.array_equal(feat.cpu().numpy(), z[f"feat_{v}"]), (f, v)
            got = cdm(data, view_id=v, radius_list=radii).cpu().numpy()
            ref = z[f"maps_{v}"]
            assert got.shape == ref.shape
            # same points on the same pixels: what is left is the cosine (OCML vs glibc, both in double)
            np.testing.assert_allclose(got, ref, rtol=2e-6, atol=1e-7, err_msg=f"{f} view {v}")


@pytest.mark.gpu
def test_hip_depthmaps_full_size_and_backward(dev):
    """BASELINE config 3 shape: [32,16384,3] -> 256^2, one view, radii in pixels; checks
    determinism, value range, coverage and that gradients reach the point cloud."""
    from sparenet_amd.utils.p2i_utils import ComputeDepthMaps

    g = torch.Generator().manual_seed(1234)
    data = (torch.rand(32, 16384, 3, generator=g) - 0.5).to(dev).requires_grad_(True)
    cdm = ComputeDepthMaps("orthorgonal", 1.0, 256).to(dev)
    maps = cdm(data, view_id=3, radius_list=[5.0, 7.0, 10.0])
    assert maps.shape == (32, 3, 256, 256)
    again = cdm(data, view_id=3, radius_list=[5.0, 7.0, 10.0])
    assert torch.equal(maps, again)
    assert float(maps.min()) >= 0.0 and float(maps.max()) <= 1.0
    cov = (maps > 0).float().mean(dim=(0, 2, 3))
    assert cov[0] < cov[1] < cov[2]
    maps.sum().backward()
    assert torch.isfinite(data.grad).all() and float(data.grad.abs().sum()) > 0
    tiny = cdm(data.detach(), view_id=0, radius_list=[0.02, 0.05])
    assert tiny.shape == (32, 2, 256, 256) and float((tiny > 0).float().mean()) < 0.01


@pytest.mark.gpu
def test_hip_empty_and_out_of_image_inputs(dev):
    """No points at all, and points that all fall outside the image / carry invalid batch ids: the output is
    the background, the ids are -1, gradients are zero (the reference's zeros-initialised outputs)."""
    from sparenet_amd.cuda.p2i_op import ext, p2i

    bg = torch.rand(2, 1, 16, 16, device=dev)
    none_p, none_f = torch.zeros(0, 2, device=dev), torch.zeros(0, 1, device=dev)
    none_b = torch.zeros(0, dtype=torch.int32, device=dev)
    out, ids = ext.p2i_max_forward_gpu(none_p, none_f, none_b, bg, 0, 3.0)
    assert torch.equal(out, bg) and int((ids != -1).sum()) == 0
    outm, idsm = ext.p2i_max_forward_multi_gpu(none_p, none_f, none_b, bg, 0, [3.0, 5.0])
    assert torch.equal(outm[0], bg) and torch.equal(outm[1], bg) and int((idsm != -1).sum()) == 0
    far = torch.full((5, 2), 9.0, device=dev).requires_grad_(True)            # NDC 9: far outside
    feat = torch.rand(5, 1, device=dev).requires_grad_(True)
    bi = torch.tensor([0, 1, 7, -1, 0], dtype=torch.int32, device=dev)
    o = p2i(far, feat, bi, bg, 2.0, "cos", "max")
    assert torch.equal(o, bg)
    o.sum().backward()
    assert float(far.grad.abs().sum()) == 0.0 and float(feat.grad.abs().sum()) == 0.0


# ------------------------------------------------------------------ float64 (the reference's own test surface)
def _np_p2i_f64(points, feat, bi, bg, radius, reduce):
    """Plain numpy float64 restatement of p2i_{sum,max}.h + utility.h:82-100 on pixel coordinates."""
    out = bg.copy()
    ids = np.full(bg.shape, -1, np.int64)
    B, C, H, W = bg.shape
    for pid in range(points.shape[0]):
        b = bi[pid]
        if b < 0 or b >= B:
            continue
        py, px = points[pid]
        x0, x1 = np.clip([np.floor(px - radius), np.ceil(px + radius)], 0, W - 1).astype(int)
        y0, y1 = np.clip([np.floor(py - radius), np.ceil(py + radius)], 0, H - 1).astype(int)
        for x in range(x0, x1 + 1):
            for y in range(y0, y1 + 1):
                r = np.sqrt((x - px) ** 2 + (y - py) ** 2)
                if r <= radius:
                    wgt = np.cos(r * np.pi / radius) * 0.5 + 0.5
                    for c in range(C):
                        v = feat[pid, c] * wgt
                        if reduce == "sum":
                            out[b, c, y, x] += v
                        elif out[b, c, y, x] < v:
                            out[b, c, y, x] = v
                            ids[b, c, y, x] = pid
    return out, ids


@pytest.mark.gpu
def test_hip_float64_known_answer_and_numpy(dev):
    """cuda/p2i_op/p2i_test.py:10-20 (a point at the image centre, R = 2, 8x8: 0.722008 on the four centre
    pixels, 0.104375 on the ring) and random float64 inputs against a numpy restatement."""
    from sparenet_amd.cuda.p2i_op import ext, p2i

    f64 = dict(dtype=torch.float64, device=dev)
    points = torch.zeros(1, 2, **f64)
    feats = torch.ones(1, 3, **f64)
    bi = torch.arange(1, dtype=torch.int32, device=dev)
    bg = torch.zeros(1, 3, 8, 8, **f64)
    for reduce in ("sum", "max"):
        out = p2i(points, feats, bi, bg, 2, "cos", reduce)
        assert out.dtype == torch.float64
        o = out.cpu().numpy()
        centre = np.cos(np.sqrt(0.5) * np.pi / 2) * 0.5 + 0.5           # 0.722008...
        ring = np.cos(np.sqrt(2.5) * np.pi / 2) * 0.5 + 0.5             # 0.104375...
        assert abs(centre - 0.722008) < 1e-6 and abs(ring - 0.104375) < 1e-6
        np.testing.assert_allclose(o[0, :, 3:5, 3:5], centre, rtol=1e-13)
        np.testing.assert_allclose(o[0, 0, 2, 3], ring, rtol=1e-12)
        assert o[0, 0, 0, 0] == 0 and np.count_nonzero(o[0, 0]) == 12
    rng = np.random.default_rng(3)
    B, n, C, S, R = 2, 60, 2, 12, 2.5
    pts = (rng.random((B * n, 2)) * 1.3 - 0.15) * (S - 1)
    ft = rng.standard_normal((B * n, C))
    bidx = rng.integers(-1, B + 1, B * n).astype(np.int32)
    bgn = rng.standard_normal((B, C, S, S)) * 0.1
    T = lambda a: torch.from_numpy(a).to(dev)
    out, ids = ext.p2i_max_forward_gpu(T(pts), T(ft), T(bidx), T(bgn), 0, R)
    ro, ri = _np_p2i_f64(pts, ft, bidx, bgn, R, "max")
    np.testing.assert_allclose(out.cpu().numpy(), ro, rtol=1e-13, atol=1e-15)
    assert np.array_equal(ids.cpu().numpy(), ri)
    so = ext.p2i_sum_forward_gpu(T(pts), T(ft), T(bidx), T(bgn), 0, R)
    np.testing.assert_allclose(so.cpu().numpy(), _np_p2i_f64(pts, ft, bidx, bgn, R, "sum")[0], rtol=1e-11, atol=1e-13)


@pytest.mark.gpu
def test_hip_float64_gradcheck_like_the_reference(dev):
    """The reference's own test (cuda/p2i_op/p2i_test.py:23-35): float64 gradcheck of p2i, sum and max."""
    from torch.autograd import gradcheck

    from sparenet_amd.cuda.p2i_op import p2i

    g = torch.Generator().manual_seed(0)
    for _ in range(3):
        points = torch.randn(2, 2, dtype=torch.float64, generator=g).to(dev).requires_grad_(True)
        feats = torch.randn(2, 3, dtype=torch.float64, generator=g).to(dev).requires_grad_(True)
        bi = torch.zeros(2, dtype=torch.int32, device=dev)
        bg = torch.randn(1, 3, 8, 8, dtype=torch.float64, generator=g).to(dev).requires_grad_(True)
        assert gradcheck(p2i, inputs=(points, feats, bi, bg, 2, "cos", "sum"))
        assert gradcheck(p2i, inputs=(points, feats, bi, bg, 2, "cos", "max"))


@pytest.mark.gpu
@pytest.mark.parametrize("projection,radii", [("orthorgonal", [5.0, 7.0, 10.0]), ("perspective", [3.0])])
def test_hip_all_views_in_one_pass_equal_per_view_calls(projection, radii, dev):
    """ComputeDepthMaps.forward_views: the V views of a sweep joined to the batch -- maps bit-equal to V
    separate forward() calls, and the gradient of the cloud equal to the sum of the per-view gradients."""
    from sparenet_amd.utils.p2i_utils import ComputeDepthMaps

    g = torch.Generator().manual_seed(8)
    cdm = ComputeDepthMaps(projection, 1.0, 64).to(dev)
    base = (torch.rand(3, 2000, 3, generator=g) - 0.5).to(dev)
    w = torch.rand(8, 3, len(radii), 64, 64, generator=g).to(dev)
    d1 = base.clone().requires_grad_(True)
    all_maps = cdm.forward_views(d1, range(8), radii)
    assert all_maps.shape == (8, 3, len(radii), 64, 64)
    (all_maps * w).sum().backward()
    d2 = base.clone().requires_grad_(True)
    loss = 0
    for v in range(8):
        m = cdm(d2, view_id=v, radius_list=radii)
        assert torch.equal(m, all_maps[v]), v
        loss = loss + (m * w[v]).sum()
    loss.backward()
    np.testing.assert_allclose(d1.grad.cpu().numpy(), d2.grad.cpu().numpy(), rtol=2e-5, atol=2e-6)
    sub = cdm.forward_views(base, [6, 2], radii)
    assert torch.equal(sub[0], all_maps[6]) and torch.equal(sub[1], all_maps[2])


@pytest.mark.gpu
def test_hip_binning_grouped_layout_and_its_fallbacks_agree(dev):
    """The one-launch binning (image b owns points [b*n, (b+1)*n)) is verified on the device, not assumed: the
    same cloud in grouped order, in shuffled order (same count per image, verification fails -> generic
    kernels), with one non-finite point (no cell -> generic kernels) and with unequal counts per image (never
    tried) must give the same maps, and ids that map back through the permutation."""
    from sparenet_amd.cuda.p2i_op import ext

    g = torch.Generator().manual_seed(77)
    B, n, S, radii = 4, 2048, 64, [5.0, 7.0, 10.0]
    pts = torch.rand(B * n, 2, generator=g) * (S + 6) - 3
    feat = torch.rand(B * n, 1, generator=g)
    bi = torch.arange(B, dtype=torch.int32).repeat_interleave(n)
    bg = torch.zeros(B, 1, S, S)
    out0, ids0 = ext.p2i_max_forward_multi_gpu(pts.to(dev), feat.to(dev), bi.to(dev), bg.to(dev), 0, radii)
    for r, R in enumerate(radii):
        o, i = oracle.p2i_max_forward(pts.numpy(), feat.numpy(), bi.numpy(), bg.numpy(), R)
        _close_maps(out0[r].cpu().numpy(), ids0[r].cpu().numpy(), o, i, f"grouped R={R}", pts, feat, bg, R)
    # shuffled: npoints % batch == 0 still holds, the layout check fails on the device
    perm = torch.randperm(B * n, generator=g)
    out1, ids1 = ext.p2i_max_forward_multi_gpu(pts[perm].to(dev), feat[perm].to(dev), bi[perm].to(dev), bg.to(dev), 0, radii)
    assert torch.equal(out1, out0)
    back = perm.to(dev)[ids1.clamp(min=0).long()]
    assert torch.equal(torch.where(ids1 < 0, ids1.long(), back), ids0.long())
    # one NaN point: it has no cell, so the grouped kernel hands over; the maps only lose that point
    pts2 = pts.clone()
    pts2[5] = float("nan")
    out2, ids2 = ext.p2i_max_forward_multi_gpu(pts2.to(dev), feat.to(dev), bi.to(dev), bg.to(dev), 0, radii)
    for r, R in enumerate(radii):
        o, i = oracle.p2i_max_forward(pts2.numpy(), feat.numpy(), bi.numpy(), bg.numpy(), R)
        _close_maps(out2[r].cpu().numpy(), ids2[r].cpu().numpy(), o, i, f"nan R={R}", pts2, feat, bg, R)
    # unequal counts per image
    keep = torch.ones(B * n, dtype=torch.bool)
    keep[:7] = False
    out3, ids3 = ext.p2i_max_forward_multi_gpu(pts[keep].to(dev), feat[keep].to(dev), bi[keep].to(dev), bg.to(dev), 0, radii)
    for r, R in enumerate(radii):
        o, i = oracle.p2i_max_forward(pts[keep].numpy(), feat[keep].numpy(), bi[keep].numpy(), bg.numpy(), R)
        _close_maps(out3[r].cpu().numpy(), ids3[r].cpu().numpy(), o, i, f"ragged R={R}", pts[keep], feat[keep], bg, R)


@pytest.mark.gpu
def test_hip_weight_series_error_bounds(dev):
    """The forward decides winners with an fp32 series of the cosine weight and only evaluates winners exactly;
    that is sound iff |series - exact| stays inside the band the kernel assumes (kWeightErr = 1.5e-6, of which
    1.1e-6 is the series' own share: 4e-7 is reserved for the roundings of u and of sqrtf).  Every fp32 u of a
    fine grid plus the neighbourhood of both ends; the backward's slope series likewise (5e-7)."""
    import ctypes
    from sparenet_amd import _lib as L

    u = np.concatenate([np.linspace(0.0, 1.0, 2_000_001, dtype=np.float64).astype(np.float32),
                        np.nextafter(np.float32(0), np.float32(1)) * np.arange(1, 1000, dtype=np.float32),
                        np.float32(1) - np.arange(0, 1000, dtype=np.float32) * np.float32(2.0 ** -24)])
    ut = torch.from_numpy(u).to(dev)
    w, sl = torch.empty_like(ut), torch.empty_like(ut)
    rc = L.lib().sn_p2i_series(L.fptr(ut, "u"), ctypes.c_int(ut.numel()), L.fptr(w, "w"), L.fptr(sl, "sl"),
                               ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    assert rc == 0
    t = np.sqrt(u.astype(np.float64))
    w_ref = np.cos(np.pi * t) * 0.5 + 0.5
    s_ref = np.where(t > 0, np.sin(np.pi * t) / (np.pi * np.maximum(t, 1e-300)), 1.0)
    assert np.abs(w.cpu().numpy().astype(np.float64) - w_ref).max() <= 1.1e-6
    assert np.abs(sl.cpu().numpy().astype(np.float64) - s_ref).max() <= 5e-7


def test_id_check_helper_mirrors_the_oracle_arithmetic():
    """tests/p2i_check.py evaluates a candidate's value with the oracle's arithmetic: on every covered pixel the
    value of the oracle's own winner is the oracle's output, bit for bit; a swapped winner is caught."""
    from p2i_check import _value, assert_ids_exact_up_to_ulp_ties

    rng = np.random.default_rng(5)
    B, n, S, R = 2, 1500, 48, 5.0
    pts = (rng.random((B * n, 2), dtype=np.float32) * (S - 1)).astype(np.float32)
    feat = rng.random((B * n, 1), dtype=np.float32)
    bi = np.repeat(np.arange(B, dtype=np.int32), n)
    bg = np.zeros((B, 1, S, S), np.float32)
    out, ids = oracle.p2i_max_forward(pts, feat, bi, bg, R)
    b, c, y, x = np.argwhere(ids >= 0).T
    v, r = _value(ids[b, c, y, x], c, y, x, pts, feat, R)
    assert np.array_equal(v.astype(np.float32), out[b, c, y, x]) and np.all(r <= R)
    assert assert_ids_exact_up_to_ulp_ties(ids, ids, pts, feat, bg, R) == 0
    wrong = ids.copy()
    k = np.flatnonzero(ids.ravel() >= 0)[:5]
    wrong.ravel()[k] = (wrong.ravel()[k] + 1) % (B * n)
    with pytest.raises(AssertionError):
        assert_ids_exact_up_to_ulp_ties(wrong, ids, pts, feat, bg, R)
