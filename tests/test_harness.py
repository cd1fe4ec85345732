"""the op-level reconstruction step (sparenet_amd/harness.py) runs end to end --
generator data flow (expansion -> MDS -> gather, twice), completion() loss with both metrics, autograd
through the whole chain -- and a few SGD steps on the surrogate parameters reduce the loss."""
import pytest
import torch


@pytest.mark.gpu
@pytest.mark.parametrize("metric", ["chamfer", "emd"])
def test_op_level_step_trains(metric, dev):
    from sparenet_amd.harness import Completion, SurrogateGenerator

    g = torch.Generator().manual_seed(2)
    B, N, M = 2, 2048, 384
    v = torch.randn(B, N, 3, generator=g)
    gt = 0.5 * v / v.norm(dim=2, keepdim=True)
    partial = gt[:, :M] + 1e-3 * torch.randn(B, M, 3, generator=g)
    init = gt + 0.03 * torch.randn(B, N, 3, generator=g)
    gen = SurrogateGenerator(B, N, n_primitives=4, init=init).to(dev)
    comp = Completion(metric).to(dev)
    opt = torch.optim.SGD(gen.parameters(), lr=200.0 if metric == "chamfer" else 20.0)
    losses = []
    for _ in range(4):
        loss, refine, middle, coarse, refine_loss, coarse_loss = comp(gen, partial.to(dev), gt.to(dev))
        assert refine.shape == middle.shape == coarse.shape == (B, N, 3)
        opt.zero_grad()
        loss.backward()
        for p in gen.parameters():
            assert p.grad is not None and torch.isfinite(p.grad).all()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(l == l for l in losses) and losses[-1] < losses[0], losses


@pytest.mark.gpu
def test_gan_step_runs_and_reaches_the_cloud(dev):
    """runners/sparenet_gan_runner.py:69-347 on surrogates: both optimisers step, the renderer carries
    gradient from the discriminator / image losses back to the generator parameters."""
    from sparenet_amd.harness import Completion, GanStep, SurrogateGenerator
    from sparenet_amd.networks import PatchDiscriminator

    g = torch.Generator().manual_seed(3)
    B, N, M, S = 2, 2048, 384, 64
    v = torch.randn(B, N, 3, generator=g)
    gt = 0.4 * v / v.norm(dim=2, keepdim=True)
    partial = gt[:, :M] + 1e-3 * torch.randn(B, M, 3, generator=g)
    gen = SurrogateGenerator(B, N, n_primitives=4, init=gt + 0.02 * torch.randn(B, N, 3, generator=g)).to(dev)
    disc = PatchDiscriminator((16, S, S)).to(dev)
    opt_g = torch.optim.Adam(gen.parameters(), lr=1e-4)
    opt_d = torch.optim.Adam(disc.parameters(), lr=1e-4)
    step = GanStep(gen, disc, Completion("chamfer").to(dev), opt_g, opt_d, radius_list=[2.0, 3.0],
                   image_size=S)
    d_before = [p.detach().clone() for p in disc.parameters()]
    g_before = gen.coarse.detach().clone()
    out = step(partial.to(dev), gt.to(dev))
    for k in ("rec_loss", "errG", "errG_D", "errD_real", "errD_fake"):
        assert torch.isfinite(out[k]).all(), k
    assert any(not torch.equal(a, b) for a, b in zip(d_before, disc.parameters()))
    assert not torch.equal(g_before, gen.coarse.detach())

    # the adversarial + image terms alone (no reconstruction loss) still produce a cloud gradient
    imgs = step._render_views(gen.coarse, 2.0)
    real = step._render_views(gt.to(dev), 2.0)
    val, feats = disc(torch.cat((real, imgs), dim=1), feat=True)
    assert val.shape == (B, 1) and [f.shape[1] for f in feats] == [16, 32, 64, 128]
    (val.mean() + torch.nn.functional.l1_loss(imgs, real)).backward()
    assert gen.coarse.grad is not None and gen.coarse.grad.abs().sum() > 0


@pytest.mark.gpu
def test_batched_emd_terms_equal_the_three_calls(dev):
    """harness.Completion with the EMD metric sends coarse / middle / refine through ONE auction call (a batch of 3 B
    clouds against the ground truth repeated): every cloud is its own auction, so the three terms, the loss and the
    gradients on the generator's parameters equal those of three separate calls bit for bit."""
    from sparenet_amd.harness import Completion, SurrogateGenerator

    g = torch.Generator().manual_seed(5)
    B, N, M = 3, 2048, 384
    v = torch.randn(B, N, 3, generator=g)
    gt = (0.5 * v / v.norm(dim=2, keepdim=True)).to(dev)
    partial = (gt[:, :M] + 1e-3 * torch.randn(B, M, 3, generator=g).to(dev)).contiguous()
    init = gt.cpu() + 0.05 * torch.randn(B, N, 3, generator=g)
    res = []
    for batch_terms in (True, False):
        gen = SurrogateGenerator(B, N, n_primitives=4, init=init).to(dev)
        with torch.no_grad():
            gen.refine1.delta.add_(0.01 * torch.randn(B, 3, N, generator=torch.Generator().manual_seed(9)).to(dev))
        comp = Completion("emd", batch_terms=batch_terms).to(dev)
        loss, refine, middle, coarse, refine_loss, coarse_loss = comp(gen, partial, gt)
        loss.backward()
        res.append((loss.detach(), refine_loss.detach(), coarse_loss.detach(),
                    [p.grad.detach().clone() for p in gen.parameters()]))
    (l1, r1, c1, g1), (l2, r2, c2, g2) = res
    assert torch.equal(l1, l2) and torch.equal(r1, r2) and torch.equal(c1, c2)
    assert all(torch.equal(a, b) for a, b in zip(g1, g2))


def test_batched_emd_terms_fall_back_above_512_clouds():
    """emdModule takes at most 512 clouds per call (cuda/emd/emd_module.py:38): three terms of a per-rank batch above
    170 clouds go back to one auction call per term instead of failing the assert (advisor, round 5)."""
    from sparenet_amd.harness import Completion

    calls = []

    class FakeEmd(torch.nn.Module):
        def forward(self, a, b, eps, iters):
            assert a.shape[0] <= 512 and a.shape == b.shape
            calls.append(a.shape[0])
            return (a - b).pow(2).sum(-1), None

    comp = Completion("emd")
    comp.emd_dist = FakeEmd()
    for b, expect in ((170, [510]), (171, [171, 171, 171]), (512, [512, 512, 512])):
        calls.clear()
        gt = torch.rand(b, 8, 3)
        clouds = [torch.rand(b, 8, 3) for _ in range(3)]
        terms = comp._metrics(clouds, gt)
        assert calls == expect and len(terms) == 3
        for c, t in zip(clouds, terms):
            assert torch.equal(t, comp._metric(c, gt))
