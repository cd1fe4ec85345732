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
