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
