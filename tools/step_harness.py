"""Times the op-level reconstruction step (sparenet_amd/harness.py) at the reference's sizes:
B=32, 16384 output points, 3000 input points, n_primitives 32."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparenet_amd.harness import Completion, GanStep, NetworkGenerator, SurrogateDiscriminator, SurrogateGenerator

dev = torch.device("cuda:0")
B, N, M = 32, 16384, 3000
g = torch.Generator().manual_seed(0)
# surface-like ground truth (a sphere), a partial view of it, a noisy start for the decoder surrogate
v = torch.randn(B, N, 3, generator=g); gt = (0.5 * v / v.norm(dim=2, keepdim=True))
key = (torch.atan2(gt[..., 1], gt[..., 0]) * 4).floor() * 100 + (gt[..., 2] * 8).floor()
gt = torch.gather(gt, 1, key.argsort(dim=1).unsqueeze(-1).expand(-1, -1, 3)).contiguous()
# (not an exact subset of gt: a resampled point that coincides with its EMD match has distance 0 and
# d sqrt(dist) is infinite there -- in the reference's loss too)
partial = (gt[:, torch.randperm(N, generator=g)[:M]] + 1e-3 * torch.randn(B, M, 3, generator=g)).contiguous()
init = gt + 0.01 * torch.randn(B, N, 3, generator=g)
for metric in ("chamfer", "emd"):
    gen = SurrogateGenerator(B, N, 32, init=init).to(dev)
    comp = Completion(metric).to(dev)
    opt = torch.optim.SGD(gen.parameters(), lr=0.1)
    part_d, gt_d = partial.to(dev), gt.to(dev)
    def step():
        loss, *_ = comp(gen, part_d, gt_d)
        opt.zero_grad(); loss.backward(); opt.step()
        return loss
    first = float(step()); torch.cuda.synchronize()
    t0 = time.perf_counter(); K = 3
    for _ in range(K): last = step()
    torch.cuda.synchronize()
    print(f"metric={metric}: {(time.perf_counter() - t0) / K * 1e3:.1f} ms per op-level step "
          f"(loss {first:.5f} -> {float(last):.5f})")

# config 5: the GAN step (completion + 3 clouds x 8 views rendered + discriminator twice + both updates)
gen = SurrogateGenerator(B, N, 32, init=init).to(dev)
disc = SurrogateDiscriminator().to(dev)
gan = GanStep(gen, disc, Completion("chamfer").to(dev), torch.optim.Adam(gen.parameters(), lr=1e-4),
              torch.optim.Adam(disc.parameters(), lr=1e-4))
part_d, gt_d = partial.to(dev), gt.to(dev)
out = gan(part_d, gt_d); torch.cuda.synchronize()
t0 = time.perf_counter(); K = 3
for _ in range(K): out = gan(part_d, gt_d)
torch.cuda.synchronize()
print(f"gan step (chamfer metric): {(time.perf_counter() - t0) / K * 1e3:.1f} ms "
      f"(errG {float(out['errG']):.4f} errD {float(out['errD_real'] + out['errD_fake']):.4f})")

# the same reconstruction step with networks in it: EdgeConv encoder at the reference's widths (hide 4096: k-NN
# graphs on 3 / 256 / 256 / 512 channels, k = 8), folding decoder, residual refiners; bf16 autocast
torch.manual_seed(0)
net = NetworkGenerator(num_points=N, n_primitives=32, hide_size=4096, feature_size=4096).to(dev)
comp = Completion("chamfer").to(dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-4)
def nstep():
    loss, *_ = comp(net, part_d, gt_d)
    opt.zero_grad(); loss.backward(); opt.step()
    return loss
nstep(); torch.cuda.synchronize()
t0 = time.perf_counter(); K = 3
for _ in range(K): last = nstep()
torch.cuda.synchronize()
print(f"network generator (EdgeConv encoder + decoder + refiners, bf16), chamfer metric: "
      f"{(time.perf_counter() - t0) / K * 1e3:.1f} ms per step, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
