"""Op-level restatement of SpareNet's reconstruction step.

The data flow of `SpareNetGenerator.forward` (models/sparenet_generator.py:63-82), of
`SpareNetRefine.forward` (:558-579) and of `SpareNetRunner.completion` (runners/sparenet_runner.py:67-108)
with the learned networks replaced by surrogates that have the same tensor shapes at the op
boundaries: the decoder is a learnable cloud, the residual network a learnable offset field.  Every op
of the hot path appears where the reference calls it:

    coarse [B,N,3]                                         (decoder output, here a parameter)
    refine(inps, partial, coarse):                         x2 ("middle", then "refine")
        dist, _, mml = expansion(coarse, N // n_primitives, 1.5)
        base  = cat([inps | 0], [partial | 1]) along points          [B,4,N+M]
        idx   = minimum_density_sample(base xyz, N, mml)
        base  = gather_operation(base, idx)                          [B,4,N]
        out   = base[:, :3] + residual(base)              (here a learnable delta)
    loss = metric(coarse, gt) + metric(middle, gt) + metric(refine, gt) + 0.1 mean(expansion)
           [+ 0.5 mean(dist1(refine, gt)) with use_consist_loss],  metric in {chamfer, emd}

It exists to measure what the ops cost inside the real step (MDS and the expansion penalty run inside the
generator, not in the loss) and to exercise autograd through the whole chain.

`GanStep` adds the adversarial half (runners/sparenet_gan_runner.py:69-113 train_step, :186-266 discriminator
update, :268-347 generator update): the middle cloud, the ground truth and the partial input are rendered from
all 8 views at one radius drawn from the list, the 8+8 maps go through a discriminator
(sparenet_amd.networks.PatchDiscriminator), LSGAN targets, feature matching weighted by channel count, L1 image
matching, and errG = 200 rec + 0.1 gan + fm + im (configs/base_config.py:67-73).

The learned networks themselves (EdgeConv encoder, style-based folding decoder, residual refiners,
discriminators) and the objectives as plain functions live in sparenet_amd/networks.py; `Completion` and
`GanStep` accept either generator (`SurrogateGenerator` here for op-level timing, `networks.Generator` for
BASELINE configs 4-5).
"""
import random

import torch

from sparenet_amd import _lib
from sparenet_amd.utils.p2i_utils import N_VIEWS_PREDEFINED, ComputeDepthMaps

from sparenet_amd.cuda.chamfer_distance import ChamferDistance, ChamferDistanceMean
from sparenet_amd.cuda.emd.emd_module import emdModule
from sparenet_amd.cuda.expansion_penalty.expansion_penalty_module import expansionPenaltyModule
from sparenet_amd.cuda.MDS import MDS_module
from sparenet_amd.networks import (completion_loss, discriminator_objective, emd_term, feature_matching,
                                   generator_objective)


class SurrogateRefine(torch.nn.Module):
    """SpareNetRefine with the residual network replaced by a learnable offset per output point."""

    def __init__(self, batch, num_points, n_primitives=32):
        super().__init__()
        self.num_points, self.n_primitives = num_points, n_primitives
        self.expansion = expansionPenaltyModule()
        self.delta = torch.nn.Parameter(torch.zeros(batch, 3, num_points))

    def forward(self, inps, partial, coarse):
        dist, _, mean_mst_dis = self.expansion(coarse, self.num_points // self.n_primitives, 1.5)
        loss_mst = torch.mean(dist)
        id0 = torch.zeros(inps.shape[0], 1, inps.shape[2], device=inps.device)
        id1 = torch.ones(partial.shape[0], 1, partial.shape[2], device=partial.device)
        base = torch.cat((torch.cat((inps, id0), 1), torch.cat((partial, id1), 1)), 2)   # [B,4,N+M]
        idx = MDS_module.minimum_density_sample(base[:, 0:3, :].transpose(1, 2).contiguous(),
                                                coarse.shape[1], mean_mst_dis)
        base = MDS_module.gather_operation(base.contiguous(), idx)
        outs = base[:, 0:3, :] + self.delta
        return outs.transpose(2, 1).contiguous(), loss_mst


class SurrogateGenerator(torch.nn.Module):
    """coarse -> middle -> refine like SpareNetGenerator.forward, decoder = a learnable cloud."""

    def __init__(self, batch, num_points=16384, n_primitives=32, init=None):
        super().__init__()
        start = init if init is not None else torch.rand(batch, num_points, 3) - 0.5
        self.coarse = torch.nn.Parameter(start.clone())
        self.refine1 = SurrogateRefine(batch, num_points, n_primitives)
        self.refine2 = SurrogateRefine(batch, num_points, n_primitives)

    def forward(self, partial):
        return self.forward_staged(partial, lambda cloud: None)

    def forward_staged(self, partial, on_cloud):
        """forward(), calling on_cloud(cloud) as soon as `coarse` and `middle` are final (Completion issues their
        losses on a second stream while the next refine stage samples)."""
        coarse = self.coarse
        part = partial.transpose(1, 2).contiguous()                      # [B,3,M]
        on_cloud(coarse)
        middle, loss_mst = self.refine1(coarse.transpose(1, 2).contiguous(), part, coarse)
        on_cloud(middle)
        refine, _ = self.refine2(middle.transpose(1, 2).contiguous(), part, middle)
        return coarse, middle, refine, loss_mst


class Completion(torch.nn.Module):
    """runners/sparenet_runner.py:67-108."""

    def __init__(self, metric="chamfer", use_consist_loss=True, overlap=True, batch_terms=True):
        """overlap: Chamfer metric only -- the loss of a finished cloud on a second stream while the next refine stage
        samples (no effect with the EMD metric, see forward); batch_terms: EMD metric -- the three terms through one
        auction call (see _metrics)."""
        super().__init__()
        self.overlap, self._side, self.batch_terms = overlap, None, batch_terms
        if metric not in ("chamfer", "emd"):
            raise Exception("unknown training metric")
        self.metric, self.use_consist_loss = metric, use_consist_loss
        self.chamfer_dist = ChamferDistance()
        self.chamfer_dist_mean = ChamferDistanceMean()
        self.emd_dist = emdModule()

    def _metric(self, cloud, gt):
        if self.metric == "chamfer":
            return self.chamfer_dist_mean(cloud, gt).mean()
        dist, _ = self.emd_dist(cloud, gt, eps=0.005, iters=50)
        return emd_term(dist)

    def _metrics(self, clouds, gt):
        """The metric of every cloud set against `gt`, one value per set.  EMD: the sets go through ONE auction call
        as a batch of len(clouds) x B clouds (the three terms of runners/sparenet_runner.py:91-93 feed one loss; every
        cloud of the batch is its own auction, so dist / assignment / gradients are those of the separate calls bit
        for bit) -- one persistent launch instead of three: 3 x 1.06 ms -> 1.3 ms at 4 clouds per rank on a trained
        generator's clouds, 3 x 8.0 -> ~12 ms on an untrained one's."""
        b = gt.shape[0]
        # emdModule takes at most 512 clouds per call (cuda/emd/emd_module.py:38): a per-rank batch above 512 / 3
        # goes back to one call per term.  The batched call's workspace and its ground-truth copy are len(clouds) x
        # those of one term (gt.repeat: 3 x 6.3 MB at 32 clouds of 16384 points; the workspace 3 x ~59 MB).
        if self.metric == "chamfer" or len(clouds) == 1 or not self.batch_terms or len(clouds) * b > 512:
            return [self._metric(c, gt) for c in clouds]
        dist, _ = self.emd_dist(torch.cat(clouds, 0), gt.repeat(len(clouds), 1, 1), eps=0.005, iters=50)
        return [emd_term(dist[i * b:(i + 1) * b]) for i in range(len(clouds))]

    def forward(self, generator, partial, gt):
        _lib.device_check("Completion")   # a team time-out of an EARLIER step raises here, at the latest (see loss_item)
        # Generators that can hand their clouds over as they become final (SurrogateGenerator, networks.Generator --
        # exactly these two: calling forward_staged() goes around nn.Module.__call__, i.e. around forward hooks and
        # wrappers such as DistributedDataParallel or torch.compile, which therefore take the plain path below).
        # Chamfer metric only: the EMD auction is one persistent launch that needs EVERY compute unit (a team of 32
        # workgroups per XCD), so beside the sampler -- one workgroup per cloud for ~18 ms -- its teams spin until the
        # sampler's CUs are free: measured, rocprofv3 shows 17.7 ms per auction launch instead of 1.2 and the step
        # does not move (config 4: 141 ms either way, profiles/r04_c_network_config4_steady.txt)
        from sparenet_amd.networks import Generator as _NetGenerator
        if (self.overlap and self.metric == "chamfer" and partial.is_cuda
                and type(generator) in (SurrogateGenerator, _NetGenerator) and not generator._forward_hooks
                and not generator._forward_pre_hooks):
            return self._forward_overlapped(generator, partial, gt)
        coarse, middle, refine, expansion_penalty = generator(partial)
        coarse_loss, middle_loss, refine_loss = self._metrics([coarse, middle, refine], gt)
        return self._compose(coarse, middle, refine, expansion_penalty, coarse_loss, middle_loss, refine_loss, gt)

    def _forward_overlapped(self, generator, partial, gt):
        """Same values, different issue order: minimum density sampling runs one workgroup per cloud (32 of
        256 CUs at B = 32) for tens of milliseconds, so the loss of the cloud that is already final is
        issued on a second HIP stream while the next refine stage samples."""
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream()
        side = self._side
        early = []
        # tensors that cross streams are registered with the caching allocator (record_stream): without it
        # a block freed on its own stream may be handed out again while the other stream still reads it
        gt.record_stream(side)

        def on_cloud(cloud):
            cloud.record_stream(side)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                loss = self._metric(cloud, gt)
            loss.record_stream(main)
            early.append(loss)

        coarse, middle, refine, expansion_penalty = generator.forward_staged(partial, on_cloud)
        if len(early) == 2:
            coarse_loss, middle_loss = early
        else:   # a generator without refine stages hands nothing over early
            coarse_loss, middle_loss = self._metric(coarse, gt), self._metric(middle, gt)
        refine_loss = self._metric(refine, gt)
        main.wait_stream(side)
        return self._compose(coarse, middle, refine, expansion_penalty, coarse_loss, middle_loss, refine_loss, gt)

    def _compose(self, coarse, middle, refine, expansion_penalty, coarse_loss, middle_loss, refine_loss, gt):
        dist1 = self.chamfer_dist(refine, gt)[0] if self.use_consist_loss else None
        loss = completion_loss(coarse_loss, middle_loss, refine_loss, expansion_penalty, dist1)
        return loss, refine, middle, coarse, refine_loss, coarse_loss


class GanStep:
    """One training step of SpareNetGANRunner on a generator with SpareNetGenerator's outputs."""

    def __init__(self, generator, discriminator, completion, opt_g, opt_d, radius_list=(5.0, 7.0, 10.0),
                 image_size=256, projection="orthorgonal", use_fm=True, use_im=True, weight_l2=200.0,
                 weight_gan=0.1, weight_fm=1.0, weight_im=1.0, seed=0):
        self.generator, self.discriminator, self.completion = generator, discriminator, completion
        self.opt_g, self.opt_d = opt_g, opt_d
        self.radius_list = list(radius_list)
        self.use_fm, self.use_im = use_fm, use_im
        self.weight_l2, self.weight_gan, self.weight_fm, self.weight_im = weight_l2, weight_gan, weight_fm, weight_im
        self.renderer = ComputeDepthMaps(projection, 1.0, image_size)
        self.rng = random.Random(seed)

    def _render_views(self, cloud, radius):
        """[B, views, S, S]: one single-radius map per predefined view (runners/sparenet_gan_runner.py:217-225 loops
        over the views and concatenates on dim 1).  On the GPU all views are ONE pass of the renderer
        (ComputeDepthMaps.forward_views: the views join the batch; slice v bit-equal to the per-view call, the depth
        normalisation stays per view and per cloud set) instead of 8 x ~10 short launches."""
        if cloud.is_cuda and cloud.dtype == torch.float32:
            maps = self.renderer.forward_views(cloud, range(N_VIEWS_PREDEFINED), [radius])   # [V,B,1,S,S]
            return maps[:, :, 0].permute(1, 0, 2, 3)
        return torch.cat([self.renderer(cloud, view_id=v, radius_list=[radius])
                          for v in range(N_VIEWS_PREDEFINED)], dim=1)

    def __call__(self, partial, gt):
        _lib.device_check("GanStep")
        batch = partial.shape[0]
        real_label = torch.ones(batch, 1, device=partial.device)
        fake_label = torch.zeros(batch, 1, device=partial.device)
        self.renderer.to(partial.device)

        rec_loss, _, middle, _, refine_loss, coarse_loss = self.completion(self.generator, partial, gt)

        # ---- discriminator update (detached images)
        self.opt_d.zero_grad()
        radius = self.rng.sample(self.radius_list, 1)[0]
        real_imgs = self._render_views(gt, radius)
        fake_imgs = self._render_views(middle, radius)
        input_imgs = self._render_views(partial, radius)
        d_real = self.discriminator(torch.cat((input_imgs, real_imgs), dim=1).detach())
        d_fake = self.discriminator(torch.cat((input_imgs, fake_imgs), dim=1).detach())
        err_d_real, err_d_fake = discriminator_objective(d_real, d_fake, real_label, fake_label)
        (err_d_real + err_d_fake).backward()
        self.opt_d.step()

        # ---- generator update: gradients reach the cloud through the renderer
        self.opt_g.zero_grad()
        loss_fm = loss_im = None
        if self.use_fm:
            d_fake, fake_feats = self.discriminator(torch.cat((input_imgs, fake_imgs), dim=1), feat=True)
            _, real_feats = self.discriminator(torch.cat((input_imgs, real_imgs), dim=1), feat=True)
            loss_fm = feature_matching(fake_feats, real_feats)
        else:
            d_fake = self.discriminator(torch.cat((input_imgs, fake_imgs), dim=1))
        if self.use_im:
            loss_im = torch.nn.functional.l1_loss(fake_imgs, real_imgs.detach())
        err_g, err_g_d = generator_objective(rec_loss, d_fake, real_label, loss_fm, loss_im, self.weight_l2,
                                             self.weight_gan, self.weight_fm, self.weight_im)
        err_g.backward()
        self.opt_g.step()
        return dict(rec_loss=rec_loss.detach(), errG=err_g.detach(), errG_D=err_g_d.detach(),
                    errD_real=err_d_real.detach(), errD_fake=err_d_fake.detach(),
                    coarse_loss=coarse_loss.detach() * 1000, refine_loss=refine_loss.detach() * 1000)
