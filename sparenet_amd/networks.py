"""The learned networks around the hot path (BASELINE configs 4-5), restated for MI355X.

Architecture facts (layer widths, what feeds what) follow the reference; the code is this repository's own and
is organised for the hardware: the 32 per-primitive folding decoders are ONE set of batched GEMMs instead of
32 sequential small networks, convolutions / linears run under bf16 autocast, every custom op stays fp32 HIP.

    Generator             models/sparenet_generator.py:12-82   encoder -> style decoder -> refine x2 (shared)
      EdgeConvEncoder     :104-120 (SpareNetEncode) + :122-260 (EdgeConvResFeat, optional SELayer :700-722)
      StyleFoldingDecoder :300-420 (SpareNetDecode, use_AdaIn="share"), GridDecoder :960-1062,
                          AdaptiveInstanceNorm1d :908-957, grid_generation :750-770
      RefineStage         :530-579 (SpareNetRefine: expansion penalty -> MDS -> gather -> residual)
      PointNetResidual    :582-650 (PointNetRes)
    PatchDiscriminator    models/sparenet_discriminator.py:13-81  (spectral norm, 6 stride-2 blocks)
    ProjectionDiscriminator :84-153
    completion_loss       runners/sparenet_runner.py:83-108
    generator_objective   runners/sparenet_gan_runner.py:284-347, discriminator_objective :243-266

Parameter counts equal the reference's modules (checked by instantiating them in the build container):
encoder 23,156,224; decoder 31,489,542 (style MLP) + 665,874 per primitive; refine 867,139 (the reference also
carries an unused BatchNorm1d(3) and an unused Conv1d(3,64): 6 + 256 parameters that never receive a
gradient and are left out so that DistributedDataParallel needs no unused-parameter search);
PatchDiscriminator 2,805,168 trainable (+ 13,809 power-iteration vectors, buffers here).
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from sparenet_amd.cuda.expansion_penalty.expansion_penalty_module import expansionPenaltyModule
from sparenet_amd.cuda.MDS import MDS_module


AUTOCAST = True   # bf16 autocast around the convolutions / linears on the GPU (tests switch it off to compare in fp32)


def _autocast(t):
    return torch.autocast(t.device.type, dtype=torch.bfloat16, enabled=t.is_cuda and AUTOCAST)


def knn_indices(x, k):
    """x [B,C,N] -> [B,N,k] neighbour indices in feature space.  GPU: the fused MFMA search (sn_knn); CPU: the
    formula of the reference's own CPU branch (models/sparenet_generator.py:872-875)."""
    if x.is_cuda:
        from sparenet_amd.cuda.knn import knn
        return knn(x, k)
    inner = -2 * torch.matmul(x.transpose(2, 1), x)
    xx = torch.sum(x ** 2, dim=1, keepdim=True)
    return (-xx - inner - xx.transpose(2, 1)).topk(k=k, dim=-1)[1]


def graph_feature(x, k):
    """[B,C,N] -> edge features [B,2C,N,k] = (neighbour - centre, centre)  (models/sparenet_generator.py:880-906)."""
    if x.is_cuda:
        from sparenet_amd.cuda.knn import get_graph_feature
        return get_graph_feature(x.float(), k=k)
    idx = knn_indices(x, k)
    b, c, n = x.shape
    nb = torch.gather(x.unsqueeze(2).expand(b, c, n, n), 3, idx.unsqueeze(1).expand(b, c, n, k))
    own = x.unsqueeze(3).expand(b, c, n, k)
    return torch.cat((nb - own, own), dim=1)


class SqueezeExcite(nn.Module):
    """Channel gate: x * sigmoid(W2 relu(W1 mean(x)))  (SELayer / SELayer1D, reduction 16)."""

    def __init__(self, channels, reduction=16):
        super().__init__()
        self.fc = nn.Sequential(nn.Linear(channels, channels // reduction, bias=False), nn.ReLU(inplace=True),
                                nn.Linear(channels // reduction, channels, bias=False), nn.Sigmoid())

    def forward(self, x):
        gate = self.fc(x.flatten(2).mean(dim=2))
        return x * gate.view(*gate.shape, *([1] * (x.dim() - 2)))


class PrimitiveSqueezeExcite(nn.Module):
    """One SqueezeExcite per primitive (the reference's GridDecoder carries its own se1..3,
    models/sparenet_generator.py:1036-1040), evaluated for all P primitives at once: weights stacked [P, C/r, C] and
    [P, C, C/r], input [B, P, C, n]."""

    def __init__(self, primitives, channels, reduction=16):
        super().__init__()
        r = channels // reduction
        self.w1 = nn.Parameter(torch.empty(primitives, r, channels))
        self.w2 = nn.Parameter(torch.empty(primitives, channels, r))
        for w, fan_in in ((self.w1, channels), (self.w2, r)):      # nn.Linear's default initialisation
            if w.numel():                                           # channels < reduction: an empty bottleneck
                nn.init.uniform_(w, -1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in))

    def forward(self, h):
        z = F.relu(torch.einsum("bpc,prc->bpr", h.mean(dim=-1), self.w1.to(h.dtype)))
        gate = torch.sigmoid(torch.einsum("bpr,pcr->bpc", z, self.w2.to(h.dtype)))
        return h * gate.unsqueeze(-1)


class EdgeConvEncoder(nn.Module):
    """partial cloud [B,3,M] -> style [B, bottleneck]: four EdgeConv stages (k-NN graph in FEATURE space on 3,
    h/16, h/16, h/8 channels; 1x1 conv on the [B,2C,M,k] edge features; BatchNorm; optional squeeze-excite;
    LeakyReLU 0.2; max over the neighbours; 1x1 residual branches), a 1x1 conv over the concatenated stages,
    global max + mean pooling, then Linear + BatchNorm + ReLU."""

    def __init__(self, hide_size=4096, output_size=4096, bottleneck_size=4096, k=8, use_se=False):
        super().__init__()
        h = hide_size
        self.k = k
        dims = [(6, h // 16), (h // 8, h // 16), (h // 8, h // 8), (h // 4, h // 4)]
        self.edge = nn.ModuleList(nn.Conv2d(i, o, 1, bias=False) for i, o in dims)
        self.norm = nn.ModuleList(nn.BatchNorm2d(o) for _, o in dims)
        self.gate = nn.ModuleList(SqueezeExcite(o) if use_se else nn.Identity() for _, o in dims)
        self.res = nn.ModuleList([nn.Conv1d(h // 16, h // 16, 1, bias=False),
                                  nn.Conv1d(h // 16, h // 8, 1, bias=False),
                                  nn.Conv1d(h // 8, h // 4, 1, bias=False)])
        self.head = nn.Conv1d(h // 2, output_size // 2, 1, bias=False)
        self.head_norm = nn.BatchNorm1d(output_size // 2)
        self.linear = nn.Linear(output_size, bottleneck_size)
        self.bn = nn.BatchNorm1d(bottleneck_size)

    def forward(self, x):
        feats, cur = [], x
        for i in range(4):
            edges = graph_feature(cur.float(), self.k)                        # fp32 graph ops
            with _autocast(x):
                y = F.leaky_relu(self.gate[i](self.norm[i](self.edge[i](edges))), 0.2).amax(dim=-1)
                if i > 0:
                    y = y + self.res[i - 1](cur)
            cur = y
            feats.append(y)
        with _autocast(x):
            z = F.leaky_relu(self.head_norm(self.head(torch.cat(feats, dim=1))), 0.2)
            pooled = torch.cat((z.amax(dim=2), z.mean(dim=2)), dim=1)         # [B, output_size]
            return F.relu(self.bn(self.linear(pooled))).float()


def folding_grid(points_per_primitive):
    """The 2-D lattice every primitive folds: (i / gx, j / gy), gx = 2^floor(log2(n)/2) - 1, gy = 2^ceil(log2(n)/2) - 1,
    rescaled to [-1, 1]  (grid_generation + the (g - 0.5) * 2 of SpareNetDecode.forward)."""
    lg = math.log2(points_per_primitive)
    gx, gy = 2 ** math.floor(lg / 2) - 1, 2 ** math.ceil(lg / 2) - 1
    i = torch.arange(gx + 1, dtype=torch.float32) / max(gx, 1)
    j = torch.arange(gy + 1, dtype=torch.float32) / max(gy, 1)
    grid = torch.stack(torch.meshgrid(i, j, indexing="ij"), 0).reshape(2, -1)
    return (grid - 0.5) * 2


def _instance_norm(x, eps=1e-5):
    """Per (sample, channel) normalisation over the points, biased variance (AdaptiveInstanceNorm1d's
    F.batch_norm on the [1, B*C, n] view)."""
    xf = x.float()
    mean = xf.mean(dim=-1, keepdim=True)
    var = xf.var(dim=-1, unbiased=False, keepdim=True)
    return (xf - mean) * torch.rsqrt(var + eps)


class StyleFoldingDecoder(nn.Module):
    """style [B,S] -> coarse cloud [B,3,P*n].  P primitives, each its own 4-layer 1x1-conv network
    2 -> W -> W/2 -> W/4 -> 3 on the SAME lattice, each hidden layer followed by AdaIN (instance norm, then
    scale / shift produced from the style by one shared MLP), BatchNorm and ReLU; tanh at the end.
    The P networks have identical shapes, so their weights are stacked and every layer is one batched GEMM
    over (sample, primitive) instead of P small launches; the first layer and its instance norm do not depend
    on the sample at all (the lattice is a constant) and are evaluated once per call for all samples."""

    def __init__(self, num_points=16384, n_primitives=32, style_dim=4096, width=1026, use_se=False):
        super().__init__()
        self.P, self.n = n_primitives, num_points // n_primitives
        self.widths = [width, width // 2, width // 4]
        w1, w2, w3 = self.widths
        self.register_buffer("grid", folding_grid(self.n), persistent=False)            # [2, n]
        assert self.grid.shape[1] == self.n, "points per primitive must be a power of two"
        self.mlp = nn.Sequential(nn.Linear(style_dim, style_dim), nn.ReLU(),
                                 nn.Linear(style_dim, 2 * (w1 + w2 + w3)))
        dims = [(2, w1), (w1, w2), (w2, w3), (w3, 3)]
        self.weight = nn.ParameterList(nn.Parameter(torch.empty(self.P, o, i)) for i, o in dims)
        self.bias = nn.ParameterList(nn.Parameter(torch.empty(self.P, o)) for i, o in dims)
        for w, b, (i, _) in zip(self.weight, self.bias, dims):      # nn.Conv1d's default initialisation
            bound = 1.0 / math.sqrt(i)
            nn.init.uniform_(w, -bound, bound)
            nn.init.uniform_(b, -bound, bound)
        self.bn = nn.ModuleList(nn.BatchNorm1d(self.P * w) for w in self.widths)         # per primitive and channel
        self.gate = nn.ModuleList(PrimitiveSqueezeExcite(self.P, w) if use_se else nn.Identity() for w in self.widths)

    def _post(self, h, layer, scale, shift):
        """AdaIN affine -> BatchNorm -> (gate) -> ReLU on [B,P,C,n]."""
        b, p, c, n = h.shape
        h = h * scale.view(b, 1, c, 1) + shift.view(b, 1, c, 1)
        h = self.bn[layer](h.reshape(b, p * c, n)).view(b, p, c, n)
        return F.relu(self.gate[layer](h))

    def forward(self, style):
        b = style.shape[0]
        w1, w2, w3 = self.widths
        with _autocast(style):
            params = self.mlp(style).float()                                            # [B, 2 (w1+w2+w3)]
        # assign_adain_params: per layer first the shift ("mean" -> bias), then the scale ("std" -> weight)
        sh1, sc1, sh2, sc2, sh3, sc3 = torch.split(params, [w1, w1, w2, w2, w3, w3], dim=1)
        # layer 1: sample independent up to the AdaIN affine
        h = torch.einsum("poi,in->pon", self.weight[0].float(), self.grid) + self.bias[0].float().unsqueeze(-1)
        h = _instance_norm(h).unsqueeze(0).expand(b, -1, -1, -1)                         # [B,P,w1,n]
        with _autocast(style):
            h = self._post(h, 0, sc1, sh1)
            h = torch.matmul(self.weight[1].unsqueeze(0), h) + self.bias[1].view(1, self.P, -1, 1)
            h = self._post(_instance_norm(h), 1, sc2, sh2)
            h = torch.matmul(self.weight[2].unsqueeze(0), h) + self.bias[2].view(1, self.P, -1, 1)
            h = self._post(_instance_norm(h), 2, sc3, sh3)
            out = torch.tanh(torch.matmul(self.weight[3].unsqueeze(0), h) + self.bias[3].view(1, self.P, -1, 1))
        # [B,P,3,n] -> [B,3,P*n], primitives one after the other along the points (torch.cat(outs, 2))
        return out.float().permute(0, 2, 1, 3).reshape(b, 3, self.P * self.n)


class PointNetResidual(nn.Module):
    """[B,4,N] (xyz + source flag) -> offsets [B,3,N] in (-1,1): per-point 64/128/1024 features, the global
    max concatenated back to the 64-wide point features, 512/256/128/3."""

    def __init__(self, use_se=False):
        super().__init__()
        mk = lambda i, o: nn.Sequential(nn.Conv1d(i, o, 1), nn.BatchNorm1d(o))
        self.l1, self.l2, self.l3 = mk(4, 64), mk(64, 128), mk(128, 1024)
        self.l4, self.l5, self.l6 = mk(1088, 512), mk(512, 256), mk(256, 128)
        self.out = nn.Conv1d(128, 3, 1)
        g = lambda c: SqueezeExcite(c) if use_se else nn.Identity()
        self.g1, self.g2, self.g4, self.g5, self.g6 = g(64), g(128), g(512), g(256), g(128)

    def forward(self, x):
        with _autocast(x):
            point = F.relu(self.g1(self.l1(x)))
            y = F.relu(self.g2(self.l2(point)))
            glob = self.l3(y).amax(dim=2, keepdim=True).expand(-1, -1, x.shape[2])
            y = torch.cat((glob, point), dim=1)
            y = F.relu(self.g4(self.l4(y)))
            y = F.relu(self.g5(self.l5(y)))
            y = F.relu(self.g6(self.l6(y)))
            return torch.tanh(self.out(y)).float()


class RefineStage(nn.Module):
    """The hot-path ops inside the generator: expansion penalty on the incoming cloud, the cloud and the
    partial input merged (flag channel 0 / 1), minimum density sampling back to N points with the penalty's own
    mean MST length, gather, residual offsets."""

    def __init__(self, num_points=16384, n_primitives=32, use_se=False):
        super().__init__()
        self.num_points, self.n_primitives = num_points, n_primitives
        self.expansion = expansionPenaltyModule()
        self.residual = PointNetResidual(use_se)

    def forward(self, inps, partial, coarse):
        dist, _, mean_mst = self.expansion(coarse, self.num_points // self.n_primitives, 1.5)
        loss_mst = torch.mean(dist)
        id0 = torch.zeros(inps.shape[0], 1, inps.shape[2], device=inps.device)
        id1 = torch.ones(partial.shape[0], 1, partial.shape[2], device=partial.device)
        base = torch.cat((torch.cat((inps, id0), 1), torch.cat((partial, id1), 1)), 2).contiguous()   # [B,4,N+M]
        idx = MDS_module.minimum_density_sample(base[:, 0:3, :].transpose(1, 2).contiguous(), coarse.shape[1],
                                                mean_mst)
        base = MDS_module.gather_operation(base, idx)
        outs = base[:, 0:3, :] + self.residual(base)
        return outs.transpose(2, 1).contiguous(), loss_mst


class Generator(nn.Module):
    """partial [B,M,3] -> (coarse, middle, refine [B,N,3], expansion loss): SpareNetGenerator.forward with the
    Residualnet encoder and the shared-AdaIN decoder; ONE refine stage applied twice, as in the reference.
    refine=False stops after the decoder (pure-torch part: CPU tests of the data-parallel wrapper)."""

    def __init__(self, num_points=16384, n_primitives=32, hide_size=4096, bottleneck_size=4096, width=1026,
                 use_se=False, refine=True):
        super().__init__()
        self.encoder = EdgeConvEncoder(hide_size, hide_size, bottleneck_size, use_se=use_se)
        self.decoder = StyleFoldingDecoder(num_points, n_primitives, bottleneck_size, width, use_se=use_se)
        self.refine = RefineStage(num_points, n_primitives, use_se) if refine else None

    def forward(self, partial):
        return self.forward_staged(partial, lambda cloud: None)

    def forward_staged(self, partial, on_cloud):
        """forward(), calling on_cloud(cloud) as soon as `coarse` and `middle` are final: harness.Completion issues
        their losses on a second HIP stream while the next refine stage's sampler (one workgroup per cloud) runs."""
        part = partial.transpose(1, 2).contiguous()                     # [B,3,M]
        outs = self.decoder(self.encoder(part))                         # [B,3,N]
        coarse = outs.transpose(1, 2).contiguous()
        if self.refine is None:
            return coarse, coarse, coarse, coarse.new_zeros(())
        on_cloud(coarse)
        middle, loss_mst = self.refine(outs, part, coarse)
        on_cloud(middle)
        refine, _ = self.refine(middle.transpose(1, 2).contiguous(), part, middle)
        return coarse, middle, refine, loss_mst


def convert_reference_state_dict(ref_sd, n_primitives=None):
    """The reference SpareNetGenerator's state_dict (encode="Residualnet", use_AdaIn="share";
    models/sparenet_generator.py:12-82) re-keyed to `Generator`'s layout: EdgeConv stages as ModuleLists, the
    per-primitive folding networks (and their squeeze-excite gates) STACKED along a leading primitive axis, their
    BatchNorms concatenated.  Entries without a counterpart are dropped: the unused `conv1`, the unused
    `refine.residual.bn7`, the AdaIN layers' running statistics (never read: AdaIN normalises per instance).
    Accepts tensors or numpy arrays; returns a dict of tensors for `Generator.load_state_dict(..., strict=False)`
    (use `load_reference_state_dict` to also verify that nothing is missing)."""
    T = lambda v: v if isinstance(v, torch.Tensor) else torch.as_tensor(v)
    sd = {k[7:] if k.startswith("module.") else k: T(v) for k, v in ref_sd.items()}   # DataParallel prefix
    out = {}
    fe = "encoder.feat_extractor."
    stats = ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")

    def norm(dst, src):
        for s_ in stats:
            if src + "." + s_ in sd:
                out[dst + "." + s_] = sd[src + "." + s_]

    for i in range(4):
        out[f"encoder.edge.{i}.weight"] = sd[f"{fe}conv{i + 1}.weight"]
        norm(f"encoder.norm.{i}", f"{fe}bn{i + 1}")
        for fc in (0, 2):
            if f"{fe}se{i + 1}.fc.{fc}.weight" in sd:
                out[f"encoder.gate.{i}.fc.{fc}.weight"] = sd[f"{fe}se{i + 1}.fc.{fc}.weight"]
    for i in range(3):
        out[f"encoder.res.{i}.weight"] = sd[f"{fe}resconv{i + 1}.weight"]
    out["encoder.head.weight"] = sd[f"{fe}conv5.weight"]
    norm("encoder.head_norm", f"{fe}bn5")
    out["encoder.linear.weight"], out["encoder.linear.bias"] = sd["encoder.linear.weight"], sd["encoder.linear.bias"]
    norm("encoder.bn", "encoder.bn")
    for k in ("mlp.0.weight", "mlp.0.bias", "mlp.2.weight", "mlp.2.bias"):
        out["decoder." + k] = sd["decoder." + k]
    P = n_primitives or 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("decoder.decoder."))
    prim = lambda p, k: sd[f"decoder.decoder.{p}.dec.{k}"]
    for l in range(4):
        out[f"decoder.weight.{l}"] = torch.stack([prim(p, f"conv{l + 1}.weight")[:, :, 0] for p in range(P)])
        out[f"decoder.bias.{l}"] = torch.stack([prim(p, f"conv{l + 1}.bias") for p in range(P)])
    for l in range(3):
        for s_ in stats[:4]:
            out[f"decoder.bn.{l}.{s_}"] = torch.cat([prim(p, f"bn{l + 1}.{s_}") for p in range(P)])
        if f"decoder.decoder.0.dec.se{l + 1}.fc.0.weight" in sd:
            out[f"decoder.gate.{l}.w1"] = torch.stack([prim(p, f"se{l + 1}.fc.0.weight") for p in range(P)])
            out[f"decoder.gate.{l}.w2"] = torch.stack([prim(p, f"se{l + 1}.fc.2.weight") for p in range(P)])
    rr = "refine.residual."
    if rr + "conv1.weight" in sd:
        for i in range(1, 7):
            out[f"{rr}l{i}.0.weight"], out[f"{rr}l{i}.0.bias"] = sd[f"{rr}conv{i}.weight"], sd[f"{rr}conv{i}.bias"]
            norm(f"{rr}l{i}.1", f"{rr}bn{i}")
        out[rr + "out.weight"], out[rr + "out.bias"] = sd[rr + "conv7.weight"], sd[rr + "conv7.bias"]
        for i in (1, 2, 4, 5, 6):
            for fc in (0, 2):
                if f"{rr}se{i}.fc.{fc}.weight" in sd:
                    out[f"{rr}g{i}.fc.{fc}.weight"] = sd[f"{rr}se{i}.fc.{fc}.weight"]
    return out


def load_reference_state_dict(generator, ref_sd):
    """Load a reference SpareNetGenerator checkpoint (its `state_dict()`, e.g. the `net_G` entry of a SpareNet
    checkpoint file) into `generator`; raises if a parameter or buffer of `generator` is left without a value or a
    shape does not fit.  Returns the reference keys that were not used."""
    conv = convert_reference_state_dict(ref_sd, generator.decoder.P)
    own = generator.state_dict()
    missing = [k for k in own if k not in conv and "num_batches_tracked" not in k]
    if missing:
        raise KeyError(f"reference state_dict lacks what these entries need: {missing[:8]}{' ...' if len(missing) > 8 else ''}")
    for k, v in conv.items():
        if k not in own:
            raise KeyError(f"converted entry {k} has no counterpart in the generator")
        if tuple(own[k].shape) != tuple(v.shape):
            raise ValueError(f"{k}: generator has {tuple(own[k].shape)}, reference gives {tuple(v.shape)}")
    generator.load_state_dict(conv, strict=False)
    used_prefixes = ("encoder.", "decoder.mlp", "decoder.decoder.", "refine.residual.")
    return sorted(k for k in ref_sd if not k.startswith(used_prefixes) or ".adain" in k or "residual.bn7" in k)


# _SpectralNormGemm below builds on a PRIVATE torch class: refuse at import, loudly, if a torch upgrade changed what it
# relies on (the CPU test tests/test_networks.py::test_spectral_norm_gemm_equals_torch is the numeric canary).
_SN_BASE = getattr(nn.utils.parametrizations, "_SpectralNorm", None)
if _SN_BASE is None or not all(hasattr(_SN_BASE, a) for a in ("_power_method", "_reshape_weight_to_matrix", "forward")):
    raise ImportError("sparenet_amd.networks: torch.nn.utils.parametrizations._SpectralNorm no longer has the methods "
                      f"_SpectralNormGemm overrides (torch {torch.__version__}; written against 2.10) -- adapt "
                      "_SpectralNormGemm before using the discriminators")


class _SpectralNormGemm(nn.utils.parametrizations._SpectralNorm):
    """torch's spectral-norm parametrization (same buffers, same state_dict keys, one power iteration per training
    forward as `nn.utils.spectral_norm` in models/sparenet_discriminator.py), with the three matrix-vector products of a
    forward written as GEMMs with one column, in fp32 outside the autocast region.  On ROCm 7.2 `torch.mv` (aten::addmv_
    -> rocBLAS gemv) costs 3.7 ms of HOST time per call: 84 calls per GAN step were 311 of config 5's 340 ms of host
    time, with the GPU busy for 148 ms of a 397 ms step (profiles/r04_c_config5_host_profile.txt)."""

    @staticmethod
    def _mv(mat, vec):
        return torch.mm(mat, vec.unsqueeze(1)).squeeze(1)

    @torch.autograd.no_grad()
    def _power_method(self, weight_mat, n_power_iterations):
        assert weight_mat.ndim > 1
        for _ in range(n_power_iterations):
            self._u = F.normalize(self._mv(weight_mat, self._v), dim=0, eps=self.eps, out=self._u)
            self._v = F.normalize(self._mv(weight_mat.t(), self._u), dim=0, eps=self.eps, out=self._v)

    def forward(self, weight):
        if weight.ndim == 1:
            return F.normalize(weight, dim=0, eps=self.eps)
        with torch.autocast(weight.device.type, enabled=False):
            weight_mat = self._reshape_weight_to_matrix(weight.float())
            if self.training:
                self._power_method(weight_mat, self.n_power_iterations)
            u = self._u.clone(memory_format=torch.contiguous_format)
            v = self._v.clone(memory_format=torch.contiguous_format)
            sigma = torch.dot(u, self._mv(weight_mat, v))
            return weight / sigma


def _sn(module, name="weight"):
    weight = getattr(module, name)
    dim = 1 if isinstance(module, (nn.ConvTranspose1d, nn.ConvTranspose2d, nn.ConvTranspose3d)) else 0
    nn.utils.parametrize.register_parametrization(module, name, _SpectralNormGemm(weight, 1, dim, 1e-12))
    return module


class PatchDiscriminator(nn.Module):
    """[B, 2*views, S, S] -> validity [B,1] (+ the first four feature maps): six spectrally normalised 4x4
    stride-2 convolutions (BatchNorm from the second on, LeakyReLU 0.2), a 3x3 head, spatial mean."""

    def __init__(self, img_shape=(16, 256, 256)):
        super().__init__()
        widths = [img_shape[0], 16, 32, 64, 128, 256, 512]
        blocks = []
        for i in range(6):
            layers = [_sn(nn.Conv2d(widths[i], widths[i + 1], 4, stride=2, padding=1))]
            if i > 0:
                layers.append(nn.BatchNorm2d(widths[i + 1]))
            layers.append(nn.LeakyReLU(0.2))
            blocks.append(nn.Sequential(*layers))
        self.blocks = nn.ModuleList(blocks)
        self.adv_layer = _sn(nn.Conv2d(512, 1, 3, padding=1, bias=False))

    def forward(self, img, feat=False, y=None):
        feats, x = [], img
        with _autocast(img):
            for blk in self.blocks:
                x = blk(x)
                feats.append(x)
            validity = self.adv_layer(x)
        validity = validity.float().mean(dim=(2, 3)).view(img.shape[0], -1)
        return (validity, [f.float() for f in feats[:4]]) if feat else validity


class ProjectionDiscriminator(nn.Module):
    """The conditional variant (use_cgan): four 3x3 stride-2 blocks (LeakyReLU, Dropout2d 0.25, BatchNorm with
    eps 0.8 from the second on), a linear head, plus <embedding(y), features> when labels are given."""

    def __init__(self, num_classes=0, img_shape=(16, 256, 256)):
        super().__init__()
        widths = [img_shape[0], 16, 32, 64, 128]
        blocks = []
        for i in range(4):
            layers = [_sn(nn.Conv2d(widths[i], widths[i + 1], 3, 2, 1)), nn.LeakyReLU(0.2), nn.Dropout2d(0.25)]
            if i > 0:
                layers.append(nn.BatchNorm2d(widths[i + 1], 0.8))
            blocks.append(nn.Sequential(*layers))
        self.blocks = nn.ModuleList(blocks)
        flat = 128 * (img_shape[1] // 16) ** 2
        self.adv_layer = _sn(nn.Linear(flat, 1))
        self.l_y = _sn(nn.Embedding(num_classes, flat)) if num_classes > 0 else None
        nn.init.xavier_uniform_(self.adv_layer.parametrizations.weight.original)
        if self.l_y is not None:
            nn.init.xavier_uniform_(self.l_y.parametrizations.weight.original)

    def forward(self, img, feat=False, y=None):
        feats, x = [], img
        with _autocast(img):
            for blk in self.blocks:
                x = blk(x)
                feats.append(x)
        out = x.float().flatten(1)
        validity = self.adv_layer(out)
        if y is not None and self.l_y is not None:
            validity = validity + torch.sum(self.l_y(y) * out, dim=1, keepdim=True)
        return (validity, [f.float() for f in feats]) if feat else validity


# ------------------------------------------------------------------ objectives (plain arithmetic on op outputs)
def emd_term(dist):
    """sqrt(dist).mean(1).mean()  (sparenet_runner.py:93-99)."""
    return torch.sqrt(dist).mean(1).mean()


def completion_loss(coarse_loss, middle_loss, refine_loss, expansion_penalty, consist_dist1=None):
    """coarse + middle + refine + 0.1 mean(expansion penalty) [+ 0.5 mean(dist1(refine, gt))]
    (sparenet_runner.py:101-106)."""
    loss = coarse_loss + middle_loss + refine_loss + expansion_penalty.mean() * 0.1
    if consist_dist1 is not None:
        loss = loss + torch.mean(consist_dist1).mean() * 0.5
    return loss


def feature_matching(fake_feats, real_feats):
    """sum_j (C_j / sum C) mean((fake_j - real_j)^2)  (sparenet_gan_runner.py:318-326)."""
    maps = [f.shape[1] for f in fake_feats]
    total = float(sum(maps))
    return sum(float(c) / total * torch.mean((f - r.detach()) ** 2)
               for f, r, c in zip(fake_feats, real_feats, maps))


def generator_objective(rec_loss, d_fake_pred, real_label, loss_fm=None, loss_im=None, weight_l2=200.0,
                        weight_gan=0.1, weight_fm=1.0, weight_im=1.0):
    """errG = w_l2 rec + w_gan MSE(D(fake), 1) [+ w_fm fm] [+ w_im L1(fake, real)]  (:336-345); returns
    (errG, errG_D)."""
    err_g_d = F.mse_loss(d_fake_pred, real_label)
    err_g = weight_l2 * rec_loss + weight_gan * err_g_d
    if loss_fm is not None:
        err_g = err_g + weight_fm * loss_fm
    if loss_im is not None:
        err_g = err_g + weight_im * loss_im
    return err_g, err_g_d


def discriminator_objective(d_real_pred, d_fake_pred, real_label, fake_label):
    """LSGAN: MSE(D(real), 1) + MSE(D(fake), 0)  (:262-265); returns (errD_real, errD_fake)."""
    return F.mse_loss(d_real_pred, real_label), F.mse_loss(d_fake_pred, fake_label)


def data_parallel(module, device=None, bucket_cap_mb=128):
    """DistributedDataParallel for the one-process-per-GPU mode (replaces the reference's nn.DataParallel,
    runners/base_runner.py:100-104): gradients are all-reduced over RCCL in buckets while backward is still
    running.  xGMI is point to point (7 links per GPU), a ring all-reduce is bound per link: few LARGE buckets
    (128 MB: the generator's 307 MB of fp32 gradients in three) keep the links streaming instead of paying a
    latency-bound collective per 25 MB.  BatchNorm statistics stay per replica, as under DataParallel."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return module
    ids = [device.index] if device is not None and device.type == "cuda" else None
    return nn.parallel.DistributedDataParallel(module, device_ids=ids, bucket_cap_mb=bucket_cap_mb,
                                               gradient_as_bucket_view=True)
