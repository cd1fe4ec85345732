"""Generate tests/golden/networks_*.npz from the REFERENCE's own model classes, imported here on the CPU.

models/sparenet_generator.py imports with cuda.MDS / cuda.expansion_penalty stubbed (they are only used by
SpareNetRefine.forward, which is not called).  For small widths the script instantiates the reference's
EdgeConvResFeat + SpareNetEncode head, its GridDecoder / StyleBasedAdaIn primitives with the shared AdaIN MLP
(the loop of SpareNetDecode.forward restated with CPU tensors: the original builds the lattice with
torch.cuda.FloatTensor), and PointNetRes; runs them in training mode (batch statistics) on seeded inputs;
and stores the inputs, the outputs and the parameters RE-KEYED to sparenet_amd.networks' layout (stacked
per-primitive weights).  Nothing of the reference's text is stored: arrays only.

Usage: python tests/golden/gen_networks.py        (needs /root/reference; CPU only)
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def reference_module():
    for name in ("cuda", "cuda.MDS", "cuda.MDS.MDS_module", "cuda.expansion_penalty",
                 "cuda.expansion_penalty.expansion_penalty_module"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["cuda.expansion_penalty.expansion_penalty_module"].expansionPenaltyModule = object
    spec = importlib.util.spec_from_file_location("ref_generator", "/root/reference/models/sparenet_generator.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def np_state(sd):
    return {k: v.detach().numpy().copy() for k, v in sd.items() if "num_batches_tracked" not in k}


def encoder_case(ref, use_se, seed):
    torch.manual_seed(seed)
    hide, out, bott, B, M = 64, 32, 24, 3, 40
    feat = ref.EdgeConvResFeat(use_SElayer=use_se, k=8, output_size=out, hide_size=hide)
    lin, bn = torch.nn.Linear(out, bott), torch.nn.BatchNorm1d(bott)
    x = torch.rand(B, 3, M) - 0.5
    y = torch.relu(bn(lin(feat(x))))
    sd = {}
    r = np_state(feat.state_dict())
    for i in range(4):
        sd[f"edge.{i}.weight"] = r[f"conv{i + 1}.weight"]
        for s in ("weight", "bias", "running_mean", "running_var"):
            sd[f"norm.{i}.{s}"] = r[f"bn{i + 1}.{s}"]
        if use_se:
            sd[f"gate.{i}.fc.0.weight"] = r[f"se{i + 1}.fc.0.weight"]
            sd[f"gate.{i}.fc.2.weight"] = r[f"se{i + 1}.fc.2.weight"]
    for i in range(3):
        sd[f"res.{i}.weight"] = r[f"resconv{i + 1}.weight"]
    sd["head.weight"] = r["conv5.weight"]
    for s in ("weight", "bias", "running_mean", "running_var"):
        sd[f"head_norm.{s}"] = r[f"bn5.{s}"]
        sd[f"bn.{s}"] = np_state(bn.state_dict())[s]
    sd["linear.weight"], sd["linear.bias"] = lin.weight.detach().numpy(), lin.bias.detach().numpy()
    return dict(kind="encoder", use_se=use_se, hide=hide, out=out, bott=bott, x=x.numpy(), y=y.detach().numpy(),
                **{"p:" + k: v for k, v in sd.items()})


def decoder_case(ref, seed):
    torch.manual_seed(seed)
    P, n, style_dim, width, B = 3, 32, 20, 18, 4
    prims = [ref.StyleBasedAdaIn(input_dim=2, style_dim=style_dim, bottleneck_size=width) for _ in range(P)]
    nparams = ref.get_num_adain_params(prims[0])
    mlp = torch.nn.Sequential(torch.nn.Linear(style_dim, style_dim), torch.nn.ReLU(),
                              torch.nn.Linear(style_dim, nparams))
    style = torch.randn(B, style_dim)
    grid = ref.grid_generation(P * n, P)
    adain = mlp(style)
    outs = []
    for i in range(P):                      # SpareNetDecode.forward, use_AdaIn == "share", on CPU tensors
        g = torch.tensor(grid[i], dtype=torch.float32).transpose(0, 1).contiguous().unsqueeze(0)
        g = ((g.expand(B, g.size(1), g.size(2)).contiguous() - 0.5) * 2).contiguous()
        outs.append(prims[i](g, style, adain))
    y = torch.cat(outs, 2)
    sd = {"mlp.0.weight": mlp[0].weight, "mlp.0.bias": mlp[0].bias, "mlp.2.weight": mlp[2].weight,
          "mlp.2.bias": mlp[2].bias}
    sd = {k: v.detach().numpy() for k, v in sd.items()}
    states = [np_state(p.dec.state_dict()) for p in prims]
    for l in range(4):
        sd[f"weight.{l}"] = np.stack([s[f"conv{l + 1}.weight"][:, :, 0] for s in states])
        sd[f"bias.{l}"] = np.stack([s[f"conv{l + 1}.bias"] for s in states])
    for l in range(3):
        for s_ in ("weight", "bias", "running_mean", "running_var"):
            sd[f"bn.{l}.{s_}"] = np.concatenate([s[f"bn{l + 1}.{s_}"] for s in states])
    return dict(kind="decoder", P=P, n=n, style_dim=style_dim, width=width, style=style.numpy(),
                y=y.detach().numpy(), **{"p:" + k: v for k, v in sd.items()})


def residual_case(ref, use_se, seed):
    torch.manual_seed(seed)
    net = ref.PointNetRes(use_SElayer=use_se)
    x = torch.rand(2, 4, 50) - 0.5
    y = net(x)
    r = np_state(net.state_dict())
    sd = {}
    for i in range(1, 7):
        sd[f"l{i}.0.weight"], sd[f"l{i}.0.bias"] = r[f"conv{i}.weight"], r[f"conv{i}.bias"]
        for s in ("weight", "bias", "running_mean", "running_var"):
            sd[f"l{i}.1.{s}"] = r[f"bn{i}.{s}"]
    sd["out.weight"], sd["out.bias"] = r["conv7.weight"], r["conv7.bias"]
    if use_se:
        for i in (1, 2, 4, 5, 6):
            sd[f"g{i}.fc.0.weight"], sd[f"g{i}.fc.2.weight"] = r[f"se{i}.fc.0.weight"], r[f"se{i}.fc.2.weight"]
    return dict(kind="residual", use_se=use_se, x=x.numpy(), y=y.detach().numpy(),
                **{"p:" + k: v for k, v in sd.items()})


def generator_case(ref, use_se, seed, with_residual):
    """A reference-KEYED state_dict of a small generator (the reference's own SpareNetEncode / StyleBasedAdaIn /
    PointNetRes instances under the key names SpareNetGenerator gives them -- SpareNetDecode itself hard-codes the
    primitive width 1026, 1.3 M parameters per primitive, too large for a fixture) and the outputs of its three
    parts in training mode: what sparenet_amd.networks.load_reference_state_dict has to reproduce."""
    torch.manual_seed(seed)
    P, n, hide, out, bott, width, B, M = 3, 32, 64, 64, 24, 32, 3, 40
    # SpareNetEncode = EdgeConvResFeat + Linear + BatchNorm1d + ReLU (models/sparenet_generator.py:104-120); built
    # from its parts because SpareNetEncode does not pass a small hide_size on to the feature extractor
    feat = ref.EdgeConvResFeat(use_SElayer=use_se, k=8, output_size=out, hide_size=hide)
    lin, bn = torch.nn.Linear(out, bott), torch.nn.BatchNorm1d(bott)
    prims = [ref.StyleBasedAdaIn(input_dim=2, style_dim=bott, bottleneck_size=width, use_SElayer=use_se) for _ in range(P)]
    mlp = torch.nn.Sequential(torch.nn.Linear(bott, bott), torch.nn.ReLU(),
                              torch.nn.Linear(bott, ref.get_num_adain_params(prims[0])))
    res = ref.PointNetRes(use_SElayer=use_se) if with_residual else None
    x = torch.rand(B, 3, M) - 0.5
    style = torch.relu(bn(lin(feat(x))))
    grid = ref.grid_generation(P * n, P)
    adain = mlp(style)
    outs = []
    for i in range(P):
        g = torch.tensor(grid[i], dtype=torch.float32).transpose(0, 1).contiguous().unsqueeze(0)
        g = ((g.expand(B, g.size(1), g.size(2)).contiguous() - 0.5) * 2).contiguous()
        outs.append(prims[i](g, style, adain))
    coarse = torch.cat(outs, 2)
    base = torch.cat((coarse, torch.zeros(B, 1, P * n)), 1)
    offs = res(base) if with_residual else torch.zeros(0)
    sd = {"conv1.weight": torch.zeros(64, 3, 1), "conv1.bias": torch.zeros(64)}     # present in the reference, unused
    sd.update({"encoder.feat_extractor." + k: v for k, v in feat.state_dict().items()})
    sd.update({"encoder.linear." + k: v for k, v in lin.state_dict().items()})
    sd.update({"encoder.bn." + k: v for k, v in bn.state_dict().items()})
    sd.update({"decoder.mlp." + k: v for k, v in mlp.state_dict().items()})
    for i, pr in enumerate(prims):
        sd.update({f"decoder.decoder.{i}." + k: v for k, v in pr.state_dict().items()})
    if with_residual:
        sd.update({"refine.residual." + k: v for k, v in res.state_dict().items()})
    return dict(kind="generator", use_se=use_se, P=P, n=n, hide=hide, out=out, bott=bott, width=width, x=x.numpy(),
                style=style.detach().numpy(), coarse=coarse.detach().numpy(), offsets=offs.detach().numpy(),
                **{"ref:" + k: v.detach().numpy() for k, v in sd.items()})


def main():
    ref = reference_module()
    cases = {"networks_encoder": encoder_case(ref, False, 1), "networks_encoder_se": encoder_case(ref, True, 2),
             "networks_decoder": decoder_case(ref, 3), "networks_residual": residual_case(ref, False, 4),
             "networks_generator_sd": generator_case(ref, False, 5, False),
             "networks_generator_sd_se": generator_case(ref, True, 6, True)}
    for name, c in cases.items():
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **c)
        print(name, {k: getattr(v, "shape", v) for k, v in c.items() if not k.startswith(("p:", "ref:"))})


if __name__ == "__main__":
    main()
