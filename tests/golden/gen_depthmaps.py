"""Generate tests/golden/depthmaps_*.npz by IMPORTING the reference's
utils/p2i_utils.py (pure PyTorch) in this container.

`cuda.p2i_op` (the CUDA extension it imports) is replaced by a stub whose p2i()
records its arguments and evaluates the max-splat with the reference's own CPU
functors' restatement (oracle.p2i_max_forward, itself pinned bit-exactly to those
functors by tests/golden/p2i_*.npz).  Stored: the 8 pre_matrix 4x4's of both
projections, and for seeded clouds pos_ijs / point_features / depth maps per view.
Usage: python tests/golden/gen_depthmaps.py
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
import oracle  # noqa: E402

CAPTURE = []


def _stub_p2i(points, point_features, batch_inds, background, kernel_radius,
              kernel_kind_str="cos", reduce="max"):
    assert kernel_kind_str == "cos" and reduce == "max"
    h, w = background.shape[2:]
    px = (points + 1) / 2 * torch.tensor([h - 1, w - 1], dtype=points.dtype).view(1, 2)
    out, ids = oracle.p2i_max_forward(px.numpy(), point_features.numpy(), batch_inds.numpy(),
                                      background.numpy(), float(kernel_radius))
    CAPTURE.append(dict(points=points.numpy().copy(), feat=point_features.numpy().copy(),
                        radius=float(kernel_radius)))
    return torch.from_numpy(out)


def load_reference():
    cuda_pkg = types.ModuleType("cuda")
    cuda_pkg.__path__ = []
    p2i_mod = types.ModuleType("cuda.p2i_op")
    p2i_mod.p2i = _stub_p2i
    sys.modules["cuda"] = cuda_pkg
    sys.modules["cuda.p2i_op"] = p2i_mod
    spec = importlib.util.spec_from_file_location("ref_p2i_utils", "/root/reference/utils/p2i_utils.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_reference()
    for proj in ("orthorgonal", "perspective"):
        for name, B, N, S, radii, seed, scale in [
            ("a", 2, 1024, 64, [5.0, 7.0, 10.0], 0, 1.0),
            ("b", 1, 777, 48, [0.02, 0.05], 1, 1.0),
            ("c", 3, 300, 32, [3.0], 2, 0.8),
        ]:
            if proj == "perspective" and name != "a":
                continue
            g = torch.Generator().manual_seed(seed)
            data = torch.rand(B, N, 3, generator=g) - 0.5
            r = ref.ComputeDepthMaps(projection=proj, eyepos_scale=scale, image_size=S)
            mats = torch.cat(r.pre_matrix_list, 0).numpy()
            views = {}
            for v in range(8):
                CAPTURE.clear()
                dm = r(data, view_id=v, radius_list=radii)
                views[f"maps_{v}"] = dm.numpy()
                views[f"ij_{v}"] = CAPTURE[0]["points"]
                views[f"feat_{v}"] = CAPTURE[0]["feat"]
            assert r(data, view_id=8) is None
            np.savez_compressed(
                os.path.join(HERE, f"depthmaps_{proj[:5]}_{name}.npz"), data=data.numpy(),
                pre_matrices=mats, radius_list=np.array(radii, np.float32),
                image_size=np.int32(S), eyepos_scale=np.float32(scale), **views,
                provenance=np.array("reference utils/p2i_utils.py ComputeDepthMaps imported on CPU; "
                                    "p2i = oracle restatement of the reference functors"))
            print(proj, name, mats[0].round(4).tolist()[0], float(views["maps_0"].max()))


if __name__ == "__main__":
    main()
