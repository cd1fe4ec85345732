"""Generate tests/golden/chamfer_*.npz from the REFERENCE's own CPU Chamfer path.

Runs only where /root/reference exists: `make -C oracle ref` compiles
/root/reference/cuda/chamfer_distance/chamfer_distance.cpp unmodified into
oracle/_ref/cd_ref.so; this script calls its forward/backward on seeded inputs
and stores inputs + outputs.  Usage:  python tests/golden/gen_chamfer.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def run(name, xyz1, xyz2, seed):
    d1, d2, i1, i2 = ref.chamfer_forward(xyz1, xyz2)
    g = torch.Generator().manual_seed(seed + 1)
    gd1 = torch.rand(d1.shape, generator=g)
    gd2 = torch.rand(d2.shape, generator=g)
    g1, g2 = ref.chamfer_backward(xyz1, xyz2, gd1, gd2, i1, i2)
    np.savez_compressed(
        os.path.join(OUT, name), xyz1=xyz1.numpy(), xyz2=xyz2.numpy(), dist1=d1.numpy(),
        dist2=d2.numpy(), idx1=i1.numpy(), idx2=i2.numpy(), graddist1=gd1.numpy(),
        graddist2=gd2.numpy(), gradxyz1=g1.numpy(), gradxyz2=g2.numpy(),
        provenance=np.array("reference chamfer_distance.cpp CPU path via oracle/_ref/cd_ref.so"))
    print(name, tuple(xyz1.shape), tuple(xyz2.shape))


def main():
    g = torch.Generator().manual_seed(0)
    # ragged sizes, not multiples of any tile/chunk/block constant
    run("chamfer_rand_2x1300x777.npz", torch.rand(2, 1300, 3, generator=g),
        torch.rand(2, 777, 3, generator=g), 0)
    # heavy exact ties: lattice coordinates k/5
    g = torch.Generator().manual_seed(1)
    run("chamfer_ties_2x600x500.npz", torch.randint(0, 6, (2, 600, 3), generator=g).float() / 5,
        torch.randint(0, 6, (2, 500, 3), generator=g).float() / 5, 1)
    # tiny: fewer targets than one chunk, single query
    g = torch.Generator().manual_seed(2)
    run("chamfer_tiny_3x1x5.npz", torch.rand(3, 1, 3, generator=g),
        torch.rand(3, 5, 3, generator=g), 2)
    # BASELINE config 1 shape, stored compactly: [4,2048,3] x2
    g = torch.Generator().manual_seed(1234)
    run("chamfer_c1_4x2048x2048.npz", torch.rand(4, 2048, 3, generator=g),
        torch.rand(4, 2048, 3, generator=g), 1234)


if __name__ == "__main__":
    main()
