"""Generate tests/golden/p2i_*.npz from the REFERENCE's own p2i functors on the CPU.

The functor templates of /root/reference/cuda/p2i_op/{p2i_max.h,p2i_sum.h} are
plain C++ and run through the reference's own kernel<cpu_device>::launch
(common.h:55-78).  This script extracts the functor line ranges at run time
(nothing is copied into the repo), compiles tests/golden/gen/ref_p2i.cpp against
them + the reference's utility.h/common.h + torch headers, runs it on seeded
inputs and stores inputs + outputs.  See ref_p2i.cpp for the two accommodations
(op wrappers skipped; corrected CPU atomic_cas).
Also stores the 8x8 known-answer geometry of cuda/p2i_op/p2i_test.py:10-20.

Usage: python tests/golden/gen_p2i.py
"""
import os
import struct
import subprocess
import sys
import sysconfig
import tempfile

import numpy as np
import torch
import torch.utils.cpp_extension as cpp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/cuda/p2i_op"
TMP = os.path.join(tempfile.gettempdir(), "sn_ref_extract")


def extract(fname, first, last, out):
    os.makedirs(TMP, exist_ok=True)
    lines = open(os.path.join(REF, fname)).read().split("\n")[first - 1:last]
    path = os.path.join(TMP, out)
    open(path, "w").write("namespace haya_ext {\n" + "\n".join(lines) + "\n}\n")
    return path


def build():
    mx = extract("p2i_max.h", 7, 143, "p2i_max_functors.inc")
    sm = extract("p2i_sum.h", 7, 131, "p2i_sum_functors.inc")
    exe = os.path.join(TMP, "ref_p2i")
    inc = cpp.include_paths() + [sysconfig.get_paths()["include"], REF]
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-std=c++17", "-O2", "-w", f'-DREF_MAX_INC="{mx}"', f'-DREF_SUM_INC="{sm}"',
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
    for p in inc:
        cmd += ["-I", p]
    cmd += [os.path.join(HERE, "gen", "ref_p2i.cpp"), "-o", exe, f"-L{libdir}",
            f"-Wl,-rpath,{libdir}", "-lc10", "-ltorch_cpu"]
    subprocess.check_call(cmd)
    return exe


def run(exe, name, points, feat, bi, bg, radius, seed):
    n, C = feat.shape
    B, _, H, W = bg.shape
    g = torch.Generator().manual_seed(seed + 100)
    og = torch.rand(bg.shape, generator=g).numpy().astype(np.float32)
    fin, fout = os.path.join(TMP, "p2i_in.bin"), os.path.join(TMP, "p2i_out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("iiiiif", n, C, B, H, W, radius))
        for a in (points, feat, bi, bg, og):
            f.write(np.ascontiguousarray(a).tobytes())
    subprocess.check_call([exe, fin, fout])
    raw = open(fout, "rb").read()
    px = B * C * H * W
    o = 0

    def take(dtype, count, shape):
        nonlocal o
        a = np.frombuffer(raw, dtype, count, o).reshape(shape)
        o += 4 * count
        return a

    out = take(np.float32, px, bg.shape)
    ids = take(np.int32, px, bg.shape)
    gp = take(np.float32, n * 2, (n, 2))
    gf = take(np.float32, n * C, (n, C))
    gb = take(np.float32, px, bg.shape)
    sout = take(np.float32, px, bg.shape)
    sgp = take(np.float32, n * 2, (n, 2))
    sgf = take(np.float32, n * C, (n, C))
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"), points=points, feat=feat, batch_inds=bi, background=bg,
        radius=np.float32(radius), out_grad=og, max_out=out, max_ids=ids, max_points_grad=gp,
        max_feat_grad=gf, max_background_grad=gb, sum_out=sout, sum_points_grad=sgp,
        sum_feat_grad=sgf,
        provenance=np.array("reference p2i_max.h:7-143 / p2i_sum.h:7-131 functors via kernel<cpu_device>"))
    print(name, "nonzero px", int((ids >= 0).sum()), "max", float(out.max()))
    return out, ids


def main():
    exe = build()
    # the reference's own test1 geometry (p2i_test.py:10-20): one point at the image centre
    # of an 8x8 map, radius 2, feature 1 -> pixel-space (3.5, 3.5)
    pts = np.array([[3.5, 3.5]], np.float32)
    out, ids = run(exe, "p2i_known_8x8_r2", pts, np.ones((1, 3), np.float32),
                   np.zeros(1, np.int32), np.zeros((1, 3, 8, 8), np.float32), 2.0, 0)
    print("  known answer centre/ring:", out[0, 0, 3, 3], out[0, 0, 2, 3])
    for name, B, n, C, S, radius, seed, bgval in [
        ("p2i_rand_2x400_32_r2.5", 2, 400, 1, 32, 2.5, 1, 0.3),
        ("p2i_rand_2x2048_64_r5", 2, 2048, 1, 64, 5.0, 2, 0.0),
        ("p2i_rand_2x2048_64_r0.02", 2, 2048, 1, 64, 0.02, 3, 0.0),
        ("p2i_rand_1x300_24_c3_r3", 1, 300, 3, 24, 3.0, 4, 0.1),
        ("p2i_rand_3x500_40_r7_oob", 3, 500, 2, 40, 7.0, 5, 0.0),
    ]:
        g = torch.Generator().manual_seed(seed)
        ndc = torch.rand(B * n, 2, generator=g) * 2.4 - 1.2      # some points fall off the image
        pts = ((ndc + 1) / 2 * torch.tensor([S - 1, S - 1], dtype=torch.float32).view(1, 2)).numpy()
        feat = torch.rand(B * n, C, generator=g).numpy()
        bi = torch.arange(B, dtype=torch.int32).unsqueeze(1).expand(B, n).reshape(-1).numpy().copy()
        if "oob" in name:
            bi[::37] = -1
            bi[5::41] = B  # out-of-range batch ids are skipped (p2i_max.h:27-29)
        bg = np.full((B, C, S, S), bgval, np.float32)
        run(exe, name, pts.astype(np.float32), feat.astype(np.float32), bi, bg, radius, seed)


if __name__ == "__main__":
    main()
