"""Differentiable multi-view depth-map rendering of point clouds.

Host-side mirror of the reference's utils/p2i_utils.py: ComputeDepthMaps (:168-252)
with the same constructor / forward signature, N_VIEWS_PREDEFINED (:9) and the camera
helpers look_at (:16-82), perspective (:85-121), orthorgonal (:124-150), transform
(:153-165).  The splat itself is the HIP p2i op (sparenet_amd.cuda.p2i_op).

What forward() computes, per view (reference :211-252):
    pos      = (P @ V) [x y z 1]^T, divided by w
    (i, j)   = (-pos.y, pos.x)                      image row / column in [-1, 1]
    feature  = 1 - (pos.z - min z) / (max z - min z)   min/max over the WHOLE input tensor
    map_r    = p2i(ij, feature, reduce="max", kernel_radius=r) on a zero background
    output   = cat_r map_r  -> [B, len(radius_list), S, S]
The camera matrices are evaluated with the same fp32 torch operations as the reference
so the eight P@V matrices agree bit for bit (tests/golden/depthmaps_*.npz); the
per-point transform is evaluated in the order of the reference's batched 4x4 product
(left to right, separately rounded), so projected coordinates and depth features are
bit-equal to the imported reference's.
"""
import ctypes
import math

import torch

from sparenet_amd import _lib

from sparenet_amd.cuda.p2i_op import P2IMaxFunction, P2IMaxMultiFunction, p2i  # noqa: F401

N_VIEWS_PREDEFINED = 8

_EYES = [(sx, sy, sz) for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]


def normalize(x, dim):
    floor = torch.tensor(1e-6, dtype=x.dtype, device=x.device)
    return x / torch.max(x.norm(None, dim=dim, keepdim=True), floor)


def _rows_to_mat(rows):
    """rows: 16 tensors of shape [batch] -> [batch, 4, 4]."""
    return torch.stack(rows, -1).view(-1, 4, 4)


def look_at(eyes, centers, ups):
    """View matrix [batch,4,4] of an observer at `eyes` looking at `centers` with head
    direction `ups` (all [batch,3]): translate the eye to the origin, then rotate the
    observer's right / up / backward axes onto x / y / z."""
    back = normalize(eyes - centers, dim=1)
    right = normalize(torch.cross(ups, back, dim=1), dim=1)
    up = torch.cross(back, right, dim=1)
    o = torch.zeros([eyes.size(0)], dtype=eyes.dtype, device=eyes.device)
    i = torch.ones([eyes.size(0)], dtype=eyes.dtype, device=eyes.device)
    shift = _rows_to_mat([i, o, o, -eyes[:, 0],
                          o, i, o, -eyes[:, 1],
                          o, o, i, -eyes[:, 2],
                          o, o, o, i])
    turn = _rows_to_mat([right[:, 0], right[:, 1], right[:, 2], o,
                         up[:, 0], up[:, 1], up[:, 2], o,
                         back[:, 0], back[:, 1], back[:, 2], o,
                         o, o, o, i])
    return turn @ shift


def perspective(fovy, aspect, z_near, z_far):
    """Right-handed perspective projection [batch,4,4]; all arguments are [batch]."""
    t = torch.tan(fovy / 2.0)
    o = torch.zeros_like(fovy)
    i = torch.ones_like(fovy)
    k1 = -(z_far + z_near) / (z_far - z_near)
    k2 = -2.0 * z_far * z_near / (z_far - z_near)
    return _rows_to_mat([1.0 / aspect / t, o, o, o,
                         o, 1.0 / t, o, o,
                         o, o, k1, k2,
                         o, o, -i, o])


def orthorgonal(scalex, scaley, z_near, z_far):
    """Orthographic projection [batch,4,4] (the reference's spelling is kept)."""
    o = torch.zeros_like(z_near)
    i = torch.ones_like(z_near)
    k1 = -2.0 / (z_far - z_near)
    k2 = (z_far + z_near) / (z_far - z_near)
    return _rows_to_mat([scalex, o, o, o,
                         o, scaley, o, o,
                         o, o, k1, k2,
                         o, o, o, i])


def transform(matrix, points):
    """matrix [4,4] or [npoints,4,4], points [npoints,3] -> projected [npoints,3]."""
    if matrix.dim() == 3:
        hom = torch.cat([points, torch.ones_like(points[:, :1])], dim=1).unsqueeze(-1)
        out = (matrix @ hom).squeeze(-1)
    else:
        # one matrix for all points: the same left-to-right sum of separately rounded products the
        # reference's batched product evaluates on the CPU (and depth_project.hip on the GPU), without
        # materialising npoints copies of the matrix
        x, y, z = points[:, 0:1], points[:, 1:2], points[:, 2:3]
        m = matrix
        out = ((m[:, 0] * x + m[:, 1] * y) + m[:, 2] * z) + m[:, 3]
    return out[:, :3] / out[:, 3:4]


class DepthProjectFunction(torch.autograd.Function):
    """(data [B,N,3], P@V as 16 host floats, image_size) -> (pixel_ijs [B*N,2], point_features
    [B*N,1]): ComputeDepthMaps.project() followed by p2i's NDC -> pixel rescale, as two HIP
    kernels each way (sn_depth_project_forward / _backward) instead of ~25 torch ops and their
    autograd nodes per view."""

    @staticmethod
    def forward(ctx, data, matrix16, image_size):
        pts = data.contiguous().float().view(-1, 3)
        n = pts.size(0)
        dev = pts.device
        pixel = torch.empty(n, 2, device=dev)
        feat = torch.empty(n, 1, device=dev)
        z = torch.empty(n, device=dev)
        zminmax = torch.empty(2, dtype=torch.int32, device=dev)
        mat = (ctypes.c_float * 16)(*matrix16)
        extent = float(image_size - 1)
        with torch.cuda.device_of(pts):
            code = _lib.lib().sn_depth_project_forward(
                _lib.fptr(pts, "data"), ctypes.c_long(n), mat, _lib.cfloat(extent),
                _lib.fptr(pixel, "pixel"), _lib.fptr(z, "z"), ctypes.c_void_p(zminmax.data_ptr()),
                _lib.fptr(feat, "feat"), _lib.stream_of(pts))
        _lib.check(code, "sn_depth_project_forward")
        ctx.save_for_backward(pts, z, zminmax)
        ctx.mat, ctx.extent, ctx.shape = mat, extent, data.shape
        return pixel, feat

    @staticmethod
    def backward(ctx, g_pixel, g_feat):
        pts, z, zminmax = ctx.saved_tensors
        n = pts.size(0)
        g_data = torch.empty_like(pts)
        ws = torch.empty(32, dtype=torch.uint8, device=pts.device)
        gp = g_pixel.contiguous().float() if g_pixel is not None else None
        gf = g_feat.contiguous().float() if g_feat is not None else None
        null = ctypes.c_void_p(0)
        with torch.cuda.device_of(pts):
            code = _lib.lib().sn_depth_project_backward(
                _lib.fptr(pts, "data"), ctypes.c_long(n), ctx.mat, _lib.cfloat(ctx.extent),
                _lib.fptr(z, "z"), ctypes.c_void_p(zminmax.data_ptr()),
                _lib.fptr(gp, "g_pixel") if gp is not None else null,
                _lib.fptr(gf, "g_feat") if gf is not None else null,
                ctypes.c_void_p(ws.data_ptr()), _lib.fptr(g_data, "g_data"), _lib.stream_of(pts))
        _lib.check(code, "sn_depth_project_backward")
        return g_data.view(ctx.shape), None, None


class DepthProjectViewsFunction(torch.autograd.Function):
    """DepthProjectFunction for several views at once: (data [B,N,3], [V][16] host matrices, image_size) ->
    (pixel_ijs [V*B*N, 2], point_features [V*B*N, 1]), view-major.  The depth feature is normalised per view
    over the whole input tensor, exactly as V separate calls would; the backward sums the views' gradients."""

    @staticmethod
    def forward(ctx, data, matrices, image_size):
        pts = data.contiguous().float().view(-1, 3)
        n, v = pts.size(0), len(matrices)
        dev = pts.device
        pixel = torch.empty(v * n, 2, device=dev)
        feat = torch.empty(v * n, 1, device=dev)
        z = torch.empty(v * n, device=dev)
        zminmax = torch.empty(2 * v, dtype=torch.int32, device=dev)
        mat = (ctypes.c_float * (16 * v))(*[x for m in matrices for x in m])
        extent = float(image_size - 1)
        with torch.cuda.device_of(pts):
            code = _lib.lib().sn_depth_project_forward_views(
                _lib.fptr(pts, "data"), ctypes.c_long(n), mat, v, _lib.cfloat(extent),
                _lib.fptr(pixel, "pixel"), _lib.fptr(z, "z"), ctypes.c_void_p(zminmax.data_ptr()),
                _lib.fptr(feat, "feat"), _lib.stream_of(pts))
        _lib.check(code, "sn_depth_project_forward_views")
        ctx.save_for_backward(pts, z, zminmax)
        ctx.mat, ctx.nviews, ctx.extent, ctx.shape = mat, v, extent, data.shape
        return pixel, feat

    @staticmethod
    def backward(ctx, g_pixel, g_feat):
        pts, z, zminmax = ctx.saved_tensors
        n = pts.size(0)
        g_data = torch.empty_like(pts)
        ws = torch.empty(32 * ctx.nviews, dtype=torch.uint8, device=pts.device)
        gp = g_pixel.contiguous().float() if g_pixel is not None else None
        gf = g_feat.contiguous().float() if g_feat is not None else None
        null = ctypes.c_void_p(0)
        with torch.cuda.device_of(pts):
            code = _lib.lib().sn_depth_project_backward_views(
                _lib.fptr(pts, "data"), ctypes.c_long(n), ctx.mat, ctx.nviews, _lib.cfloat(ctx.extent),
                _lib.fptr(z, "z"), ctypes.c_void_p(zminmax.data_ptr()),
                _lib.fptr(gp, "g_pixel") if gp is not None else null,
                _lib.fptr(gf, "g_feat") if gf is not None else null,
                ctypes.c_void_p(ws.data_ptr()), _lib.fptr(g_data, "g_data"), _lib.stream_of(pts))
        _lib.check(code, "sn_depth_project_backward_views")
        return g_data.view(ctx.shape), None, None


class ComputeDepthMaps(torch.nn.Module):
    def __init__(self, projection: str = "orthorgonal", eyepos_scale: float = 1.0,
                 image_size: int = 256):
        super().__init__()
        assert projection in {"perspective", "orthorgonal"}
        self.image_size = image_size
        self.eyes_pos_list = [list(e) for e in _EYES]
        self.num_views = len(self.eyes_pos_list)
        f32 = dict(dtype=torch.float32)
        if projection == "perspective":
            self.projection_matrix = perspective(
                fovy=torch.tensor([math.pi / 4], **f32), aspect=torch.tensor([1.0], **f32),
                z_near=torch.tensor([0.1], **f32), z_far=torch.tensor([10.0], **f32))
        else:
            self.projection_matrix = orthorgonal(
                scalex=torch.tensor([1.5], **f32), scaley=torch.tensor([1.5], **f32),
                z_near=torch.tensor([0.1], **f32), z_far=torch.tensor([10.0], **f32))
        mats = []
        for eye in self.eyes_pos_list:
            view = look_at(eyes=torch.tensor([eye], **f32) * eyepos_scale,
                           centers=torch.tensor([[0, 0, 0]], **f32),
                           ups=torch.tensor([[0, 0, 1]], **f32))
            mats.append((self.projection_matrix @ view)[0])
        # [8,4,4]; a real buffer, so .to(device) moves all views once (the reference keeps a
        # Python list of CPU matrices and copies one to the device on every call, :208,:217)
        self.register_buffer("pre_matrices", torch.stack(mats), persistent=False)
        self.register_buffer("_extent", torch.tensor([[image_size - 1.0, image_size - 1.0]], **f32),
                             persistent=False)
        self.pre_matrix_list = [m.unsqueeze(0) for m in mats]
        self._host_mats = [[float(v) for v in m.reshape(-1).tolist()] for m in mats]
        self._batch_inds_cache = {}

    def _batch_inds(self, batch, npoints, device):
        key = (batch, npoints, str(device))
        if key not in self._batch_inds_cache:
            self._batch_inds_cache = {key: torch.arange(batch, dtype=torch.int32, device=device)
                                      .repeat_interleave(npoints)}
        return self._batch_inds_cache[key]

    def project(self, data, view_id):
        """data [B,N,3] -> (pos_ijs [B*N,2], point_features [B*N,1]) exactly as the reference
        builds them before calling p2i."""
        m = self.pre_matrices[view_id].to(device=data.device, dtype=data.dtype)
        pos = transform(m, data.reshape(-1, 3))
        pos_ijs = torch.stack([-pos[:, 1], pos[:, 0]], dim=1)
        z = pos[:, 2:3]
        zmin, zmax = z.min(), z.max()
        point_features = 1.0 - (z - zmin) / (zmax - zmin)
        return pos_ijs, point_features

    def forward(self, data, view_id=0, radius_list=[10.0]):
        if view_id >= self.num_views:
            return None
        batch, npoints = data.size(0), data.size(1)
        background = torch.zeros(batch, 1, self.image_size, self.image_size, dtype=data.dtype,
                                 device=data.device)
        batch_inds = self._batch_inds(batch, npoints, data.device)
        # the reference calls p2i() once per radius (:230-251); the projection, the depth feature,
        # the NDC -> pixel rescale that p2i() performs (cuda/p2i_op/__init__.py:117-121), the zero
        # background and the points do not depend on the radius, so they are computed once (on
        # the GPU by the fused projection kernels) and up to four radii share one splat pass
        if data.is_cuda and data.dtype == torch.float32:
            pixel_ijs, point_features = DepthProjectFunction.apply(data, self._host_mats[view_id],
                                                                   self.image_size)
        else:
            pos_ijs, point_features = self.project(data, view_id)
            pixel_ijs = (pos_ijs + 1) / 2 * self._extent.to(device=data.device, dtype=data.dtype)
        radii = [float(r) for r in radius_list]
        maps = []
        for i in range(0, len(radii), 4):
            chunk = radii[i:i + 4]
            if len(chunk) == 1:
                maps.append(P2IMaxFunction.apply(pixel_ijs, point_features, batch_inds, background,
                                                 0, chunk[0]))
            else:
                stacked = P2IMaxMultiFunction.apply(pixel_ijs, point_features, batch_inds,
                                                    background, 0, chunk, True)   # [B,r,1,S,S], written that way
                maps.append(stacked.view(batch, len(chunk), self.image_size, self.image_size))
        return maps[0] if len(maps) == 1 else torch.cat(maps, dim=1)

    def forward_views(self, data, view_ids=None, radius_list=[10.0]):
        """All requested views in one pass: returns [V, B, len(radius_list), S, S] with slice v equal to
        forward(data, view_ids[v], radius_list) bit for bit.  The reference renders view by view
        (runners/sparenet_gan_runner.py:217-225 loops over the 8 predefined views); a sweep over V views is
        V x B independent images, so here the views join the batch -- one projection, one binning, one
        splat and one backward launch set for the whole sweep instead of ~10 short launches per view.
        MI355X extension of the reference API (CUDA fp32 tensors, up to four radii)."""
        view_ids = list(range(self.num_views)) if view_ids is None else [int(v) for v in view_ids]
        radii = [float(r) for r in radius_list]
        if not (data.is_cuda and data.dtype == torch.float32 and 1 <= len(radii) <= 4 and 1 <= len(view_ids) <= 8):
            return torch.stack([self.forward(data, v, radius_list) for v in view_ids])
        batch, npoints, nv = data.size(0), data.size(1), len(view_ids)
        s = self.image_size
        pixel_ijs, point_features = DepthProjectViewsFunction.apply(
            data, [self._host_mats[v] for v in view_ids], s)
        batch_inds = self._batch_inds(nv * batch, npoints, data.device)
        if len(radii) > 1 and max(radii) <= 16.0:   # the zero background as a shape: nothing to allocate or read
            stacked = P2IMaxMultiFunction.apply(pixel_ijs, point_features, batch_inds, (nv * batch, 1, s, s), 0,
                                                radii, True)
            return stacked.view(nv, batch, len(radii), s, s)      # [V*B,R,1,S,S] as the kernel wrote it
        background = torch.zeros(nv * batch, 1, s, s, dtype=data.dtype, device=data.device)
        if len(radii) == 1:
            maps = P2IMaxFunction.apply(pixel_ijs, point_features, batch_inds, background, 0, radii[0])
            return maps.view(nv, batch, 1, s, s)
        stacked = P2IMaxMultiFunction.apply(pixel_ijs, point_features, batch_inds, background, 0, radii, True)
        return stacked.view(nv, batch, len(radii), s, s)      # [V*B,R,1,S,S] as the kernel wrote it
