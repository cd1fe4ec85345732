"""p2i -- paint point-cloud features onto a 2-D feature map.  Host-side mirror of
cuda/p2i_op/__init__.py (P2ISumFunction :22-56, P2IMaxFunction :59-93, p2i()
:99-131, custom_fun :133), backed by sn_p2i_{max,sum}_{forward,backward}[_f64]
(include/sparenet_hip.h).  float32 (the rendering path: binned gather, exact fixed-point
backward) and float64 (the reference dispatches both; its own test is a float64 gradcheck)
CUDA tensors.
"""
import ctypes

import torch
from torch.autograd import Function

from sparenet_amd import _lib

__all__ = ["p2i"]


def _shapes(points, point_features, background):
    npoints = points.size(0)
    channels = point_features.size(1)
    batch, bc, out_h, out_w = background if isinstance(background, (tuple, list)) else background.shape
    if points.dim() != 2 or points.size(1) != 2:
        raise ValueError("p2i: points must be [npoints, 2]")
    if point_features.size(0) != npoints or bc != channels:
        raise ValueError("p2i: point_features must be [npoints, channels] and background "
                         "[batch, channels, out_h, out_w]")
    return npoints, channels, batch, out_h, out_w


class _Ext:
    """Stand-in for the reference's JIT-built `ext` module (cuda/p2i_op/ext.cpp:5-15)."""

    @staticmethod
    def p2i_max_forward_gpu(points, point_features, batch_inds, background, kernel_kind,
                            kernel_radius):
        if kernel_kind != 0:
            raise ValueError("p2i: only kernel_kind 0 ('cos') exists")
        n, c, b, h, w = _shapes(points, point_features, background)
        out = torch.empty_like(background)
        ids = torch.empty(background.shape, dtype=torch.int32, device=background.device)
        if points.dtype == torch.float64:
            with torch.cuda.device_of(background):
                nbytes = _lib.lib().sn_p2i_f64_workspace_bytes(b, c, h, w)
                ws = torch.empty(nbytes, dtype=torch.uint8, device=background.device)
                code = _lib.lib().sn_p2i_max_forward_f64(
                    _lib.dptr(points, "points"), _lib.dptr(point_features, "point_features"),
                    _lib.iptr(batch_inds, "batch_inds"), _lib.dptr(background, "background"),
                    n, c, b, h, w, ctypes.c_double(kernel_radius), _lib.dptr(out, "out"),
                    _lib.iptr(ids, "out_point_ids"), ctypes.c_void_p(ws.data_ptr()),
                    ctypes.c_size_t(nbytes), _lib.stream_of(background))
            _lib.check(code, "sn_p2i_max_forward_f64")
            return out, ids
        with torch.cuda.device_of(background):
            nbytes = _lib.lib().sn_p2i_max_workspace_bytes(b, c, h, w)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=background.device)
            code = _lib.lib().sn_p2i_max_forward(
                _lib.fptr(points, "points"), _lib.fptr(point_features, "point_features"),
                _lib.iptr(batch_inds, "batch_inds"), _lib.fptr(background, "background"),
                n, c, b, h, w, _lib.cfloat(kernel_radius), _lib.fptr(out, "out"),
                _lib.iptr(ids, "out_point_ids"), ctypes.c_void_p(ws.data_ptr()),
                ctypes.c_size_t(nbytes), _lib.stream_of(background))
        _lib.check(code, "sn_p2i_max_forward")
        return out, ids

    @staticmethod
    def p2i_max_forward_multi_gpu(points, point_features, batch_inds, background, kernel_kind,
                                  radii, image_major=False):
        """All radii of one ComputeDepthMaps call in one pass (sn_p2i_max_forward_multi):
        returns out / ids of shape [len(radii), batch, channels, h, w], or, image_major,
        [batch, len(radii), channels, h, w] (radii <= 16 px) -- the memory layout of the
        [B, len(radius_list), S, S] tensor ComputeDepthMaps returns."""
        if kernel_kind != 0:
            raise ValueError("p2i: only kernel_kind 0 ('cos') exists")
        n, c, b, h, w = _shapes(points, point_features, background)
        nr = len(radii)
        zero_bg = isinstance(background, (tuple, list))   # a SHAPE instead of a tensor: an all-zero background
        shape = (b, nr, c, h, w) if image_major else (nr, b, c, h, w)
        out = torch.empty(shape, dtype=points.dtype, device=points.device)
        ids = torch.empty(out.shape, dtype=torch.int32, device=points.device)
        host_radii = (ctypes.c_float * nr)(*[float(r) for r in radii])
        if zero_bg and max(float(r) for r in radii) > 16.0:
            background, zero_bg = torch.zeros(b, c, h, w, dtype=points.dtype, device=points.device), False
        with torch.cuda.device_of(points):
            nbytes = _lib.lib().sn_p2i_max_multi_workspace_bytes(n, b, c, h, w)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=points.device)
            code = _lib.lib().sn_p2i_max_forward_multi(
                _lib.fptr(points, "points"), _lib.fptr(point_features, "point_features"),
                _lib.iptr(batch_inds, "batch_inds"),
                ctypes.c_void_p(0) if zero_bg else _lib.fptr(background, "background"),
                n, c, b, h, w, host_radii, nr, int(bool(image_major)), _lib.fptr(out, "out"),
                _lib.iptr(ids, "out_point_ids"), ctypes.c_void_p(ws.data_ptr()),
                ctypes.c_size_t(nbytes), _lib.stream_of(points))
        _lib.check(code, "sn_p2i_max_forward_multi")
        return out, ids

    @staticmethod
    def p2i_max_backward_gpu(out_grad, out_point_ids, points, point_features, kernel_kind,
                             kernel_radius, batch_inds=None):
        n = points.size(0)
        c = point_features.size(1)
        b, _, h, w = out_grad.shape
        points_grad = torch.empty_like(points)
        feat_grad = torch.empty_like(point_features)
        bg_grad = torch.empty_like(out_grad)
        if points.dtype == torch.float64:
            with torch.cuda.device_of(out_grad):
                code = _lib.lib().sn_p2i_max_backward_f64(
                    _lib.dptr(out_grad, "out_grad"), _lib.iptr(out_point_ids, "out_point_ids"),
                    _lib.dptr(points, "points"), _lib.dptr(point_features, "point_features"),
                    n, c, b, h, w, ctypes.c_double(kernel_radius), _lib.dptr(points_grad, "points_grad"),
                    _lib.dptr(feat_grad, "point_features_grad"), _lib.dptr(bg_grad, "background_grad"),
                    _lib.stream_of(out_grad))
            _lib.check(code, "sn_p2i_max_backward_f64")
            return points_grad, feat_grad, bg_grad
        with torch.cuda.device_of(out_grad):
            nbytes = _lib.lib().sn_p2i_max_backward_workspace_bytes(b, c, h, w)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=out_grad.device)
            code = _lib.lib().sn_p2i_max_backward(
                _lib.fptr(out_grad, "out_grad"), _lib.iptr(out_point_ids, "out_point_ids"),
                _lib.fptr(points, "points"), _lib.fptr(point_features, "point_features"),
                _lib.iptr(batch_inds, "batch_inds") if batch_inds is not None else ctypes.c_void_p(0),
                n, c, b, h, w, _lib.cfloat(kernel_radius), _lib.fptr(points_grad, "points_grad"),
                _lib.fptr(feat_grad, "point_features_grad"), _lib.fptr(bg_grad, "background_grad"),
                ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(nbytes), _lib.stream_of(out_grad))
        _lib.check(code, "sn_p2i_max_backward")
        return points_grad, feat_grad, bg_grad

    @staticmethod
    def p2i_max_backward_multi_gpu(out_grad, out_point_ids, points, point_features, kernel_kind,
                                   radii, image_major=False, want_background_grad=True):
        """out_grad / out_point_ids [len(radii), B, C, H, W] (image_major: [B, len(radii), C, H, W])
        -> gradients summed over the radii (sn_p2i_max_backward_multi: exact fixed-point
        accumulation, bit-reproducible)."""
        n = points.size(0)
        c = point_features.size(1)
        if image_major:
            b, nr, _, h, w = out_grad.shape
        else:
            nr, b, _, h, w = out_grad.shape
        points_grad = torch.empty_like(points)
        feat_grad = torch.empty_like(point_features)
        bg_grad = (torch.empty((b, c, h, w), dtype=out_grad.dtype, device=out_grad.device)
                   if want_background_grad else None)
        host_radii = (ctypes.c_float * nr)(*[float(r) for r in radii])
        with torch.cuda.device_of(out_grad):
            nbytes = _lib.lib().sn_p2i_max_backward_multi_workspace_bytes(n, c)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=out_grad.device)
            code = _lib.lib().sn_p2i_max_backward_multi(
                _lib.fptr(out_grad, "out_grad"), _lib.iptr(out_point_ids, "out_point_ids"),
                _lib.fptr(points, "points"), _lib.fptr(point_features, "point_features"),
                n, c, b, h, w, host_radii, nr, int(bool(image_major)), _lib.fptr(points_grad, "points_grad"),
                _lib.fptr(feat_grad, "point_features_grad"),
                _lib.fptr(bg_grad, "background_grad") if want_background_grad else ctypes.c_void_p(0),
                ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(nbytes), _lib.stream_of(out_grad))
        _lib.check(code, "sn_p2i_max_backward_multi")
        return points_grad, feat_grad, bg_grad

    @staticmethod
    def p2i_sum_forward_gpu(points, point_features, batch_inds, background, kernel_kind,
                            kernel_radius):
        if kernel_kind != 0:
            raise ValueError("p2i: only kernel_kind 0 ('cos') exists")
        n, c, b, h, w = _shapes(points, point_features, background)
        out = background.clone()
        if points.dtype == torch.float64:
            with torch.cuda.device_of(background):
                code = _lib.lib().sn_p2i_sum_forward_f64(
                    _lib.dptr(points, "points"), _lib.dptr(point_features, "point_features"),
                    _lib.iptr(batch_inds, "batch_inds"), n, c, b, h, w, ctypes.c_double(kernel_radius),
                    _lib.dptr(out, "out"), _lib.stream_of(background))
            _lib.check(code, "sn_p2i_sum_forward_f64")
            return out
        with torch.cuda.device_of(background):
            code = _lib.lib().sn_p2i_sum_forward(
                _lib.fptr(points, "points"), _lib.fptr(point_features, "point_features"),
                _lib.iptr(batch_inds, "batch_inds"), n, c, b, h, w, _lib.cfloat(kernel_radius),
                _lib.fptr(out, "out"), _lib.stream_of(background))
        _lib.check(code, "sn_p2i_sum_forward")
        return out

    @staticmethod
    def p2i_sum_backward_gpu(out_grad, points, point_features, batch_inds, kernel_kind,
                             kernel_radius):
        n = points.size(0)
        c = point_features.size(1)
        b, _, h, w = out_grad.shape
        points_grad = torch.empty_like(points)
        feat_grad = torch.empty_like(point_features)
        if points.dtype == torch.float64:
            with torch.cuda.device_of(out_grad):
                code = _lib.lib().sn_p2i_sum_backward_f64(
                    _lib.dptr(out_grad, "out_grad"), _lib.dptr(points, "points"),
                    _lib.dptr(point_features, "point_features"), _lib.iptr(batch_inds, "batch_inds"),
                    n, c, b, h, w, ctypes.c_double(kernel_radius), _lib.dptr(points_grad, "points_grad"),
                    _lib.dptr(feat_grad, "point_features_grad"), _lib.stream_of(out_grad))
            _lib.check(code, "sn_p2i_sum_backward_f64")
            return points_grad, feat_grad
        with torch.cuda.device_of(out_grad):
            code = _lib.lib().sn_p2i_sum_backward(
                _lib.fptr(out_grad, "out_grad"), _lib.fptr(points, "points"),
                _lib.fptr(point_features, "point_features"), _lib.iptr(batch_inds, "batch_inds"),
                n, c, b, h, w, _lib.cfloat(kernel_radius), _lib.fptr(points_grad, "points_grad"),
                _lib.fptr(feat_grad, "point_features_grad"), _lib.stream_of(out_grad))
        _lib.check(code, "sn_p2i_sum_backward")
        return points_grad, feat_grad


ext = _Ext()


def _c(*tensors):
    return tuple(t.contiguous() for t in tensors)


class P2ISumFunction(Function):
    """reduce="sum": out = background + sum_p w(p, pixel) * feature_p."""

    @staticmethod
    def forward(ctx, points, point_features, batch_inds, background, kernel_kind, kernel_radius):
        ctx.save_for_backward(points, point_features, batch_inds)
        ctx.kind_radius = (kernel_kind, kernel_radius)
        return ext.p2i_sum_forward_gpu(*_c(points, point_features, batch_inds, background),
                                       kernel_kind, kernel_radius)

    @staticmethod
    def backward(ctx, out_grad):
        points, point_features, batch_inds = ctx.saved_tensors
        g_points, g_feat = ext.p2i_sum_backward_gpu(
            *_c(out_grad, points, point_features, batch_inds), *ctx.kind_radius)
        # d out / d background is the identity (cuda/p2i_op/__init__.py:55)
        return g_points, g_feat, None, out_grad, None, None


class P2IMaxFunction(Function):
    """reduce="max": out = max(background, max_p w(p, pixel) * feature_p); the winner's
    point id per pixel is kept for the backward pass."""

    @staticmethod
    def forward(ctx, points, point_features, batch_inds, background, kernel_kind, kernel_radius):
        out, winner_ids = ext.p2i_max_forward_gpu(
            *_c(points, point_features, batch_inds, background), kernel_kind, kernel_radius)
        ctx.save_for_backward(points, point_features, winner_ids, batch_inds.contiguous())
        ctx.kind_radius = (kernel_kind, kernel_radius)
        return out

    @staticmethod
    def backward(ctx, out_grad):
        points, point_features, winner_ids, batch_inds = ctx.saved_tensors
        kind, radius = ctx.kind_radius
        if points.dtype == torch.float64:
            g_points, g_feat, g_bg = ext.p2i_max_backward_gpu(
                out_grad.contiguous(), winner_ids, *_c(points, point_features), kind, radius)
            return g_points, g_feat, None, g_bg, None, None
        g_points, g_feat, g_bg = ext.p2i_max_backward_multi_gpu(
            out_grad.contiguous().unsqueeze(0), winner_ids.unsqueeze(0),
            *_c(points, point_features), kind, [radius])
        return g_points, g_feat, None, g_bg, None, None


class P2IMaxMultiFunction(Function):
    """P2IMaxFunction for several kernel radii at once: returns [len(radii), B, C, H, W], slice
    r equal to P2IMaxFunction.apply(..., radii[r]) -- or, image_major, [B, len(radii), C, H, W]
    (written in that layout by the kernel: no transpose, and the gradient arrives contiguous).
    `background` may be a shape (B, C, H, W) instead of a tensor: an all-zero background that is never
    allocated, filled or read (ComputeDepthMaps' case).
    The forward shares one binning and one pixel walk between the radii; the backward adds the
    radii's gradients in one pass."""

    @staticmethod
    def forward(ctx, points, point_features, batch_inds, background, kernel_kind, radii, image_major=False):
        native = bool(image_major) and max(float(r) for r in radii) <= 16.0   # the kernel writes [B,R,...] itself
        bg = background if isinstance(background, (tuple, list)) else background.contiguous()
        out, winner_ids = ext.p2i_max_forward_multi_gpu(
            *_c(points, point_features, batch_inds), bg, kernel_kind, radii, native)
        ctx.save_for_backward(points, point_features, winner_ids, batch_inds.contiguous())
        ctx.kind_radii = (kernel_kind, tuple(float(r) for r in radii), native, bool(image_major) and not native)
        return out.transpose(0, 1).contiguous() if (image_major and not native) else out

    @staticmethod
    def backward(ctx, out_grad):
        points, point_features, winner_ids, batch_inds = ctx.saved_tensors
        kind, radii, native, transposed = ctx.kind_radii
        if transposed:
            out_grad = out_grad.transpose(0, 1)
        g_points, g_feat, g_bg = ext.p2i_max_backward_multi_gpu(
            out_grad.contiguous(), winner_ids, *_c(points, point_features), kind, radii, native,
            want_background_grad=ctx.needs_input_grad[3])
        return g_points, g_feat, None, g_bg, None, None, None


_kernel_kind_dict = {"cos": 0}
_REDUCERS = {"sum": P2ISumFunction, "max": P2IMaxFunction}


def p2i(points, point_features, batch_inds, background, kernel_radius, kernel_kind_str="cos",
        reduce="sum"):
    """Splat per-point features onto images.

    points          float [npoints, 2] in normalised (row, col) coordinates; (-1,-1) and
                    (+1,+1) are opposite image corners
    point_features  float [npoints, channels]
    batch_inds      int32 [npoints], image index of every point (out-of-range ids are skipped)
    background      float [batch, channels, out_h, out_w]
    kernel_radius   footprint radius in PIXELS; kernel_kind_str: only "cos"
    reduce          "sum" or "max"
    returns         float [batch, channels, out_h, out_w]
    Same contract as the reference's p2i (cuda/p2i_op/__init__.py:99-131).
    """
    kind = _kernel_kind_dict[kernel_kind_str]
    if reduce not in _REDUCERS:
        raise RuntimeError(f"Invalid reduce value: {reduce}")
    out_h, out_w = background.shape[2:]
    extent = points.new_tensor([out_h - 1, out_w - 1]).view(1, 2)
    pixel_points = (points + 1) / 2 * extent
    return _REDUCERS[reduce].apply(pixel_points, point_features, batch_inds, background, kind,
                                   kernel_radius)


custom_fun = P2ISumFunction.apply
