"""Fused surface regularizers on the HIP library.

    surface_regularizer(allmap, viewpoint_camera, depth_ratio, lambda_normal, lambda_dist)
        -> (loss, normal_error_mean, dist_mean)                      # 0-dim device tensors
    loss = lambda_normal * mean(1 - rend_normal . surf_normal) + lambda_dist * mean(rend_dist)

replaces, for training, the chain allmap -> rend_normal / surf_depth / depth_to_normal -> normal
and distortion losses (gaussian_renderer/__init__.py:117-156, utils/point_utils.py:9-37,
train.py:132-140 of the reference).  render() still produces those maps for callers that want them.
"""
import ctypes as C

import torch

from . import _lib


def camera_kinv(view):
    """Inverse pixel intrinsics as the reference derives them (utils/point_utils.py:10-17), cached
    on the camera as 9 host floats (one device->host copy per camera, ever)."""
    cached = getattr(view, "_gsr_kinv", None)
    if cached is not None and cached[0] == (view.image_width, view.image_height):
        return cached[1]
    wvt = view.world_view_transform.detach().float().cpu()
    full = view.full_proj_transform.detach().float().cpu()
    W, H = view.image_width, view.image_height
    c2w = wvt.T.inverse()
    ndc2pix = torch.tensor([[W / 2, 0, 0, W / 2], [0, H / 2, 0, H / 2], [0, 0, 0, 1]], dtype=torch.float32).T
    intrins = ((c2w.T @ full) @ ndc2pix)[:3, :3].T
    kinv = (C.c_float * 9)(*intrins.inverse().reshape(-1).tolist())
    try:
        view._gsr_kinv = ((W, H), kinv)
    except Exception:
        pass
    return kinv


class _SurfaceRegularizer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, allmap, kinv, depth_ratio, lambda_normal, lambda_dist):
        L = _lib.lib()
        if allmap.device.type != "cuda":
            raise _lib.GsrError("surface_regularizer needs a tensor on a HIP device (torch 'cuda'); there is no CPU path")
        am = allmap.detach().float().contiguous()
        if am.dim() != 3 or am.shape[0] != 7:
            raise ValueError("allmap must be [7,H,W]")
        _, H, W = am.shape
        dev = am.device
        with torch.cuda.device(dev):
            partials = torch.empty((L.gsr_loss_num_partials(H, W) // 2, 2), dtype=torch.float32, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(L.gsr_regularizer_forward(C.c_void_p(am.data_ptr()), H, W, kinv, float(depth_ratio),
                                                 C.c_void_p(partials.data_ptr()), C.c_void_p(stream)))
        means = partials.sum(0) / float(H * W)
        normal_mean, dist_mean = means[0], means[1]
        loss = lambda_normal * normal_mean + lambda_dist * dist_mean
        ctx.save_for_backward(am)
        ctx.cfg = (kinv, float(depth_ratio), float(lambda_normal), float(lambda_dist))
        ctx.mark_non_differentiable(normal_mean, dist_mean)
        return loss, normal_mean, dist_mean

    @staticmethod
    def backward(ctx, g_loss, _g1, _g2):
        L = _lib.lib()
        (am,) = ctx.saved_tensors
        kinv, depth_ratio, ln, ld = ctx.cfg
        _, H, W = am.shape
        dev = am.device
        with torch.cuda.device(dev):
            scale = g_loss.detach().float().reshape(1).contiguous()
            dam = torch.empty_like(am)
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(L.gsr_regularizer_backward(C.c_void_p(am.data_ptr()), H, W, kinv, depth_ratio, ln, ld,
                                                  C.c_void_p(scale.data_ptr()), C.c_void_p(dam.data_ptr()),
                                                  C.c_void_p(stream)))
        return dam, None, None, None, None


def surface_regularizer(allmap, viewpoint_camera, depth_ratio=0.0, lambda_normal=0.05, lambda_dist=0.0):
    return _SurfaceRegularizer.apply(allmap, camera_kinv(viewpoint_camera), depth_ratio, lambda_normal, lambda_dist)
