"""render()'s post-processing of allmap (gaussian_renderer/__init__.py:117-156 of the reference) as ONE autograd node on the HIP
library -- for callers that want the reference's dictionary (rend_normal, surf_depth, surf_normal as tensors) and its own
objective, not the fused one (fused_objective.py):

    rend_normal, surf_depth, surf_normal = surface_maps(allmap, viewpoint_camera, depth_ratio)

Same values and gradients as the torch formulation in gaussian_renderer.render() (tests/test_gpu_surface_maps.py), one launch
each way instead of ~40 elementwise / GEMM / cross / normalize kernels over the frame.  rend_alpha = allmap[1:2] and rend_dist =
allmap[6:7] stay plain slices.  At pixels with alpha = 0 torch's own backward leaves 0 / 0 on allmap[0:2]; this node writes 0
there (no splat covers such a pixel; the rasterizer's backward never reads its gradient).
"""
import ctypes as C

import torch

from . import _lib


def camera_world_rays(view):
    """(c2w[:3,:3] K^-1, world_view[:3,:3]) as two arrays of 9 host floats, cached on the camera: ray(x, y) = first * [x, y, 1]
    is the reference's rays_d (utils/point_utils.py:9-24), the second rotates view-space normals to world space."""
    cached = getattr(view, "_gsr_world_rays", None)
    if cached is not None and cached[0] == (view.image_width, view.image_height):
        return cached[1], cached[2]
    wvt = view.world_view_transform.detach().double().cpu()
    full = view.full_proj_transform.detach().double().cpu()
    W, H = view.image_width, view.image_height
    c2w = wvt.T.inverse()
    ndc2pix = torch.tensor([[W / 2, 0, 0, W / 2], [0, H / 2, 0, H / 2], [0, 0, 0, 1]], dtype=torch.float64).T
    intrins = ((c2w.T @ full) @ ndc2pix)[:3, :3].T
    m = c2w[:3, :3] @ intrins.inverse()
    rays = (C.c_float * 9)(*m.reshape(-1).tolist())
    rot = (C.c_float * 9)(*wvt[:3, :3].reshape(-1).tolist())
    try:
        view._gsr_world_rays = ((W, H), rays, rot)
    except Exception:
        pass
    return rays, rot


class _SurfaceMaps(torch.autograd.Function):
    @staticmethod
    def forward(ctx, allmap, rays, rot, depth_ratio):
        L = _lib.lib()
        if allmap.device.type != "cuda":
            raise _lib.GsrError("surface_maps needs a tensor on a HIP device (torch 'cuda'); there is no CPU path")
        am = allmap.detach().float().contiguous()
        if am.dim() != 3 or am.shape[0] != 7:
            raise ValueError("allmap must be [7,H,W]")
        _, H, W = am.shape
        dev = am.device
        with torch.cuda.device(dev):
            out = torch.empty((7, H, W), dtype=torch.float32, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(L.gsr_surface_maps_forward(C.c_void_p(am.data_ptr()), H, W, rays, rot, float(depth_ratio),
                                                  C.c_void_p(out.data_ptr()), C.c_void_p(stream)))
        ctx.save_for_backward(am)
        ctx.cfg = (rays, rot, float(depth_ratio))
        ctx.set_materialize_grads(False)
        return out[0:3], out[3:4], out[4:7]

    @staticmethod
    def backward(ctx, g_rn, g_sd, g_sn):
        if g_rn is None and g_sd is None and g_sn is None:
            return None, None, None, None
        L = _lib.lib()
        (am,) = ctx.saved_tensors
        rays, rot, depth_ratio = ctx.cfg
        _, H, W = am.shape
        dev = am.device
        with torch.cuda.device(dev):
            gout = torch.zeros((7, H, W), dtype=torch.float32, device=dev) if (g_rn is None or g_sd is None or g_sn is None) \
                else torch.empty((7, H, W), dtype=torch.float32, device=dev)
            if g_rn is not None:
                gout[0:3].copy_(g_rn)
            if g_sd is not None:
                gout[3:4].copy_(g_sd)
            if g_sn is not None:
                gout[4:7].copy_(g_sn)
            dam = torch.empty_like(am)
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(L.gsr_surface_maps_backward(C.c_void_p(am.data_ptr()), H, W, rays, rot, depth_ratio,
                                                   C.c_void_p(gout.data_ptr()), C.c_void_p(dam.data_ptr()), C.c_void_p(stream)))
        return dam, None, None, None


def surface_maps(allmap, viewpoint_camera, depth_ratio=0.0):
    """-> (rend_normal [3,H,W] world space, surf_depth [1,H,W], surf_normal [3,H,W] world space, times alpha.detach())."""
    rays, rot = camera_world_rays(viewpoint_camera)
    return _SurfaceMaps.apply(allmap, rays, rot, depth_ratio)
