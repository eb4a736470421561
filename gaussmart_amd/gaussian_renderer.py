"""render(): the reference's L2 boundary, signature kept verbatim
(gaussian_renderer/__init__.py:19 of alevalve/gaussmart):

    render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None) -> dict
    keys: render, viewspace_points, visibility_filter, radii, rend_alpha, rend_normal, rend_dist,
          surf_depth, surf_normal   (+ "allmap", the rasterizer's raw [7,H,W] side output)

The extra keyword `surface_maps=False` skips the five derived maps: the trainer's fast path feeds
`allmap` to the fused regularizer kernels instead (gaussmart_amd/fused_regularizer.py).
`factored_sh_grad=True` (only honoured on the raw-parameter path; set by trainer.training_step, which owns the
optimiser step) makes the backward leave the colour gradient [N,3] instead of the two SH gradient tensors.
`color_only=True` (raw-parameter path with `surface_maps=False`; set by trainer.training_step while no regularizer is
active, i.e. before iteration 7,000 or with lambda_normal = lambda_dist = 0): "allmap" comes back as None.
`no_dist_median=True` (same path; set while lambda_dist = 0 and depth_ratio = 0, the reference's defaults): channels 5
(median depth) and 6 (distortion) of "allmap" come back as zeros -- nobody reads them in that configuration.

Device follows the model's tensors (the reference hard-codes "cuda").
"""
import math

import torch

from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians_raw
from .sh import eval_sh

_HipRasterizer = GaussianRasterizer   # tests may monkeypatch `GaussianRasterizer`; the raw path is HIP-only


def _camera_rays(view, device, dtype):
    """Per-pixel ray directions / origin in world space (utils/point_utils.py:9-24), cached on the
    camera because they only depend on it."""
    cache = getattr(view, "_gsr_rays", None)
    key = (str(device), dtype, view.image_width, view.image_height)
    if cache is not None and cache[0] == key:
        return cache[1], cache[2]
    wvt = view.world_view_transform.to(device=device, dtype=dtype)
    c2w = wvt.T.inverse()
    W, H = view.image_width, view.image_height
    ndc2pix = torch.tensor([[W / 2, 0, 0, W / 2], [0, H / 2, 0, H / 2], [0, 0, 0, 1]],
                           dtype=dtype, device=device).T
    proj = c2w.T @ view.full_proj_transform.to(device=device, dtype=dtype)
    intrins = (proj @ ndc2pix)[:3, :3].T
    gx, gy = torch.meshgrid(torch.arange(W, device=device, dtype=dtype),
                            torch.arange(H, device=device, dtype=dtype), indexing="xy")
    pix = torch.stack([gx, gy, torch.ones_like(gx)], dim=-1).reshape(-1, 3)
    rays_d = pix @ intrins.inverse().T @ c2w[:3, :3].T
    rays_o = c2w[:3, 3]
    try:
        view._gsr_rays = (key, rays_d, rays_o)
    except Exception:
        pass
    return rays_d, rays_o


def depths_to_points(view, depthmap):
    rays_d, rays_o = _camera_rays(view, depthmap.device, depthmap.dtype)
    return depthmap.reshape(-1, 1) * rays_d + rays_o


def depth_to_normal(view, depth):
    """Finite-difference normal of the back-projected depth map (utils/point_utils.py:26-37)."""
    points = depths_to_points(view, depth).reshape(*depth.shape[1:], 3)
    output = torch.zeros_like(points)
    dx = points[2:, 1:-1] - points[:-2, 1:-1]
    dy = points[1:-1, 2:] - points[1:-1, :-2]
    output[1:-1, 1:-1, :] = torch.nn.functional.normalize(torch.cross(dx, dy, dim=-1), dim=-1)
    return output


class _LazyRenderPkg(dict):
    def __missing__(self, key):
        if key == "visibility_filter":
            self[key] = self["radii"] > 0
            return self[key]
        raise KeyError(key)

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default


_ZERO_POINTS = {}


def _zero_leaf(xyz):
    """A fresh autograd leaf of zeros shaped like xyz.  Nothing ever reads or writes its VALUES (the rasterizer only
    fills its .grad), so on a HIP device all leaves alias one cached zero buffer instead of being filled every call."""
    if not xyz.is_cuda:
        return torch.zeros_like(xyz, requires_grad=True)
    key = (tuple(xyz.shape), xyz.dtype, xyz.device)
    z = _ZERO_POINTS.get(key)
    if z is None:
        _ZERO_POINTS.clear()          # one shape at a time (N changes with densification)
        z = _ZERO_POINTS[key] = torch.zeros(xyz.shape, dtype=xyz.dtype, device=xyz.device)
    return z.detach().requires_grad_(True)


def _use_raw_path(pc, pipe, override_color, xyz):
    """Opt-in (pipe.fused_activations) fast path: only when the model exposes the raw parameter
    tensors with the reference's activations (exp / sigmoid / normalize) and nothing is overridden."""
    if not getattr(pipe, "fused_activations", False) or override_color is not None or not xyz.is_cuda:
        return False
    if getattr(pipe, "compute_cov3D_python", False) or GaussianRasterizer is not _HipRasterizer:
        return False
    need = ("_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
    if not all(hasattr(pc, a) for a in need):
        return False
    return (getattr(pc, "scaling_activation", None) is torch.exp and getattr(pc, "opacity_activation", None) is torch.sigmoid
            and getattr(pc, "rotation_activation", None) is torch.nn.functional.normalize)


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None, *,
           surface_maps=True, factored_sh_grad=False, color_only=False, no_dist_median=False):
    xyz = pc.get_xyz
    device = xyz.device
    # the reference adds 0 and calls retain_grad() (gaussian_renderer/__init__.py:27-31); a leaf that requires
    # grad receives the same .grad without the extra add and the gradient clone of retain_grad
    screenspace_points = _zero_leaf(xyz)

    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=tanfovx, tanfovy=tanfovy, bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree, campos=viewpoint_camera.camera_center,
        prefiltered=False, debug=False)
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)

    raw = _use_raw_path(pc, pipe, override_color, xyz)
    # the model's hand-over slots with the optimiser step (rasterizer.RasterState): a pipelined step may still be updating
    # the SH tensors on a side stream -- the raw forward orders its colour pass behind that itself, everything else that
    # reads the parameters (the torch activations below) waits here
    state = getattr(pc, "raster_state", None)
    if state is not None and not raw and xyz.is_cuda:
        state.wait_pending(device)
    means3D, means2D = xyz, screenspace_points
    opacity = None if raw else pc.get_opacity
    scales = rotations = cov3D_precomp = None
    if getattr(pipe, "compute_cov3D_python", False):
        # T matrix assembled in Python, rows (Tu, Tv, Tw) flattened: gaussian_renderer/__init__.py:62-75
        splat2world = pc.get_covariance(scaling_modifier)
        W, H = viewpoint_camera.image_width, viewpoint_camera.image_height
        near, far = viewpoint_camera.znear, viewpoint_camera.zfar
        ndc2pix = torch.tensor([[W / 2, 0, 0, (W - 1) / 2], [0, H / 2, 0, (H - 1) / 2],
                                [0, 0, far - near, near], [0, 0, 0, 1]], dtype=torch.float32, device=device).T
        world2pix = viewpoint_camera.full_proj_transform @ ndc2pix
        cov3D_precomp = (splat2world[:, [0, 1, 3]] @ world2pix[:, [0, 1, 3]]).permute(0, 2, 1).reshape(-1, 9)
    elif not raw:
        scales, rotations = pc.get_scaling, pc.get_rotation

    shs = colors_precomp = None
    if override_color is None:
        if getattr(pipe, "convert_SHs_python", False) and False:   # the reference forces this off (:82)
            shs_view = pc.get_features.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
            d = xyz - viewpoint_camera.camera_center.repeat(pc.get_features.shape[0], 1)
            colors_precomp = torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, d / d.norm(dim=1, keepdim=True)) + 0.5, 0.0)
        elif not raw:
            shs = pc.get_features
    else:
        colors_precomp = override_color

    if raw:
        # same kernels, activations + dc|rest concatenation fused inside (rasterizer.py: _RasterizeGaussiansRaw)
        factored = factored_sh_grad and pc._features_rest.shape[1] > 0
        # the optimiser step may have left the SH colour of exactly this view for exactly these parameters
        # (FusedAdam.color_cache, gsr_adam_sh_factored_next): then the forward skips its SH colour pass
        cache = None
        lookup = getattr(getattr(pc, "optimizer", None), "lookup_color_cache", None)
        if factored and lookup is not None and torch.is_grad_enabled():
            cache = lookup(viewpoint_camera.camera_center, pc.active_sh_degree, pc._xyz, pc._features_dc, pc._features_rest)
        rendered_image, radii, allmap = rasterize_gaussians_raw(
            xyz, means2D, pc._features_dc, pc._features_rest, pc._opacity, pc._scaling, pc._rotation, raster_settings,
            factored_sh_grad=factored and state is not None, color_cache=cache if state is not None else None, state=state,
            color_only=bool(color_only) and not surface_maps and torch.is_grad_enabled(),
            no_dist_median=bool(no_dist_median) and not surface_maps and torch.is_grad_enabled())
    else:
        rendered_image, radii, allmap = rasterizer(
            means3D=means3D, means2D=means2D, shs=shs, colors_precomp=colors_precomp, opacities=opacity,
            scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)

    if not surface_maps:
        # fused-objective path: "visibility_filter" (radii > 0) is computed on first access instead of every call
        return _LazyRenderPkg({"render": rendered_image, "viewspace_points": means2D, "radii": radii, "allmap": allmap})
    rets = {"render": rendered_image, "viewspace_points": means2D, "visibility_filter": radii > 0, "radii": radii,
            "allmap": allmap}

    render_alpha = allmap[1:2]
    if allmap.is_cuda and getattr(pipe, "fused_surface_maps", False):
        # the same five maps from one HIP launch each way (fused_surface_maps.py) instead of the ~40 torch kernels below
        from .fused_surface_maps import surface_maps as _fused_maps
        render_normal, surf_depth, surf_normal = _fused_maps(allmap, viewpoint_camera, pipe.depth_ratio)
        rets.update({"rend_alpha": render_alpha, "rend_normal": render_normal, "rend_dist": allmap[6:7],
                     "surf_depth": surf_depth, "surf_normal": surf_normal})
        return rets
    # view-space normals -> world space
    render_normal = allmap[2:5]
    render_normal = (render_normal.permute(1, 2, 0) @ (viewpoint_camera.world_view_transform[:3, :3].T)).permute(2, 0, 1)
    render_depth_median = torch.nan_to_num(allmap[5:6], 0, 0)
    render_depth_expected = torch.nan_to_num(allmap[0:1] / render_alpha, 0, 0)
    render_dist = allmap[6:7]
    # depth_ratio 1 = median depth (bounded scenes), 0 = expected depth (unbounded)
    surf_depth = render_depth_expected * (1 - pipe.depth_ratio) + pipe.depth_ratio * render_depth_median
    surf_normal = depth_to_normal(viewpoint_camera, surf_depth).permute(2, 0, 1)
    surf_normal = surf_normal * render_alpha.detach()   # render_normal is un-normalised too

    rets.update({"rend_alpha": render_alpha, "rend_normal": render_normal, "rend_dist": render_dist,
                 "surf_depth": surf_depth, "surf_normal": surf_normal})
    return rets
