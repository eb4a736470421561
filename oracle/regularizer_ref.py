"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's surface regularizers:
allmap post-processing (gaussian_renderer/__init__.py:117-156), depths_to_points / depth_to_normal
(utils/point_utils.py:9-37, world space, as the reference) and the two losses (train.py:132-140).
No reference fixtures exist for these functions (point_utils imports cv2, absent here): parity
unpinned beyond the line-by-line restatement."""
import torch


def depths_to_points(world_view_transform, full_proj_transform, W, H, depthmap):
    dt = depthmap.dtype
    c2w = (world_view_transform.to(dt).T).inverse()
    ndc2pix = torch.tensor([[W / 2, 0, 0, W / 2], [0, H / 2, 0, H / 2], [0, 0, 0, 1]], dtype=dt).T
    projection_matrix = c2w.T @ full_proj_transform.to(dt)
    intrins = (projection_matrix @ ndc2pix)[:3, :3].T
    gx, gy = torch.meshgrid(torch.arange(W, dtype=dt), torch.arange(H, dtype=dt), indexing="xy")
    pts = torch.stack([gx, gy, torch.ones_like(gx)], dim=-1).reshape(-1, 3)
    rays_d = pts @ intrins.inverse().T @ c2w[:3, :3].T
    return depthmap.reshape(-1, 1) * rays_d + c2w[:3, 3]


def depth_to_normal(world_view_transform, full_proj_transform, W, H, depth):
    points = depths_to_points(world_view_transform, full_proj_transform, W, H, depth).reshape(*depth.shape[1:], 3)
    out = torch.zeros_like(points)
    dx = points[2:, 1:-1] - points[:-2, 1:-1]
    dy = points[1:-1, 2:] - points[1:-1, :-2]
    out[1:-1, 1:-1, :] = torch.nn.functional.normalize(torch.cross(dx, dy, dim=-1), dim=-1)
    return out


def surface_maps(allmap, world_view_transform, full_proj_transform, depth_ratio):
    _, H, W = allmap.shape
    dt = allmap.dtype
    alpha = allmap[1:2]
    rend_normal = (allmap[2:5].permute(1, 2, 0) @ (world_view_transform.to(dt)[:3, :3].T)).permute(2, 0, 1)
    med = torch.nan_to_num(allmap[5:6], 0, 0)
    exp = torch.nan_to_num(allmap[0:1] / alpha, 0, 0)
    surf_depth = exp * (1 - depth_ratio) + depth_ratio * med
    surf_normal = depth_to_normal(world_view_transform, full_proj_transform, W, H, surf_depth).permute(2, 0, 1)
    surf_normal = surf_normal * alpha.detach()
    return dict(rend_alpha=alpha, rend_normal=rend_normal, rend_dist=allmap[6:7], surf_depth=surf_depth,
                surf_normal=surf_normal)


def regularizer_loss(allmap, world_view_transform, full_proj_transform, depth_ratio, lambda_normal, lambda_dist):
    m = surface_maps(allmap, world_view_transform, full_proj_transform, depth_ratio)
    normal_error = (1 - (m["rend_normal"] * m["surf_normal"]).sum(dim=0))[None]
    nm, dm = normal_error.mean(), m["rend_dist"].mean()
    return lambda_normal * nm + lambda_dist * dm, nm, dm
