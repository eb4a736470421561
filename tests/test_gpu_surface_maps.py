"""fused_surface_maps.surface_maps (one HIP launch each way) against the torch formulation render() mirrors from the reference
(gaussian_renderer/__init__.py:117-156, utils/point_utils.py:9-37): the five maps and the gradients of a random scalar of them,
on drawn image sizes, depth ratios and allmaps with holes; and through render() itself (PipelineParams.fused_surface_maps)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _torch_maps(allmap, cam, depth_ratio):
    from gaussmart_amd.gaussian_renderer import depth_to_normal
    render_alpha = allmap[1:2]
    render_normal = (allmap[2:5].permute(1, 2, 0) @ (cam.world_view_transform[:3, :3].T)).permute(2, 0, 1)
    render_depth_median = torch.nan_to_num(allmap[5:6], 0, 0)
    render_depth_expected = torch.nan_to_num(allmap[0:1] / render_alpha, 0, 0)
    surf_depth = render_depth_expected * (1 - depth_ratio) + depth_ratio * render_depth_median
    surf_normal = depth_to_normal(cam, surf_depth).permute(2, 0, 1) * render_alpha.detach()
    return render_normal, surf_depth, surf_normal


@pytest.mark.parametrize("seed", list(range(20)))
def test_maps_and_gradients_match_torch(gpu_device, seed):
    from gaussmart_amd.fused_surface_maps import surface_maps
    from gaussmart_amd.synthetic import jittered_cameras
    dev = gpu_device
    g = torch.Generator().manual_seed(600 + seed)
    W, H = [(3, 3), (2, 7), (16, 16), (17, 33), (64, 40), (97, 5), (130, 71), (33, 150), (48, 48), (250, 19)][seed % 10]
    ratio = [0.0, 0.0, 0.3, 1.0][seed % 4]
    cam = jittered_cameras(3, W, H, seed=seed, device=dev, amount=0.4)[seed % 3]
    alpha = torch.rand(1, H, W, generator=g)
    if seed % 2:
        alpha = torch.where(torch.rand(1, H, W, generator=g) < 0.3, torch.zeros_like(alpha), alpha)          # holes
    depth = 0.5 + 19.5 * torch.rand(1, H, W, generator=g)
    nrm = torch.randn(3, H, W, generator=g) * alpha
    am = torch.cat([alpha * depth, alpha, nrm, torch.where(alpha > 0, depth * (0.8 + 0.4 * torch.rand(1, H, W, generator=g)),
                                                           torch.zeros_like(depth)), alpha * torch.rand(1, H, W, generator=g)]).to(dev)
    w = [torch.randn(3, H, W, generator=g).to(dev), torch.randn(1, H, W, generator=g).to(dev), torch.randn(3, H, W, generator=g).to(dev)]
    res = []
    for fn in (lambda a: surface_maps(a, cam, ratio), lambda a: _torch_maps(a.double(), _Double(cam), ratio)):
        a = am.clone().requires_grad_(True)
        maps = fn(a)
        sum((m * wi.to(m.dtype)).sum() for m, wi in zip(maps, w)).backward()
        res.append(([m.detach().double() for m in maps], a.grad.double()))
    (mh, gh), (mt, gt) = res
    for name, x, y in zip(("rend_normal", "surf_depth", "surf_normal"), mh, mt):
        assert float((x - y).abs().max()) <= 2e-5 * max(float(y.abs().max()), 1.0), (seed, name)
    lit = am[1] > 0
    gt = torch.nan_to_num(gt, 0.0, 0.0, 0.0)
    assert bool(torch.isfinite(gh).all())
    for c in range(7):
        sc = float(gt[c][lit].abs().max()) if bool(lit.any()) else 0.0
        err = float((gh[c] - gt[c])[lit].abs().max()) if bool(lit.any()) else 0.0
        assert err <= 2e-4 * sc + 1e-9, (seed, c, err, sc)
    assert float(gh[6].abs().max()) == 0.0          # rend_dist is a plain slice of allmap, outside this node


class _Double:
    """The camera with its matrices in fp64 (the torch side of the comparison runs in double)."""
    def __init__(self, cam):
        self.world_view_transform = cam.world_view_transform.double()
        self.full_proj_transform = cam.full_proj_transform.double()
        self.image_width, self.image_height = cam.image_width, cam.image_height


def test_render_with_fused_maps_equals_render_with_torch_maps(gpu_device):
    from gaussmart_amd.gaussian_model import GaussianModel
    from gaussmart_amd.gaussian_renderer import render
    from gaussmart_amd.params import OptimizationParams, PipelineParams
    from gaussmart_amd.synthetic import jittered_cameras, make_scene
    from gaussmart_amd.trainer import training_losses
    dev = gpu_device
    params, _ = make_scene(300, 192, 112, seed=4, radius_px=5.0)            # a scene with uncovered pixels
    cam = jittered_cameras(2, 192, 112, seed=4, device=dev, amount=0.2)[1]
    opt, bg = OptimizationParams(), torch.zeros(3, device=dev)
    gt = torch.rand(3, 112, 192, device=dev)
    grads, pkgs = [], []
    for fused in (False, True):
        pipe = PipelineParams()
        pipe.fused_surface_maps, pipe.reference_objective, pipe.depth_ratio = fused, True, 0.3
        m = GaussianModel(3, device=dev)
        m.create_from_params(params)
        pkg = render(cam, m, pipe, bg)
        opt.lambda_dist = 100.0
        total, _ = training_losses(pkg, gt, opt, 8000, cam, pipe)
        total.backward()
        grads.append([p.grad.clone() for p in m.parameters()])
        pkgs.append({k: pkg[k].detach() for k in ("rend_alpha", "rend_normal", "rend_dist", "surf_depth", "surf_normal")})
    for k in pkgs[0]:
        assert float((pkgs[0][k] - pkgs[1][k]).abs().max()) <= 2e-5 * max(float(pkgs[0][k].abs().max()), 1.0), k
    for a, b in zip(*grads):
        assert bool(torch.isfinite(b).all())
        assert float((a - b).abs().max()) <= 2e-4 * float(a.abs().max()) + 1e-12
