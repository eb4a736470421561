"""Factored SH gradient (GSR_FLAG_FACTORED_SH_GRAD + gsr_adam_sh_factored, include/gsr.h).

The SH colour of the reference is linear in the coefficients (utils/sh_utils.py:57-112), so the gradient of one view
is basis_k(dir) x g_c with g the clamp-masked dL/drgb.  Checked here, through the C ABI on the device:
  * the record the backward leaves rebuilds exactly the dL/dshs the unfactored backward writes;
  * the fused optimiser step from records equals Adam on the explicit gradients (one view: same bits up to the
    contraction of the basis polynomials; several views: against torch.optim.Adam on the summed gradient);
  * a training run with the factored step equals the unfactored one."""
import math

import pytest
import torch

from conftest import hip_settings

pytestmark = pytest.mark.gpu


def _raw_inputs(gpu_device, n=6000, w=256, h=160, seed=5):
    from gaussmart_amd.synthetic import make_scene
    params, cam = make_scene(n, w, h, seed=seed)
    p = {k: v.to(gpu_device).requires_grad_(True) for k, v in params.items()}
    return p, cam


def _backward(gpu_device, p, cam, factored, deg=3):
    from gaussmart_amd import rasterizer as R
    rs = hip_settings(cam, deg=deg, device=gpu_device)
    for v in p.values():
        v.grad = None
    m2d = torch.zeros_like(p["xyz"], requires_grad=True)
    state = R.RasterState()      # the per-model hand-over slots: the factored backward leaves its record there
    color, radii, allmap = R.rasterize_gaussians_raw(p["xyz"], m2d, p["features_dc"], p["features_rest"], p["opacity"],
                                                     p["scaling"], p["rotation"], rs, factored_sh_grad=factored, state=state)
    gen = torch.Generator().manual_seed(11)
    wc = torch.randn(color.shape, generator=gen).to(gpu_device)
    wa = torch.randn(allmap.shape, generator=gen).to(gpu_device) * 0.1
    ((color * wc).sum() + (allmap * wa).sum()).backward()
    return state.take_color_grad(), radii


def _sh_grad_from_record(xyz, record, n, deg, coeffs=16):
    """[N,coeffs,3] = basis(normalize(xyz - campos)) x g, float64."""
    from gaussmart_amd.sh import sh_basis
    g = record[:3 * n].view(n, 3).double()
    campos = record[3 * n:3 * n + 3].double()
    d = xyz.detach().double() - campos
    b = sh_basis(deg, d / d.norm(dim=1, keepdim=True))
    full = torch.zeros(n, coeffs, dtype=torch.float64, device=xyz.device)
    full[:, :b.shape[1]] = b
    return full[:, :, None] * g[:, None, :]


@pytest.mark.parametrize("deg", [3, 1, 0])
def test_record_rebuilds_the_sh_gradient(gpu_device, deg):
    p, cam = _raw_inputs(gpu_device)
    assert _backward(gpu_device, p, cam, False, deg)[0] is None
    ref = {k: v.grad.clone() for k, v in p.items()}
    rec, radii = _backward(gpu_device, p, cam, True, deg)
    n = p["xyz"].shape[0]
    assert rec is not None and rec.n == n and rec.record.numel() == 3 * n + 4
    assert p["features_dc"].grad is None and p["features_rest"].grad is None
    for k in ("opacity", "scaling", "rotation"):                   # the geometry gradients are the same kernels
        assert torch.equal(p[k].grad, ref[k]), k
    # dL/dxyz: its view-direction term g^T d(rgb)/d(dir) is formed from the 3x3 Jacobian the colour pass left (factored:
    # the SH coefficients are not read again) instead of from the coefficients themselves -- same sum, other order
    dx = (p["xyz"].grad - ref["xyz"]).abs().max().item()
    assert dx <= 1e-6 * ref["xyz"].abs().max().item() + 1e-12, dx
    assert torch.equal(rec.record[3 * n:3 * n + 3], cam.camera_center.to(gpu_device).float())
    g = rec.record[:3 * n].view(n, 3)
    assert torch.all(g[radii <= 0] == 0)                             # culled Gaussians carry no colour gradient
    assert (g != 0).any()
    sh = _sh_grad_from_record(p["xyz"], rec.record, n, deg)
    want = torch.cat([ref["features_dc"], ref["features_rest"]], dim=1).double()
    scale = want.abs().max().item()
    assert (sh - want).abs().max().item() <= 2e-6 * scale
    # views of one allocation: [xyz | opacity | scaling | rotation | record]
    assert rec.head.data_ptr() == p["xyz"].grad.data_ptr() and rec.flat.numel() == rec.head.numel() + 3 * n + 4


def _adam_setup(gpu_device, n, m=16, seed=0):
    from gaussmart_amd.fused_adam import FusedAdam
    gen = torch.Generator().manual_seed(seed)
    f_dc = torch.nn.Parameter(torch.randn(n, 1, 3, generator=gen).to(gpu_device))
    f_rest = torch.nn.Parameter((0.1 * torch.randn(n, m - 1, 3, generator=gen)).to(gpu_device))
    groups = [{"params": [f_dc], "lr": 0.0025, "name": "f_dc"}, {"params": [f_rest], "lr": 0.0025 / 20, "name": "f_rest"}]
    return f_dc, f_rest, FusedAdam(groups, lr=0.0, eps=1e-15), gen


@pytest.mark.parametrize("n,views,deg", [(5000, 1, 3), (4099, 3, 3), (777, 8, 2), (64, 2, 0)])
def test_factored_step_equals_adam_on_explicit_gradients(gpu_device, n, views, deg):
    f_dc, f_rest, opt, gen = _adam_setup(gpu_device, n)
    r_dc, r_rest = [torch.nn.Parameter(t.detach().clone()) for t in (f_dc, f_rest)]
    ref = torch.optim.Adam([{"params": [r_dc], "lr": 0.0025}, {"params": [r_rest], "lr": 0.0025 / 20}], lr=0.0, eps=1e-15)
    xyz = (torch.randn(n, 3, generator=gen) * 3).to(gpu_device)
    stride = 3 * n + 4
    scale = 1.0 / views
    for it in range(3):
        rec = torch.zeros(views, stride)
        rec[:, :3 * n] = torch.randn(views, 3 * n, generator=gen) * 1e-3
        rec[:, :3 * n].view(views, n, 3)[:, ::5] = 0.0               # some Gaussians unseen by a view
        rec[:, 3 * n:3 * n + 3] = torch.randn(views, 3, generator=gen) * 0.3 + torch.tensor([0.0, 0.0, -6.0])
        rec = rec.to(gpu_device).contiguous()
        grad = sum(_sh_grad_from_record(xyz, rec[r], n, deg) for r in range(views)) * scale
        r_dc.grad, r_rest.grad = grad[:, :1].float().contiguous(), grad[:, 1:].float().contiguous()
        ref.step()
        opt.step_sh_factored(f_dc, f_rest, xyz, rec.view(-1), views, stride, deg, scale)
    torch.cuda.synchronize()
    assert float(opt.state[f_rest]["step"]) == 3.0 and float(opt.state[f_dc]["step"]) == 3.0
    for a, b, lr in ((f_dc, r_dc, 0.0025), (f_rest, r_rest, 0.0025 / 20)):
        # Adam normalises the step to ~lr: compare in units of lr
        assert (a - b).abs().max().item() <= 2e-3 * lr
    for a, b in ((opt.state[f_rest]["exp_avg"], ref.state[r_rest]["exp_avg"]),
                 (opt.state[f_rest]["exp_avg_sq"], ref.state[r_rest]["exp_avg_sq"])):
        assert (a - b).abs().max().item() <= 1e-5 * b.abs().max().item()


def test_factored_step_in_ranges_equals_one_call(gpu_device):
    n, views, deg = 3001, 2, 3
    gen = torch.Generator().manual_seed(9)
    xyz = (torch.randn(n, 3, generator=gen) * 3).to(gpu_device)
    stride = 3 * n + 4
    rec = (torch.randn(views * stride, generator=gen) * 1e-2).to(gpu_device)
    outs = []
    for cuts in ([0, n], [0, 129, 1500, n]):                        # 129: a range that starts off the 16-byte grid
        f_dc, f_rest, opt, _ = _adam_setup(gpu_device, n, seed=3)
        for a, b in zip(cuts[:-1], cuts[1:]):
            opt.step_sh_factored(f_dc, f_rest, xyz, rec, views, stride, deg, 0.5, first=a, count=b - a, count_step=a == 0)
        assert float(opt.state[f_rest]["step"]) == 1.0
        outs.append((f_dc.detach().clone(), f_rest.detach().clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_factored_step_rejects_bad_arguments(gpu_device):
    from gaussmart_amd import _lib
    f_dc, f_rest, opt, _ = _adam_setup(gpu_device, 10)
    xyz = torch.zeros(10, 3, device=gpu_device)
    with pytest.raises(_lib.GsrError):
        opt.step_sh_factored(f_dc, f_rest, xyz, torch.zeros(10, device=gpu_device), 1, 34, 3)       # record too short
    with pytest.raises(_lib.GsrError):
        opt.step_sh_factored(f_dc, f_rest, xyz, torch.zeros(17 * 34, device=gpu_device), 17, 34, 3)  # > 16 views


def _train(gpu_device, factored, steps=5, sh_degree=3):
    from gaussmart_amd.gaussian_model import GaussianModel
    from gaussmart_amd.params import OptimizationParams, PipelineParams
    from gaussmart_amd.synthetic import jittered_cameras, make_scene
    from gaussmart_amd.trainer import training_step
    params, _ = make_scene(20001, 320, 200, seed=3, sh_degree=sh_degree)
    cams = jittered_cameras(steps, 320, 200, seed=1, device=gpu_device)
    gt = torch.rand(3, 200, 320, generator=torch.Generator().manual_seed(2)).to(gpu_device)
    opt, pipe, bg = OptimizationParams(), PipelineParams(factored_sh_grad=factored), torch.zeros(3, device=gpu_device)
    m = GaussianModel(sh_degree, device=gpu_device)
    m.create_from_params(params)
    m.training_setup(opt)
    losses = [training_step(m, cams[i], gt, opt, pipe, bg, 10000 + i)[1]["total"] for i in range(steps)]
    torch.cuda.synchronize()
    return m, [float(x) for x in losses]


@pytest.mark.parametrize("sh_degree", [3, 2, 1, 0])
def test_training_with_factored_step_equals_unfactored(gpu_device, sh_degree):
    """Models with 16, 9, 4 coefficients per channel; with 1 (degree 0) there is no features_rest and the explicit
    path is taken."""
    a, la = _train(gpu_device, True, sh_degree=sh_degree)
    b, lb = _train(gpu_device, False, sh_degree=sh_degree)
    assert la[0] == lb[0]
    assert max(abs(x - y) for x, y in zip(la, lb)) <= 1e-6 * abs(lb[0])
    for name, lr in (("_xyz", 1.6e-4), ("_opacity", 0.05), ("_scaling", 0.005), ("_rotation", 0.001),
                     ("_features_dc", 0.0025), ("_features_rest", 0.0025 / 20)):
        pa, pb = getattr(a, name), getattr(b, name)
        if pa.numel() == 0:
            continue
        assert (pa - pb).abs().max().item() <= 0.05 * lr, name      # 5 steps of size ~lr each
        assert (pa - pb).abs().mean().item() <= 1e-4 * lr, name
    assert a._features_rest.grad is None and a.optimizer.pending_sh is None
    assert a._features_rest.shape[1] == (sh_degree + 1) ** 2 - 1
