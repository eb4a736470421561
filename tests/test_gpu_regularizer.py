"""Fused surface regularizers (HIP) vs the oracle restatement of the reference's allmap
post-processing + depth_to_normal + normal / distortion losses, and render()'s own torch maps."""

import numpy as np
import pytest
import torch

from oracle import regularizer_ref as R

pytestmark = pytest.mark.gpu


def _allmap_from_render(dev, n=20000, w=320, h=240, seed=0):
    from gaussmart_amd import gaussian_renderer
    from gaussmart_amd.gaussian_model import GaussianModel
    from gaussmart_amd.params import PipelineParams
    from gaussmart_amd.synthetic import make_scene, jittered_cameras
    params, _ = make_scene(n, w, h, seed=seed)
    cam = jittered_cameras(2, w, h, seed=seed, device=dev)[1]          # a rotated / translated view
    m = GaussianModel(3, device=dev)
    m.create_from_params(params)
    with torch.no_grad():
        pkg = gaussian_renderer.render(cam, m, PipelineParams(), torch.zeros(3, device=dev))
    return pkg, cam


@pytest.mark.parametrize("depth_ratio,ln,ld", [(0.0, 0.05, 0.0), (1.0, 0.05, 1000.0), (0.3, 0.0, 100.0), (0.5, 1.0, 1.0)])
def test_forward_backward_vs_oracle(gpu_device, depth_ratio, ln, ld):
    from gaussmart_amd.fused_regularizer import surface_regularizer
    pkg, cam = _allmap_from_render(gpu_device)
    am = pkg["allmap"].detach()
    amh = am.clone().requires_grad_(True)
    loss, nm, dm = surface_regularizer(amh, cam, depth_ratio, ln, ld)
    (2.5 * loss).backward()
    amo = am.cpu().double().requires_grad_(True)
    lo, nmo, dmo = R.regularizer_loss(amo, cam.world_view_transform.cpu().double(), cam.full_proj_transform.cpu().double(),
                                      depth_ratio, ln, ld)
    (2.5 * lo).backward()
    np.testing.assert_allclose(nm.item(), nmo.item(), rtol=2e-5)
    np.testing.assert_allclose(dm.item(), dmo.item(), rtol=2e-5)
    np.testing.assert_allclose(loss.item(), lo.item(), rtol=2e-5, atol=1e-9)
    gh, go = amh.grad.cpu().double(), torch.nan_to_num(amo.grad, 0.0, 0.0, 0.0)
    for c in range(7):
        sc = float(go[c].abs().max())
        d = (gh[c] - go[c]).abs()
        # fp32 cross products of nearly parallel finite differences lose digits on a few pixels
        assert float(d.max()) <= 2e-3 * sc + 1e-12, (c, float(d.max()), sc)
        assert float(d.mean()) <= 2e-5 * sc + 1e-14, (c, float(d.mean()), sc)


def test_full_frame_tile_order(gpu_device):
    """1920x1080 (8,160 tiles of 16x16 in the XCD-aware order, band boundaries inside tile rows) against the oracle."""
    from gaussmart_amd.fused_regularizer import surface_regularizer
    pkg, cam = _allmap_from_render(gpu_device, n=60000, w=1920, h=1080, seed=5)
    am = pkg["allmap"].detach()
    amh = am.clone().requires_grad_(True)
    loss, nm, dm = surface_regularizer(amh, cam, 0.0, 0.05, 100.0)
    loss.backward()
    amo = am.cpu().double().requires_grad_(True)
    lo, nmo, dmo = R.regularizer_loss(amo, cam.world_view_transform.cpu().double(), cam.full_proj_transform.cpu().double(),
                                      0.0, 0.05, 100.0)
    lo.backward()
    np.testing.assert_allclose(nm.item(), nmo.item(), rtol=2e-5)
    np.testing.assert_allclose(dm.item(), dmo.item(), rtol=2e-5)
    gh, go = amh.grad.cpu().double(), torch.nan_to_num(amo.grad, 0.0, 0.0, 0.0)
    for c in range(7):
        sc = float(go[c].abs().max())
        d = (gh[c] - go[c]).abs()
        assert float(d.max()) <= 2e-3 * sc + 1e-12, (c, float(d.max()), sc)
        assert float(d.mean()) <= 2e-5 * sc + 1e-14, (c, float(d.mean()), sc)


def test_matches_render_maps(gpu_device):
    """Same numbers as render()'s own (torch) surf_normal / rend_normal path on the device."""
    from gaussmart_amd.fused_regularizer import surface_regularizer
    pkg, cam = _allmap_from_render(gpu_device, seed=3)
    normal_error = (1 - (pkg["rend_normal"] * pkg["surf_normal"]).sum(dim=0))
    loss, nm, dm = surface_regularizer(pkg["allmap"], cam, 0.0, 0.05, 2.0)
    np.testing.assert_allclose(nm.item(), normal_error.mean().item(), rtol=2e-5)
    np.testing.assert_allclose(dm.item(), pkg["rend_dist"].mean().item(), rtol=2e-5)


def test_empty_image_and_odd_sizes(gpu_device):
    from gaussmart_amd.fused_regularizer import surface_regularizer
    from gaussmart_amd.synthetic import jittered_cameras
    cam = jittered_cameras(1, 37, 21, device=gpu_device)[0]
    am = torch.zeros(7, 21, 37, device=gpu_device, requires_grad=True)     # alpha = 0 everywhere: 0/0 depth
    loss, nm, dm = surface_regularizer(am, cam, 0.0, 0.05, 1.0)
    loss.backward()
    assert abs(nm.item() - 1.0) < 1e-6 and dm.item() == 0.0
    assert torch.isfinite(am.grad).all() and float(am.grad[:6].abs().max()) == 0.0
    np.testing.assert_allclose(am.grad[6].cpu().numpy(), 1.0 / (21 * 37), rtol=1e-6)


@pytest.mark.parametrize("ln,ld", [(0.05, 0.0), (0.05, 100.0), (0.0, 0.0)])
def test_fused_objective_vs_oracles(gpu_device, ln, ld):
    """One autograd node for L1 + SSIM + regularizers: value and both gradients vs the two oracles."""
    from gaussmart_amd.fused_objective import training_objective
    from oracle import loss_ref
    pkg, cam = _allmap_from_render(gpu_device, seed=5)
    img = pkg["render"].detach()
    gt = (img + 0.1 * torch.randn_like(img)).clamp(0, 1)
    am = pkg["allmap"].detach()
    ih, ah = img.clone().requires_grad_(True), am.clone().requires_grad_(True)
    total, parts = training_objective(ih, ah, gt, cam, 0.2, ln, ld, 0.0)
    total.backward()
    io, ao = img.cpu().double().requires_grad_(True), am.cpu().double().requires_grad_(True)
    lo, l1o, so = loss_ref.photometric_loss(io, gt.cpu().double(), 0.2)
    ro, nmo, dmo = R.regularizer_loss(ao, cam.world_view_transform.cpu().double(), cam.full_proj_transform.cpu().double(), 0.0, ln, ld)
    (lo + ro).backward()
    np.testing.assert_allclose(total.item(), (lo + ro).item(), rtol=2e-5)
    np.testing.assert_allclose(parts[0].item(), l1o.item(), rtol=2e-5)
    np.testing.assert_allclose(parts[1].item(), so.item(), rtol=2e-5)
    if ln > 0 or ld > 0:
        np.testing.assert_allclose(parts[2].item(), nmo.item(), rtol=2e-5)
        np.testing.assert_allclose(parts[3].item(), dmo.item(), rtol=2e-5)
    gi, gio = ih.grad.cpu().double(), io.grad
    assert float((gi - gio).abs().max()) <= 2e-4 * float(gio.abs().max())
    if ln > 0 or ld > 0:
        ga, gao = ah.grad.cpu().double(), torch.nan_to_num(ao.grad, 0.0, 0.0, 0.0)
        for c in range(7):
            sc = float(gao[c].abs().max())
            assert float((ga[c] - gao[c]).abs().max()) <= 2e-3 * sc + 1e-12
    else:
        assert ah.grad is None or float(ah.grad.abs().max()) == 0.0


@pytest.mark.parametrize("ln,ld", [(0.05, 100.0), (0.0, 0.0)])
def test_objective_value_written_during_the_backward(gpu_device, ln, ld):
    """training_objective(defer_value=True): the five scalars come from a workgroup riding along with the backward's first
    kernel (gsr_loss_backward_finish) instead of a launch of their own -- same values, same gradients, bit for bit."""
    from gaussmart_amd.fused_objective import training_objective
    pkg, cam = _allmap_from_render(gpu_device, seed=2)
    img, am = pkg["render"].detach(), pkg["allmap"].detach()
    gt = (img * 0.9 + 0.05).clamp(0, 1).contiguous()
    res = []
    for defer in (False, True):
        a, b = img.clone().requires_grad_(True), am.clone().requires_grad_(True)
        total, parts = training_objective(a, b, gt, cam, 0.2, ln, ld, 0.0, defer_value=defer)
        total.backward()
        torch.cuda.synchronize()
        res.append((total.detach().clone(), parts.clone(), a.grad.clone(), None if b.grad is None else b.grad.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    assert (res[0][3] is None) == (res[1][3] is None) and (res[0][3] is None or torch.equal(res[0][3], res[1][3]))
    assert float(res[1][0]) > 0.0
