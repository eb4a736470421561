"""Inference path (render.py / view.py / GaussianExtractor.reconstruction run render() under torch.no_grad():
utils/mesh_utils.py:100-123, view.py:15-31) and per-model hand-over state.

  * under no_grad (or when no input requires grad) the operator takes GSR_FLAG_FORWARD_ONLY: no touch words, no
    per-pixel state, no autograd node -- the images and radii must be BIT-identical to the training forward;
  * the factored-gradient record and the pending-update event are kept per model, so two models trained in an
    interleaved fashion on one device behave exactly as if each were trained alone."""
import pytest
import torch

from conftest import hip_settings
from gaussmart_amd.synthetic import make_scene, activate, perturb, jittered_cameras

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,w,h,seed", [(3000, 256, 200, 0), (200_000, 1237, 822, 1)])
def test_no_grad_forward_is_bit_identical(gpu_device, n, w, h, seed):
    from gaussmart_amd import rasterizer as R
    dev = gpu_device
    p, cam = make_scene(n, w, h, seed=seed)
    a = {k: v.to(dev) for k, v in activate(p).items()}
    rs = hip_settings(cam, 3, (0.1, 0.2, 0.3), dev)
    m2d = torch.zeros(n, 3, device=dev)
    ins = {k: v.clone().requires_grad_(True) for k, v in a.items()}
    c1, r1, am1 = R.GaussianRasterizer(rs)(means3D=ins["means3D"], means2D=m2d, shs=ins["shs"], opacities=ins["opacities"],
                                           scales=ins["scales"], rotations=ins["rotations"])
    assert c1.requires_grad and c1.grad_fn is not None
    with torch.no_grad():
        c2, r2, am2 = R.GaussianRasterizer(rs)(means3D=ins["means3D"], means2D=m2d, shs=ins["shs"], opacities=ins["opacities"],
                                               scales=ins["scales"], rotations=ins["rotations"])
    assert not c2.requires_grad and c2.grad_fn is None
    assert torch.equal(c1, c2) and torch.equal(am1, am2) and torch.equal(r1, r2)
    # no input requires grad: forward-only as well, even with grad mode on
    c3, r3, am3 = R.GaussianRasterizer(rs)(means3D=a["means3D"], means2D=m2d, shs=a["shs"], opacities=a["opacities"],
                                           scales=a["scales"], rotations=a["rotations"])
    assert c3.grad_fn is None and torch.equal(c1, c3) and torch.equal(am1, am3)
    # precomputed colours
    col = torch.rand(n, 3, device=dev)
    with torch.no_grad():
        c4, _, am4 = R.GaussianRasterizer(rs)(means3D=a["means3D"], means2D=m2d, colors_precomp=col, opacities=a["opacities"],
                                              scales=a["scales"], rotations=a["rotations"])
    c5, _, am5 = R.GaussianRasterizer(rs)(means3D=ins["means3D"], means2D=m2d, colors_precomp=col, opacities=ins["opacities"],
                                          scales=ins["scales"], rotations=ins["rotations"])
    assert torch.equal(c4, c5) and torch.equal(am4, am5)
    # the training forward still backpropagates afterwards (the pooled buffers of the inference calls were recycled)
    (c1.square().sum() + am1.sum()).backward()
    assert all(torch.isfinite(v.grad).all() for v in ins.values())


def test_render_under_no_grad_matches_training_render(gpu_device):
    """render() on a GaussianModel (raw-parameter path, fused activations) the way render.py calls it."""
    from gaussmart_amd.gaussian_model import GaussianModel
    from gaussmart_amd.gaussian_renderer import render
    from gaussmart_amd.params import PipelineParams
    dev = gpu_device
    params, _ = make_scene(50_000, 640, 400, seed=2)
    cam = jittered_cameras(2, 640, 400, seed=2, device=dev)[1]
    m = GaussianModel(3, device=dev)
    m.create_from_params(params)
    pipe, bg = PipelineParams(), torch.zeros(3, device=dev)
    pkg = render(cam, m, pipe, bg)
    with torch.no_grad():
        pkg2 = render(cam, m, pipe, bg)
    for k in ("render", "allmap", "radii", "rend_alpha", "rend_normal", "surf_depth", "surf_normal", "rend_dist"):
        assert torch.equal(pkg[k].detach(), pkg2[k]), k
    assert pkg["render"].grad_fn is not None and pkg2["render"].grad_fn is None


def test_two_models_on_one_device_do_not_share_state(gpu_device):
    """Interleaved training of two models (forward A, forward B, backward A, backward B, step A, step B) with the
    factored SH gradient must equal training each one alone, bit for bit."""
    from gaussmart_amd.gaussian_model import GaussianModel
    from gaussmart_amd.gaussian_renderer import render
    from gaussmart_amd.params import OptimizationParams, PipelineParams
    from gaussmart_amd.trainer import training_losses, optimizer_step
    dev = gpu_device
    w, h = 320, 200
    pipe, opt, bg = PipelineParams(), OptimizationParams(), torch.zeros(3, device=dev)
    cams = jittered_cameras(2, w, h, seed=3, device=dev)
    scenes = [make_scene(6000, w, h, seed=10)[0], make_scene(9000, w, h, seed=11)[0]]
    gts = []
    for sc, cam in zip(scenes, cams):
        t = GaussianModel(3, device=dev)
        t.create_from_params(perturb(sc))
        with torch.no_grad():
            gts.append(render(cam, t, pipe, bg)["render"].clamp(0, 1).contiguous())

    def fresh():
        ms = []
        for sc in scenes:
            m = GaussianModel(3, device=dev)
            m.create_from_params(sc)
            m.training_setup(opt)
            ms.append(m)
        return ms

    def fwd_bwd_parts(m, cam, gt, it):
        m.update_learning_rate(it)
        pkg = render(cam, m, pipe, bg, surface_maps=False, factored_sh_grad=True)
        total, _ = training_losses(pkg, gt, opt, it, cam, pipe)
        return total

    def finish(m):
        rec = m.raster_state.take_color_grad()
        assert rec is not None and rec.n == m._xyz.shape[0]
        m.optimizer.park_sh_gradient(m._features_dc, m._features_rest, rec)
        optimizer_step(m)

    alone = fresh()
    for it in range(8000, 8004):
        for m, cam, gt in zip(alone, cams, gts):
            fwd_bwd_parts(m, cam, gt, it).backward()
            finish(m)
    mixed = fresh()
    for it in range(8000, 8004):
        totals = [fwd_bwd_parts(m, cam, gt, it) for m, cam, gt in zip(mixed, cams, gts)]
        for t in totals:
            t.backward()
        for m in mixed:
            finish(m)
    torch.cuda.synchronize()
    for ma, mb in zip(alone, mixed):
        for pa, pb in zip(ma.parameters(), mb.parameters()):
            assert torch.equal(pa, pb)
