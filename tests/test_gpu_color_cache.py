"""Colour cache (GSR_FLAG_COLOR_CACHED, gsr_adam_sh_factored_next; include/gsr.h): the factored SH optimiser step has every
Gaussian's new coefficients on chip, so it also evaluates the SH colour of the NEXT view (rgb, clamp mask, d(rgb)/d(dir));
the next forward copies it into its records instead of running the SH colour pass and the backward takes the Jacobian
from it.  Reference semantics that must not change: utils/sh_utils.py:57-112 + gaussian_renderer/__init__.py:86-91
(colour = clamp_min(eval_sh + 0.5, 0)), train.py:104-144 (one forward + backward + Adam step per iteration)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(dev, n=20000, w=320, h=200, seed=3):
    from gaussmart_amd.gaussian_model import GaussianModel
    from gaussmart_amd.gaussian_renderer import render
    from gaussmart_amd.params import OptimizationParams, PipelineParams
    from gaussmart_amd.synthetic import make_scene, perturb, jittered_cameras
    params, _ = make_scene(n, w, h, seed=seed)
    cams = jittered_cameras(3, w, h, seed=seed, device=dev, amount=0.3)
    pipe, opt, bg = PipelineParams(), OptimizationParams(), torch.zeros(3, device=dev)
    tgt = GaussianModel(3, device=dev)
    tgt.create_from_params(perturb(params))
    with torch.no_grad():
        gts = [render(c, tgt, pipe, bg)["render"].clamp(0, 1).contiguous() for c in cams]

    def fresh():
        m = GaussianModel(3, device=dev)
        m.create_from_params(params)
        m.training_setup(opt)
        return m
    return fresh, cams, gts, pipe, opt, bg


def test_cached_colour_equals_the_colour_pass(gpu_device):
    """One step with next_cam: the cache is valid for exactly that view; the forward that uses it produces the image of
    the ordinary colour pass (same formulas, other summation order) and the same gradients."""
    from gaussmart_amd.gaussian_renderer import render
    from gaussmart_amd.trainer import training_step
    fresh, cams, gts, pipe, opt, bg = _setup(gpu_device)
    m = fresh()
    training_step(m, cams[0], gts[0], opt, pipe, bg, 10000, next_cam=cams[1])
    o = m.optimizer
    look = lambda cam, deg=None: o.lookup_color_cache(cam.camera_center, m.active_sh_degree if deg is None else deg, m._xyz,
                                                      m._features_dc, m._features_rest)
    assert look(cams[1]) is not None and look(cams[0]) is None and look(cams[2]) is None and look(cams[1], deg=2) is None

    def fwd_bwd(use_cache):
        for p in m.parameters():
            p.grad = None
        cache = look(cams[1]) if use_cache else None
        assert (cache is not None) == use_cache
        saved = o.color_cache
        if not use_cache:
            o.color_cache = None                     # render() looks it up itself
        pkg = render(cams[1], m, pipe, bg, surface_maps=False, factored_sh_grad=True)
        ((pkg["render"] - gts[1]).square().sum() + pkg["allmap"][0].sum() * 1e-3).backward()
        rec = m.raster_state.take_color_grad()
        o.color_cache = saved
        return pkg["render"].detach().clone(), pkg["allmap"].detach().clone(), [p.grad.clone() for p in m.parameters()
                                                                                 if p.grad is not None], rec.record.clone()
    c0, a0, g0, r0 = fwd_bwd(False)
    c1, a1, g1, r1 = fwd_bwd(True)
    assert float((c0 - c1).abs().max()) < 2e-6 and torch.equal(a0, a1)      # geometry channels do not involve the colour
    for x, y in zip(g0, g1):
        assert float((x - y).abs().max()) <= 1e-5 * float(x.abs().max()) + 1e-12
    assert float((r0 - r1).abs().max()) <= 1e-5 * float(r0.abs().max())
    # an in-place edit of a parameter through torch invalidates the cache
    with torch.no_grad():
        m._features_dc.mul_(1.0)
    assert look(cams[1]) is None


@pytest.mark.parametrize("overlap", [False, True])
def test_training_with_colour_cache_tracks_training_without(gpu_device, overlap):
    """Ten iterations over three alternating views with the next view announced (cache used on every iteration but the
    first) against the same run without: same trajectory up to the rounding of the colour sums."""
    from gaussmart_amd.trainer import training_step
    from gaussmart_amd.view_parallel import ViewParallel
    from gaussmart_amd import rasterizer as R
    fresh, cams, gts, pipe, opt, bg = _setup(gpu_device)
    runs = []
    for use in (False, True):
        m = fresh()
        vp = ViewParallel(m, overlap_local=True) if overlap else None
        used = 0
        losses = []
        for it in range(10):
            cam, gt = cams[it % 3], gts[it % 3]
            nxt = cams[(it + 1) % 3] if use else None
            if m.optimizer.lookup_color_cache(cam.camera_center, m.active_sh_degree, m._xyz, m._features_dc,
                                              m._features_rest) is not None:
                used += 1
            _, parts = training_step(m, cam, gt, opt, pipe, bg, 10000 + it, view_parallel=vp, next_cam=nxt)
            losses.append(float(parts["total"]))
        if vp is not None:
            vp.finish()
        torch.cuda.synchronize()
        assert used == (9 if use else 0)
        runs.append(([p.detach().clone() for p in m.parameters()], losses))
    lrs = (1.6e-4, 0.0025, 0.0025 / 20, 0.05, 0.005, 0.001)
    for x, y, lr in zip(runs[0][0], runs[1][0], lrs):
        assert float((x - y).abs().max()) <= 0.2 * lr * 10 and float((x - y).abs().mean()) <= 2e-3 * lr * 10
    for a, b in zip(runs[0][1], runs[1][1]):
        assert abs(a - b) <= 1e-4 * abs(a)
    assert runs[1][1][-1] < runs[1][1][0]


def test_optimiser_writes_retire_the_cache(gpu_device):
    """The optimiser's kernels write parameters through raw pointers (no torch version bump): a step WITHOUT a next view, a
    partial step and a slice step must each retire a cache an earlier step built, or a later forward of that view would
    render colours of parameters that are steps old."""
    from gaussmart_amd.gaussian_renderer import render
    from gaussmart_amd.trainer import training_step
    fresh, cams, gts, pipe, opt, bg = _setup(gpu_device, n=5000)
    m = fresh()
    o = m.optimizer
    look = lambda cam: o.lookup_color_cache(cam.camera_center, m.active_sh_degree, m._xyz, m._features_dc, m._features_rest)
    training_step(m, cams[0], gts[0], opt, pipe, bg, 10000, next_cam=cams[0])
    assert look(cams[0]) is not None
    training_step(m, cams[0], gts[0], opt, pipe, bg, 10001, next_cam=None)       # parameters move, no new cache
    assert look(cams[0]) is None
    with torch.no_grad():                                                         # the forward runs the ordinary colour pass
        a = render(cams[0], m, pipe, bg)["render"].clone()
        o.color_cache = None
        b = render(cams[0], m, pipe, bg)["render"]
    assert torch.equal(a, b)
    # partial and slice updates retire it too
    training_step(m, cams[0], gts[0], opt, pipe, bg, 10002, next_cam=cams[1])
    assert look(cams[1]) is not None
    m._xyz.grad = torch.zeros_like(m._xyz)
    o.step(only=[m._xyz])
    assert look(cams[1]) is None
    training_step(m, cams[0], gts[0], opt, pipe, bg, 10003, next_cam=cams[1])
    assert look(cams[1]) is not None
    m._opacity.grad = torch.zeros_like(m._opacity)
    o.step_slice(m._opacity, 0, 10)
    assert look(cams[1]) is None
    training_step(m, cams[0], gts[0], opt, pipe, bg, 10004, next_cam=cams[1])
    assert look(cams[1]) is not None
    o.invalidate_color_cache()
    assert look(cams[1]) is None
