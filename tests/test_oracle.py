"""The oracle itself: its backward is the derivative of its forward (gradcheck, fp64, quirks
off), the quirk flags only change the backward, its binning is a stable sort, and the operator
surface mirrors the reference's argument checks.  PARITY UNPINNED against upstream (no fixtures
exist for the rasterizer, see oracle/surfel_ref.py header)."""
import numpy as np
import pytest
import torch

from conftest import oracle_settings
from gaussmart_amd.synthetic import make_scene, activate
from oracle import surfel_ref as O


def _inputs(n, w, h, seed, dtype, **kw):
    p, cam = make_scene(n, w, h, seed=seed, dtype=dtype, **kw)
    return activate(p), cam


def test_gradcheck_fp64_no_quirks():
    torch.manual_seed(0)
    a, cam = _inputs(16, 32, 32, 3, torch.float64, radius_px=5)
    S = oracle_settings(cam, 3, torch.float64, bg=(0.3, 0.5, 0.7))
    wc = torch.randn(3, 32, 32, dtype=torch.float64)
    wa = torch.randn(7, 32, 32, dtype=torch.float64)

    def f(means3D, opac, shs, scales, rots):
        c, _, am = O.rasterize(means3D, torch.zeros_like(means3D), opac, shs, None, scales,
                               torch.nn.functional.normalize(rots), None, settings=S, flags=0)
        return (c * wc).sum() + (am * wa).sum()

    ins = [a[k].detach().clone().requires_grad_(True) for k in ("means3D", "opacities", "shs", "scales", "rotations")]
    assert torch.autograd.gradcheck(f, ins, eps=1e-6, atol=1e-5, rtol=1e-4)


def test_gradcheck_precomputed_paths():
    torch.manual_seed(1)
    a, cam = _inputs(10, 32, 32, 5, torch.float64, radius_px=6)
    S = oracle_settings(cam, 0, torch.float64)
    geom = O.preprocess(a["means3D"], a["scales"], a["rotations"], a["opacities"], a["shs"], None, None, S)
    T = torch.zeros(10, 9, dtype=torch.float64)
    T[geom.vis_idx] = geom.Tm.reshape(-1, 9)
    keep = torch.zeros(10, dtype=torch.bool); keep[geom.vis_idx] = True
    T[~keep] = torch.tensor([1., 0, 0, 0, 1, 0, 0, 0, 1], dtype=torch.float64)
    colors = torch.rand(10, 3, dtype=torch.float64)
    wc = torch.randn(3, 32, 32, dtype=torch.float64)

    def f(means3D, opac, col, cov):
        c, _, am = O.rasterize(means3D, torch.zeros_like(means3D), opac, None, col, None, None, cov, settings=S, flags=0)
        return (c * wc).sum() + am[0].sum() + am[6].sum()

    ins = [x.detach().clone().requires_grad_(True) for x in (a["means3D"], a["opacities"], colors, T)]
    assert torch.autograd.gradcheck(f, ins, eps=1e-6, atol=1e-5, rtol=1e-4)


def test_quirk_flags_change_backward_only():
    a, cam = _inputs(300, 64, 64, 0, torch.float64)
    a["opacities"] = torch.full_like(a["opacities"], 0.995)      # forces the 0.99 clamp
    S = oracle_settings(cam, 3, torch.float64)
    outs = {}
    for flags in (0, O.QUIRKS_UPSTREAM):
        ins = {k: v.clone().requires_grad_(True) for k, v in a.items()}
        c, r, am = O.rasterize(ins["means3D"], torch.zeros(300, 3, dtype=torch.float64), ins["opacities"], ins["shs"],
                               None, ins["scales"], ins["rotations"], None, settings=S, flags=flags)
        (c.sum() + am.sum()).backward()
        outs[flags] = (c.detach(), am.detach(), ins["opacities"].grad.clone(), ins["scales"].grad.clone())
    assert torch.equal(outs[0][0], outs[3][0]) and torch.equal(outs[0][1], outs[3][1])
    assert not torch.allclose(outs[0][2], outs[3][2])      # clamp pass-through reaches opacity
    assert not torch.allclose(outs[0][3], outs[3][3])      # filter-depth quirk reaches scales


def test_binning_is_stable_sort_of_64bit_keys():
    a, cam = _inputs(800, 128, 96, 2, torch.float32)
    S = oracle_settings(cam)
    geom = O.preprocess(a["means3D"], a["scales"], a["rotations"], a["opacities"], a["shs"], None, None, S)
    depth = np.zeros(800, np.float32)
    depth[geom.vis_idx.numpy()] = geom.depth.numpy()
    depth[geom.vis_idx.numpy()[:50]] = depth[geom.vis_idx.numpy()[0]]     # force depth ties
    gx, gy = 8, 6
    keys, plist = O.bin_tiles(None, geom.radii.numpy(), geom.rect.numpy(), depth, gx)
    assert np.all(keys[1:] >= keys[:-1])
    # ties keep emission order = ascending Gaussian index
    same = keys[1:] == keys[:-1]
    assert np.all(plist[1:][same] > plist[:-1][same])
    # every (Gaussian, tile) of every rect is present exactly once
    rect = geom.rect.numpy().astype(np.int64)
    expect = int(((rect[:, 2] - rect[:, 0]) * (rect[:, 3] - rect[:, 1]))[geom.radii.numpy() > 0].sum())
    assert keys.size == expect == np.unique(np.stack([keys >> np.uint64(32), plist.astype(np.uint64)]), axis=1).shape[1]
    ranges = O.tile_ranges(keys, gx * gy)
    for t in range(gx * gy):
        s, e = ranges[t]
        assert np.all((keys[s:e] >> np.uint64(32)) == t)
    assert int((ranges[:, 1] - ranges[:, 0]).sum()) == keys.size


def test_compositing_semantics_small():
    """Hand-checkable case: one opaque-ish surfel facing the camera at depth 4."""
    cam = make_scene(1, 32, 32)[1]
    S = oracle_settings(cam, 0, torch.float64, bg=(0.0, 0.0, 1.0))
    means = torch.tensor([[0.0, 0.0, 4.0]], dtype=torch.float64)
    scales = torch.tensor([[0.5, 0.5]], dtype=torch.float64)
    rots = torch.tensor([[1.0, 0, 0, 0]], dtype=torch.float64)
    opac = torch.tensor([[0.8]], dtype=torch.float64)
    col = torch.tensor([[1.0, 0.0, 0.0]], dtype=torch.float64)
    c, radii, am = O.rasterize(means, torch.zeros(1, 3, dtype=torch.float64), opac, None, col, scales, rots, None, settings=S)
    assert radii[0] > 0
    # centre pixel: alpha = 0.8 * exp(-rho/2) with tiny rho, colour = alpha*red + (1-alpha)*blue
    y, x = 16, 16
    alpha = float(am[1, y, x])
    assert 0.75 < alpha <= 0.8
    np.testing.assert_allclose(c[:, y, x].numpy(), [alpha, 0.0, 1 - alpha], atol=1e-12)
    np.testing.assert_allclose(float(am[0, y, x]) / alpha, 4.0, rtol=1e-6)       # expected depth
    np.testing.assert_allclose(float(am[5, y, x]), 4.0, rtol=1e-6)               # median depth (T=1 > 0.5)
    np.testing.assert_allclose(am[2:5, y, x].numpy() / alpha, [0, 0, -1], atol=1e-9)  # normal faces the camera
    assert float(am[6, y, x]) == 0.0                                             # single surfel: no distortion
    # a corner pixel far outside the 3-sigma box sees only background
    np.testing.assert_allclose(c[:, 0, 0].numpy(), [0, 0, 1], atol=1e-12)


def test_operator_argument_checks():
    a, cam = _inputs(5, 32, 32, 0, torch.float32)
    S = oracle_settings(cam)
    m2d = torch.zeros(5, 3)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        O.rasterize(a["means3D"], m2d, a["opacities"], None, None, a["scales"], a["rotations"], None, settings=S)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed"):
        O.rasterize(a["means3D"], m2d, a["opacities"], a["shs"], None, None, None, None, settings=S)


def test_means2d_grad_is_densification_statistic():
    a, cam = _inputs(200, 64, 64, 4, torch.float64)
    S = oracle_settings(cam, 3, torch.float64)
    m2d = torch.zeros(200, 3, dtype=torch.float64, requires_grad=True)
    ins = {k: v.clone().requires_grad_(True) for k, v in a.items()}
    c, radii, am = O.rasterize(ins["means3D"], m2d, ins["opacities"], ins["shs"], None, ins["scales"], ins["rotations"],
                               None, settings=S)
    c.sum().backward()
    assert torch.all(m2d.grad[:, 2] == 0)
    assert torch.all(m2d.grad[radii == 0] == 0)
    assert m2d.grad[radii > 0].abs().sum() > 0


def test_aabb_centre_quirk_keeps_the_forward_value():
    """QUIRK_AABB_GRAD_CUTOFF1 (third recalled non-derivative, include/gsr.h GSR_FLAG_AABB_GRAD_CUTOFF1): the screen-space
    centre keeps the value of the (9, 9, -1) weights; only its gradient is the one of the (1, 1, -1) form, which is the
    exact derivative of THAT form (checked against autograd of the form itself)."""
    a, cam = _inputs(200, 64, 64, 1, torch.float64)
    S = oracle_settings(cam, 3, torch.float64)
    g0 = O.preprocess(a["means3D"], a["scales"], a["rotations"], a["opacities"], a["shs"], None, None, S)
    leaves = {k: a[k].detach().clone().requires_grad_(True) for k in ("means3D", "scales", "rotations")}
    g1 = O.preprocess(leaves["means3D"], leaves["scales"], leaves["rotations"], a["opacities"], a["shs"], None, None, S,
                      O.QUIRK_AABB_GRAD_CUTOFF1)
    assert torch.equal(g0.xy, g1.xy.detach()) and torch.equal(g0.radii, g1.radii)
    w = torch.randn_like(g1.xy)
    grads = torch.autograd.grad((g1.xy * w).sum(), list(leaves.values()))
    # the (1, 1, -1) form written out on the same T rows
    leaves2 = {k: a[k].detach().clone().requires_grad_(True) for k in ("means3D", "scales", "rotations")}
    g2 = O.preprocess(leaves2["means3D"], leaves2["scales"], leaves2["rotations"], a["opacities"], a["shs"], None, None, S)
    Tu, Tv, Tw = g2.Tm[:, 0], g2.Tm[:, 1], g2.Tm[:, 2]
    t1 = torch.tensor([1.0, 1.0, -1.0], dtype=torch.float64)
    f1 = t1 / (t1 * Tw * Tw).sum(-1, keepdim=True)
    xy1 = torch.stack([(f1 * Tu * Tw).sum(-1), (f1 * Tv * Tw).sum(-1)], -1)
    ref = torch.autograd.grad((xy1 * w).sum(), list(leaves2.values()), retain_graph=True)
    for x, y in zip(grads, ref):
        assert torch.allclose(x, y, rtol=1e-10, atol=1e-12)
    # and it is NOT the derivative of the forward's own centre
    own = torch.autograd.grad((g2.xy * w).sum(), list(leaves2.values()))
    assert any(float((x - y).abs().max()) > 1e-9 for x, y in zip(grads, own))
