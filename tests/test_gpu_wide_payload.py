"""Wide per-pixel payloads: colors_precomp [N,C], C = 4..64, composited by the same
kernels as RGB (template FEAT16 of render_fwd / render_bwd) and checked against the oracle, which is
channel-count agnostic.  Tolerances as in test_gpu_rasterizer.py (fp32 kernels vs fp64 oracle)."""
import pytest
import torch

from conftest import facing_scene, hip_settings
from gaussmart_amd.synthetic import activate
from oracle_farm import FARM, spec, check_gradient_bars
from test_gpu_rasterizer import _grad_compare

pytestmark = pytest.mark.gpu


def _wide_inputs(n, w, h, C, seed):
    p, cam = facing_scene(n, w, h, seed=seed)
    a = activate(p)
    g = torch.Generator().manual_seed(100 + seed)
    b = dict(means3D=a["means3D"], opacities=a["opacities"], scales=a["scales"], rotations=a["rotations"],
             colors_precomp=torch.randn(n, C, generator=g))
    bg = tuple(float(x) for x in torch.rand(C, generator=g))
    return b, cam, bg


# (the oracle side of every case runs in a worker process from the start of the session: tests/oracle_farm.py)
WIDE = {C: FARM.register(f"wide-C{C}-{n}@{w}x{h}", spec("facing", n, w, h, C, wide=(C, 100 + C), sens_tols=(1e-3,)))
        for C, n, w, h in ((4, 1500, 200, 120), (16, 2000, 256, 256), (24, 800, 130, 70), (64, 1200, 192, 160))}
WIDE_BIG = {C: FARM.register(f"wide-screen-filling-C{C}", spec("facing", 300, 160, 128, 12, radius_px=45.0, opa_const=0.05, wide=(C, C), sens_tols=(1e-3,)))
            for C in (8, 24)}


@pytest.mark.oracle_cases(*WIDE.values())
@pytest.mark.parametrize("C", sorted(WIDE))
def test_wide_payload_forward_backward_parity(gpu_device, C):
    sp = FARM.specs[WIDE[C]]
    stats, stats32, c_h, c_o, res = _grad_compare(WIDE[C], gpu_device)
    assert c_h.shape == (C, sp["h"], sp["w"])
    scale = float(c_o.abs().max())
    assert float((c_h - c_o).abs().max()) < 5e-3 * max(1.0, scale)
    assert float((c_h - c_o).abs().median()) < 1e-5
    check_gradient_bars(WIDE[C], stats, stats32, flips=res["flips"])


def test_wide_payload_first_three_channels_equal_rgb_path(gpu_device):
    """Channels 0..2 of a wide payload must be bit-identical to the RGB kernel fed the same three
    columns (same pairs, same order, same fp32 operations); all gradients agree to rounding (the wide
    kernel adds the colour term of q, and the sub-rows of the feature gradient, in a different order)."""
    from gaussmart_amd.rasterizer import GaussianRasterizer
    b, cam, bg = _wide_inputs(3000, 320, 200, 8, seed=5)
    dev = gpu_device
    W, H = cam.image_width, cam.image_height
    g = torch.Generator().manual_seed(9)
    wc, wa = torch.randn(3, H, W, generator=g).to(dev), torch.randn(7, H, W, generator=g).to(dev)

    def run(cols, bgv):
        ins = {k: b[k].clone().to(dev).requires_grad_(True) for k in ("means3D", "opacities", "scales", "rotations")}
        col = cols.clone().to(dev).requires_grad_(True)
        m2d = torch.zeros(3000, 3, device=dev, requires_grad=True)
        rast = GaussianRasterizer(hip_settings(cam, 3, bgv, dev), flags=3)
        c, r, am = rast(means3D=ins["means3D"], means2D=m2d, colors_precomp=col, opacities=ins["opacities"],
                        scales=ins["scales"], rotations=ins["rotations"])
        ((c[:3] * wc).sum() + (am * wa).sum()).backward()
        torch.cuda.synchronize()
        return c.detach(), am.detach(), r, {k: v.grad for k, v in ins.items()}, col.grad, m2d.grad

    c8, am8, r8, g8, gc8, m8 = run(b["colors_precomp"], bg)
    c3, am3, r3, g3, gc3, m3 = run(b["colors_precomp"][:, :3].contiguous(), bg[:3])
    assert torch.equal(c8[:3], c3) and torch.equal(am8, am3) and torch.equal(r8, r3)
    # feature gradients are summed over the sub-rows in a different (fixed) order than the RGB columns
    assert float((gc8[:, :3] - gc3).abs().max()) < 1e-5 * float(gc3.abs().max())
    assert float(gc8[:, 3:].abs().max()) == 0.0
    g8["means2D"], g3["means2D"] = m8, m3
    for k in g3:
        sc = float(g3[k].abs().max())
        assert float((g8[k] - g3[k]).abs().max()) < 1e-4 * sc, k


def test_wide_payload_is_bitwise_deterministic_and_rejects_bad_shapes(gpu_device):
    from gaussmart_amd import _lib
    from gaussmart_amd.rasterizer import GaussianRasterizer
    b, cam, bg = _wide_inputs(1000, 160, 96, 16, seed=2)
    dev = gpu_device

    def run(cols, bgv):
        col = cols.clone().to(dev).requires_grad_(True)
        m2d = torch.zeros(cols.shape[0], 3, device=dev, requires_grad=True)
        rast = GaussianRasterizer(hip_settings(cam, 3, bgv, dev), flags=3)
        c, r, am = rast(means3D=b["means3D"].to(dev), means2D=m2d, colors_precomp=col, opacities=b["opacities"].to(dev),
                        scales=b["scales"].to(dev), rotations=b["rotations"].to(dev))
        (c.square().sum() + am.sum()).backward()
        return c.detach().clone(), col.grad.clone()

    c1, g1 = run(b["colors_precomp"], bg)
    c2, g2 = run(b["colors_precomp"], bg)
    assert torch.equal(c1, c2) and torch.equal(g1, g2)
    with pytest.raises(_lib.GsrError):          # 6 is not a multiple of 4
        run(b["colors_precomp"][:, :6].contiguous(), bg[:6])
    with pytest.raises(ValueError):             # one background value per channel
        run(b["colors_precomp"], bg[:3])


def test_wide_payload_nothing_visible(gpu_device):
    """All surfels mirrored behind the camera: the image is the per-channel background, gradients are zero."""
    from gaussmart_amd.rasterizer import GaussianRasterizer
    b, cam, bg = _wide_inputs(64, 96, 64, 8, seed=3)
    dev = gpu_device
    w2v = cam.world_view_transform.cpu()
    fwd = w2v[:3, 2]
    z = b["means3D"] @ fwd + w2v[3, 2]
    means = b["means3D"] - 2.0 * z[:, None] * fwd[None, :]
    col = b["colors_precomp"].clone().to(dev).requires_grad_(True)
    m2d = torch.zeros(64, 3, device=dev, requires_grad=True)
    rast = GaussianRasterizer(hip_settings(cam, 3, bg, dev), flags=3)
    c, r, am = rast(means3D=means.to(dev), means2D=m2d, colors_precomp=col, opacities=b["opacities"].to(dev),
                    scales=b["scales"].to(dev), rotations=b["rotations"].to(dev))
    c.sum().backward()
    torch.cuda.synchronize()
    assert int((r > 0).sum()) == 0
    assert torch.equal(c, torch.tensor(bg, device=dev).view(-1, 1, 1).expand_as(c).contiguous())
    assert float(col.grad.abs().max()) == 0.0 and float(am.detach().abs().max()) == 0.0


@pytest.mark.oracle_cases(*WIDE_BIG.values())
@pytest.mark.parametrize("C", sorted(WIDE_BIG))
def test_wide_payload_screen_filling_splats(gpu_device, C):
    """Splats with thousands of gradient rows each: the feature-row reduction hands them to the whole workgroup
    (reduce_feat_rows_kernel, like reduce_rows); checked against the oracle with faint splats, so every one of them is
    blended far down the lists.  C = 24: 6 pieces per Gaussian, which does not divide the 256-thread workgroup."""
    stats, stats32, c_h, c_o, _ = _grad_compare(WIDE_BIG[C], gpu_device)
    assert float((c_h - c_o).abs().max()) < 5e-3 * max(1.0, float(c_o.abs().max()))
    for k, s in stats.items():
        assert s["median"] < 1e-4 and s["p99"] < 2e-3, (k, s)
