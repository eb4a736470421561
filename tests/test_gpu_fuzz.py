"""Differential fuzz: 214 deterministic pseudo-random configurations (72 + 36 + 24 + 30 + 16 + 20 + 16 over the seven tests) -- image sizes that are no multiple of the 16-pixel tile
(down to 1 x 1), 1 ... 600 Gaussians, SH degree 0 ... 3, every combination of the three recalled-behaviour flags, random
background and scale modifier, jittered views, SH or precomputed colours, a share of extreme Gaussians (tests/test_gpu_degenerate.py's kinds) -- each
through the operator and the C ABI against the fp64 oracle, with the fp32 oracle beside it as the yardstick of what single
precision can reach on that scene.  No configuration is tuned: what a seed draws is what runs."""
import math

import pytest
import torch

from conftest import hip_settings, oracle_settings
from oracle_farm import GRAD_NAMES

pytestmark = pytest.mark.gpu

SIZES = [(1, 1), (7, 3), (16, 16), (17, 33), (33, 17), (48, 48), (50, 35), (64, 40), (95, 31), (100, 75), (129, 65), (160, 9),
         (9, 160), (250, 17)]
FLAGS = [0, 1, 2, 3, 512, 513, 514, 515]


def draw(seed):
    g = torch.Generator().manual_seed(7000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    w, h = SIZES[seed % len(SIZES)]
    n = [1, 2, 5, 17, 64, 150, 300, 600][ri(0, 7)]
    deg = ri(0, 3)
    flags = FLAGS[ri(0, len(FLAGS) - 1)]
    bg = tuple(float(x) for x in torch.rand(3, generator=g))
    mod = float(0.5 + 1.5 * torch.rand(1, generator=g))
    radius = float(2.0 + 10.0 * torch.rand(1, generator=g))
    view = ri(0, 3)
    extreme = float(torch.rand(1, generator=g)) < 0.5
    precomp = float(torch.rand(1, generator=g)) < 0.25
    return dict(seed=seed, w=w, h=h, n=n, deg=deg, flags=flags, bg=bg, mod=mod, radius=radius, view=view, extreme=extreme,
                precomp=precomp)


def build(cfg):
    from gaussmart_amd.synthetic import activate, jittered_cameras, make_scene
    p, cam = make_scene(cfg["n"], cfg["w"], cfg["h"], seed=cfg["seed"], radius_px=cfg["radius"])
    a = activate(p)
    if cfg["view"]:
        cam = jittered_cameras(cfg["view"] + 1, cfg["w"], cfg["h"], seed=11, amount=0.25)[cfg["view"]]
    n = cfg["n"]
    if cfg["extreme"] and n >= 5:
        g = torch.Generator().manual_seed(9000 + cfg["seed"])
        idx = torch.randperm(n, generator=g)[: max(1, n // 4)]
        for j, i in enumerate(idx.tolist()):
            kind = j % 5
            if kind == 0:
                a["scales"][i] = torch.exp(-10 - 45 * torch.rand(2, generator=g))
            elif kind == 1:
                a["scales"][i] = torch.exp(2.5 * torch.rand(2, generator=g))
            elif kind == 2:
                a["scales"][i, 0] = math.exp(-20 - 25 * float(torch.rand(1, generator=g)))
            elif kind == 3:
                a["means3D"][i] = a["means3D"][i] / a["means3D"][i, 2] * (0.15 + 0.2 * float(torch.rand(1, generator=g)))
            else:
                a["opacities"][i] = 1e-6 if float(torch.rand(1, generator=g)) < 0.5 else 1.0 - 1e-7
    if cfg["precomp"]:                                  # override_color / convert_SHs_python: colours arrive precomputed
        a["colors_precomp"] = torch.rand(n, 3, generator=torch.Generator().manual_seed(300 + cfg["seed"]))
        del a["shs"]
    g = torch.Generator().manual_seed(100 + cfg["seed"])
    wc, wa = torch.randn(3, cfg["h"], cfg["w"], generator=g), torch.randn(7, cfg["h"], cfg["w"], generator=g)
    return a, cam, wc, wa


def oracle(cfg, a, cam, wc, wa, dtype):
    from oracle import surfel_ref as O
    S = oracle_settings(cam, cfg["deg"], dtype, cfg["bg"], scale_modifier=cfg["mod"])
    names = [k for k in GRAD_NAMES if a.get(k) is not None]
    oin = {k: a[k].clone().to(dtype).requires_grad_(True) for k in names}
    m2d = torch.zeros(cfg["n"], 3, dtype=dtype, requires_grad=True)
    c, r, am = O.rasterize(oin["means3D"], m2d, oin["opacities"], oin.get("shs"), oin.get("colors_precomp"), oin.get("scales"), oin.get("rotations"),
                           None, settings=S, flags=cfg["flags"])
    ((c * wc.to(dtype)).sum() + (am * wa.to(dtype)).sum()).backward()
    g = {k: oin[k].grad.double() for k in names}
    g["means2D"] = m2d.grad.double()
    return g, c.detach().double(), am.detach().double(), r


def hip(cfg, a, cam, wc, wa, dev):
    from gaussmart_amd.rasterizer import GaussianRasterizer
    names = [k for k in GRAD_NAMES if a.get(k) is not None]
    hin = {k: a[k].clone().to(dev).requires_grad_(True) for k in names}
    m2d = torch.zeros(cfg["n"], 3, device=dev, requires_grad=True)
    rast = GaussianRasterizer(hip_settings(cam, cfg["deg"], cfg["bg"], dev, scale_modifier=cfg["mod"]), flags=cfg["flags"])
    c, r, am = rast(means3D=hin["means3D"], means2D=m2d, shs=hin.get("shs"), colors_precomp=hin.get("colors_precomp"),
                    opacities=hin["opacities"], scales=hin["scales"], rotations=hin["rotations"])
    torch.autograd.backward([c, am], [wc.to(dev), wa.to(dev)])
    torch.cuda.synchronize()
    g = {k: hin[k].grad.cpu().double() for k in names}
    g["means2D"] = m2d.grad.cpu().double()
    return g, c.detach().cpu().double(), am.detach().cpu().double(), r.cpu()


@pytest.mark.parametrize("seed", list(range(72)))
def test_random_configuration_against_the_oracle(gpu_device, seed):
    cfg = draw(seed)
    a, cam, wc, wa = build(cfg)
    n = cfg["n"]
    gh, ch, amh, rh = hip(cfg, a, cam, wc, wa, gpu_device)
    g32, c32, am32, r32 = oracle(cfg, a, cam, wc, wa, torch.float32)
    go, co, amo, ro = oracle(cfg, a, cam, wc, wa, torch.float64)
    line = f"seed {seed}: {cfg['w']}x{cfg['h']} n {n} deg {cfg['deg']} flags {cfg['flags']} mod {cfg['mod']:.2f} radius {cfg['radius']:.1f}" \
           f" view {cfg['view']} extreme {cfg['extreme']} precomp {cfg['precomp']} visible {int((ro > 0).sum())}"
    # everything finite, radii as the oracle's up to what fp32 itself gets differently
    assert bool(torch.isfinite(ch).all()) and bool(torch.isfinite(amh).all())
    assert all(bool(torch.isfinite(v).all()) for v in gh.values()) and all(bool(torch.isfinite(v).all()) for v in g32.values())
    bad_r = int((rh != ro).sum())
    assert bad_r <= max(int((r32 != ro).sum()) + 1, n // 100), (line, bad_r)
    # image: per pixel over the 3 + 7 channels, HIP beside the fp32 oracle
    P = cfg["w"] * cfg["h"]
    dh = torch.cat([(ch - co).abs(), (amh - amo).abs()]).amax(0).flatten()
    d32 = torch.cat([(c32 - co).abs(), (am32 - amo).abs()]).amax(0).flatten()
    sc = max(float(torch.cat([co, amo]).abs().max()), 1.0)
    qh = [float(dh.median()), float(dh.quantile(0.99)) if P > 1 else float(dh.max()), float(dh.max())]
    q32 = [float(d32.median()), float(d32.quantile(0.99)) if P > 1 else float(d32.max()), float(d32.max())]
    line += f" | image err/scale HIP med {qh[0] / sc:.1e} p99 {qh[1] / sc:.1e} max {qh[2] / sc:.1e}; fp32 oracle {q32[0] / sc:.1e} {q32[1] / sc:.1e} {q32[2] / sc:.1e}"
    assert qh[0] <= max(4 * q32[0], 2e-6 * sc), line
    assert qh[1] <= max(4 * q32[1], 1e-4 * sc), line
    # the maximum belongs to the pixels where a discrete decision of the walk falls the other way (T > 0.5 picks another
    # splat's depth as the median, alpha >= 1/255, ...): few, counted, and named by channel in the report line
    big = dh > max(4 * q32[2], 2e-3 * sc)
    if bool(big.any()):
        per_ch = torch.cat([(ch - co).abs(), (amh - amo).abs()]).flatten(1)[:, big].amax(1)
        line += f" | {int(big.sum())} pixel(s) beyond the maximum bar, per channel (rgb, depth, alpha, normal xyz, median, dist): " \
                + " ".join(f"{float(v):.1e}" for v in per_ch)
    assert int(big.sum()) <= max(1, P // 2000), line
    # gradients: per-row error relative to the tensor's scale; median and p90 beside the fp32 oracle
    for k in gh:
        ref = go[k].reshape(n, -1)
        if ref.numel() == 0:            # (features_rest [N,0,3]: only the dc coefficient is stored)
            assert gh[k].numel() == 0
            continue
        tsc = float(ref.abs().max())
        if tsc == 0.0:
            assert float(gh[k].abs().max()) == 0.0, (line, k)
            continue
        eh = (gh[k].reshape(n, -1) - ref).abs().amax(1) / tsc
        e32 = (g32[k].reshape(n, -1) - ref).abs().amax(1) / tsc
        mh, m32 = float(eh.median()), float(e32.median())
        ph, p32 = (float(eh.quantile(0.9)), float(e32.quantile(0.9))) if n > 1 else (float(eh.max()), float(e32.max()))
        line += f" | {k} med {mh:.1e}/{m32:.1e} p90 {ph:.1e}/{p32:.1e} max {float(eh.max()):.1e}/{float(e32.max()):.1e}"
        assert mh <= max(4 * m32, 1e-5 if n >= 16 else 1e-4), (line, k)      # (a median over a handful of rows is one row)
        assert ph <= max(4 * p32, 2e-4), (line, k)
        assert float(eh.max()) <= max(4 * float(e32.max()), 2e-2), (line, k)
    print("\n   " + line)


# ---- the raw-parameter operator (what the fused trainer calls): logits / log-scales / un-normalised quaternions / split SH in,
# ---- activations inside K1 / K8, and its two lean variants ------------------------------------------------------------------
RAW_NAMES = ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation")


def build_raw(cfg):
    from gaussmart_amd.synthetic import jittered_cameras, make_scene
    p, cam = make_scene(cfg["n"], cfg["w"], cfg["h"], seed=cfg["seed"], radius_px=cfg["radius"])
    if cfg["view"]:
        cam = jittered_cameras(cfg["view"] + 1, cfg["w"], cfg["h"], seed=11, amount=0.25)[cfg["view"]]
    n = cfg["n"]
    g = torch.Generator().manual_seed(9500 + cfg["seed"])
    p["rotation"] = p["rotation"] * torch.exp(2.0 * torch.randn(n, 1, generator=g))          # norms from 0.02 to 50
    if cfg["extreme"] and n >= 5:
        idx = torch.randperm(n, generator=g)[: max(1, n // 4)]
        for j, i in enumerate(idx.tolist()):
            kind = j % 4
            if kind == 0:
                p["scaling"][i] = -10 - 45 * torch.rand(2, generator=g)
            elif kind == 1:
                p["scaling"][i] = 2.5 * torch.rand(2, generator=g)
            elif kind == 2:
                p["opacity"][i] = 16.0 if float(torch.rand(1, generator=g)) < 0.5 else -14.0
            else:
                p["rotation"][i] = p["rotation"][i] * 1e-12                                  # a quaternion of norm ~1e-12
    K = (cfg["deg_store"] + 1) ** 2
    p["features_rest"] = p["features_rest"][:, : K - 1].contiguous()
    g = torch.Generator().manual_seed(100 + cfg["seed"])
    wc, wa = torch.randn(3, cfg["h"], cfg["w"], generator=g), torch.randn(7, cfg["h"], cfg["w"], generator=g)
    return p, cam, wc, wa


def oracle_raw(cfg, p, cam, wc, wa, dtype):
    from oracle import surfel_ref as O
    S = oracle_settings(cam, cfg["deg"], dtype, cfg["bg"], scale_modifier=cfg["mod"])
    oin = {k: p[k].clone().to(dtype).requires_grad_(True) for k in RAW_NAMES}
    m2d = torch.zeros(cfg["n"], 3, dtype=dtype, requires_grad=True)
    # scene/gaussian_model.py:103-123: exp, normalize, sigmoid, cat(dc, rest)
    c, r, am = O.rasterize(oin["xyz"], m2d, torch.sigmoid(oin["opacity"]), torch.cat((oin["features_dc"], oin["features_rest"]), 1),
                           None, torch.exp(oin["scaling"]), torch.nn.functional.normalize(oin["rotation"]), None, settings=S,
                           flags=cfg["flags"])
    ((c * wc.to(dtype)).sum() + (am * wa.to(dtype)).sum()).backward()
    g = {k: oin[k].grad.double() for k in RAW_NAMES}
    g["means2D"] = m2d.grad.double()
    return g, c.detach().double(), am.detach().double(), r


def hip_raw(cfg, p, cam, wc, wa, dev, variant):
    from gaussmart_amd.rasterizer import rasterize_gaussians_raw
    hin = {k: p[k].clone().to(dev).requires_grad_(True) for k in RAW_NAMES}
    m2d = torch.zeros(cfg["n"], 3, device=dev, requires_grad=True)
    rs = hip_settings(cam, cfg["deg"], cfg["bg"], dev, scale_modifier=cfg["mod"])
    c, r, am = rasterize_gaussians_raw(hin["xyz"], m2d, hin["features_dc"], hin["features_rest"], hin["opacity"], hin["scaling"],
                                       hin["rotation"], rs, flags=cfg["flags"], color_only=variant == "color_only",
                                       no_dist_median=variant == "no_dist_median")
    if am is None:
        c.backward(wc.to(dev))
    else:
        torch.autograd.backward([c, am], [wc.to(dev), wa.to(dev)])
    torch.cuda.synchronize()
    g = {k: hin[k].grad.cpu().double() for k in RAW_NAMES}
    g["means2D"] = m2d.grad.cpu().double()
    return g, c.detach().cpu().double(), None if am is None else am.detach().cpu().double(), r.cpu()


@pytest.mark.parametrize("seed", list(range(100, 136)))
def test_random_configuration_raw_parameter_operator(gpu_device, seed):
    cfg = draw(seed)
    cfg["deg_store"] = max(cfg["deg"], (seed // 3) % 4)                      # stored coefficients >= active degree
    variant = ("all", "no_dist_median", "color_only")[seed % 3]
    p, cam, wc, wa = build_raw(cfg)
    if variant == "no_dist_median":
        wa[5:] = 0                      # the caller's promise: it consumes neither channel
    elif variant == "color_only":
        wa[:] = 0
    n = cfg["n"]
    gh, ch, amh, rh = hip_raw(cfg, p, cam, wc, wa, gpu_device, variant)
    g32, c32, am32, r32 = oracle_raw(cfg, p, cam, wc, wa, torch.float32)
    go, co, amo, ro = oracle_raw(cfg, p, cam, wc, wa, torch.float64)
    line = f"seed {seed} [{variant}]: {cfg['w']}x{cfg['h']} n {n} deg {cfg['deg']}/{cfg['deg_store']} flags {cfg['flags']} mod {cfg['mod']:.2f}" \
           f" view {cfg['view']} extreme {cfg['extreme']} visible {int((ro > 0).sum())}"
    assert bool(torch.isfinite(ch).all()) and all(bool(torch.isfinite(v).all()) for v in gh.values()), line
    assert all(bool(torch.isfinite(v).all()) for v in g32.values()), ("fp32 oracle", line)
    assert int((rh != ro).sum()) <= max(int((r32 != ro).sum()) + 1, n // 100), line
    if variant == "color_only":
        assert amh is None
        img_h, img_o, img_32 = ch, co, c32
    elif variant == "no_dist_median":
        assert float(amh[5:].abs().max()) == 0.0
        img_h, img_o, img_32 = torch.cat([ch, amh[:5]]), torch.cat([co, amo[:5]]), torch.cat([c32, am32[:5]])
    else:
        img_h, img_o, img_32 = torch.cat([ch, amh]), torch.cat([co, amo]), torch.cat([c32, am32])
    P = cfg["w"] * cfg["h"]
    sc = max(float(img_o.abs().max()), 1.0)
    dh, d32 = (img_h - img_o).abs().amax(0).flatten(), (img_32 - img_o).abs().amax(0).flatten()
    assert float(dh.median()) <= max(4 * float(d32.median()), 2e-6 * sc), line
    big = dh > max(4 * float(d32.max()), 2e-3 * sc)
    assert int(big.sum()) <= max(1, P // 2000), (line, int(big.sum()))
    for k in gh:
        ref = go[k].reshape(n, -1)
        if ref.numel() == 0:            # (features_rest [N,0,3]: only the dc coefficient is stored)
            assert gh[k].numel() == 0
            continue
        tsc = float(ref.abs().max())
        if tsc == 0.0:
            assert float(gh[k].abs().max()) == 0.0, (line, k)
            continue
        eh = (gh[k].reshape(n, -1) - ref).abs().amax(1) / tsc
        e32 = (g32[k].reshape(n, -1) - ref).abs().amax(1) / tsc
        mh, m32 = float(eh.median()), float(e32.median())
        ph, p32 = (float(eh.quantile(0.9)), float(e32.quantile(0.9))) if n > 1 else (float(eh.max()), float(e32.max()))
        line += f" | {k} med {mh:.1e}/{m32:.1e} p90 {ph:.1e}/{p32:.1e} max {float(eh.max()):.1e}/{float(e32.max()):.1e}"
        assert mh <= max(4 * m32, 1e-5 if n >= 16 else 1e-4), (line, k)      # (a median over a handful of rows is one row)
        assert ph <= max(4 * p32, 2e-4), (line, k)
        assert float(eh.max()) <= max(4 * float(e32.max()), 2e-2), (line, k)
    print("\n   " + line)


# ---- wide payloads (colors_precomp [N,C], C = 4 ... 64: the matrix-pipe kernels) and a precomputed T (cov3D_precomp) ---------
@pytest.mark.parametrize("seed", list(range(200, 224)))
def test_random_configuration_wide_payload_and_precomputed_transmat(gpu_device, seed):
    from gaussmart_amd.rasterizer import GaussianRasterizer
    from oracle import surfel_ref as O
    dev = gpu_device
    cfg = draw(seed)
    cfg["precomp"] = False
    a, cam, _, wa = build(cfg)
    n = cfg["n"]
    g = torch.Generator().manual_seed(400 + seed)
    C = [4, 8, 12, 16, 20, 32, 48, 64, 3, 3][seed % 10]                       # (3: the RGB kernels, with a precomputed T)
    a.pop("shs")
    a["colors_precomp"] = torch.randn(n, C, generator=g)
    bg = tuple(float(x) for x in torch.rand(C, generator=g))
    use_T = seed % 3 == 0
    if use_T:       # T as K1 would build it, in fp64 and rounded once (the reference's compute_cov3D_python alternative)
        S64 = oracle_settings(cam, 0, torch.float64, bg, scale_modifier=cfg["mod"])
        geom = O.preprocess(a["means3D"].double(), a["scales"].double(), a["rotations"].double(), a["opacities"].double(), None,
                            a["colors_precomp"].double(), None, S64)
        T = torch.tensor([1., 0, 0, 0, 1, 0, 0, 0, 1]).repeat(n, 1)
        T[geom.vis_idx] = geom.Tm.reshape(-1, 9).float()
        a["cov3D_precomp"] = T
        a.pop("scales"); a.pop("rotations")
    wc = torch.randn(C, cfg["h"], cfg["w"], generator=g)
    names = [k for k in GRAD_NAMES if a.get(k) is not None]

    def run_oracle(dtype):
        S = oracle_settings(cam, 0, dtype, bg, scale_modifier=cfg["mod"])
        oin = {k: a[k].clone().to(dtype).requires_grad_(True) for k in names}
        m2d = torch.zeros(n, 3, dtype=dtype, requires_grad=True)
        c, r, am = O.rasterize(oin["means3D"], m2d, oin["opacities"], None, oin["colors_precomp"], oin.get("scales"),
                               oin.get("rotations"), oin.get("cov3D_precomp"), settings=S, flags=cfg["flags"])
        ((c * wc.to(dtype)).sum() + (am * wa.to(dtype)).sum()).backward()
        gr = {k: oin[k].grad.double() for k in names}
        gr["means2D"] = m2d.grad.double()
        return gr, torch.cat([c.detach().double(), am.detach().double()]), r

    hin = {k: a[k].clone().to(dev).requires_grad_(True) for k in names}
    m2d = torch.zeros(n, 3, device=dev, requires_grad=True)
    rast = GaussianRasterizer(hip_settings(cam, 0, bg, dev, scale_modifier=cfg["mod"]), flags=cfg["flags"])
    c, r, am = rast(means3D=hin["means3D"], means2D=m2d, colors_precomp=hin["colors_precomp"], opacities=hin["opacities"],
                    scales=hin.get("scales"), rotations=hin.get("rotations"), cov3D_precomp=hin.get("cov3D_precomp"))
    torch.autograd.backward([c, am], [wc.to(dev), wa.to(dev)])
    torch.cuda.synchronize()
    gh = {k: hin[k].grad.cpu().double() for k in names}
    gh["means2D"] = m2d.grad.cpu().double()
    img_h, rh = torch.cat([c.detach(), am.detach()]).cpu().double(), r.cpu()
    g32, img_32, r32 = run_oracle(torch.float32)
    go, img_o, ro = run_oracle(torch.float64)
    line = f"seed {seed}: C {C} precomputed T {use_T} {cfg['w']}x{cfg['h']} n {n} flags {cfg['flags']} mod {cfg['mod']:.2f} view {cfg['view']}" \
           f" extreme {cfg['extreme']} visible {int((ro > 0).sum())}"
    assert bool(torch.isfinite(img_h).all()) and all(bool(torch.isfinite(v).all()) for v in gh.values()), line
    assert int((rh != ro).sum()) <= max(int((r32 != ro).sum()) + 1, n // 100), line
    P = cfg["w"] * cfg["h"]
    sc = max(float(img_o.abs().max()), 1.0)
    dh, d32 = (img_h - img_o).abs().amax(0).flatten(), (img_32 - img_o).abs().amax(0).flatten()
    assert float(dh.median()) <= max(4 * float(d32.median()), 2e-6 * sc), line
    assert int((dh > max(4 * float(d32.max()), 2e-3 * sc)).sum()) <= max(1, P // 2000), line
    for k in gh:
        ref = go[k].reshape(n, -1)
        tsc = float(ref.abs().max())
        if tsc == 0.0:
            assert float(gh[k].abs().max()) == 0.0, (line, k)
            continue
        eh = (gh[k].reshape(n, -1) - ref).abs().amax(1) / tsc
        e32 = (g32[k].reshape(n, -1) - ref).abs().amax(1) / tsc
        mh, m32 = float(eh.median()), float(e32.median())
        ph, p32 = (float(eh.quantile(0.9)), float(e32.quantile(0.9))) if n > 1 else (float(eh.max()), float(e32.max()))
        line += f" | {k} med {mh:.1e}/{m32:.1e} p90 {ph:.1e}/{p32:.1e} max {float(eh.max()):.1e}/{float(e32.max()):.1e}"
        assert mh <= max(4 * m32, 1e-5 if n >= 16 else 1e-4), (line, k)
        assert ph <= max(4 * p32, 2e-4), (line, k)
        assert float(eh.max()) <= max(4 * float(e32.max()), 2e-2), (line, k)
    print("\n   " + line)


# ---- the fused objective (L1 + SSIM + normal consistency + distortion) on drawn image sizes, weights and allmaps with holes ----
@pytest.mark.parametrize("seed", list(range(300, 330)))
def test_random_objective_against_the_oracles(gpu_device, seed):
    from gaussmart_amd.fused_objective import training_objective
    from gaussmart_amd.synthetic import jittered_cameras
    from oracle import loss_ref, regularizer_ref as R
    dev = gpu_device
    g = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    W, H = [(1, 1), (2, 5), (11, 11), (16, 16), (31, 47), (64, 33), (97, 5), (130, 71), (40, 150), (33, 32)][seed % 10]
    lam = [0.0, 0.2, 0.2, 0.7, 1.0][ri(0, 4)]
    ln = [0.0, 0.05, 0.05, 1.0][ri(0, 3)]
    ld = [0.0, 0.0, 100.0, 1000.0][ri(0, 3)]
    ratio = [0.0, 0.0, 0.5, 1.0][ri(0, 3)]
    cam = jittered_cameras(3, W, H, seed=seed, device=dev, amount=0.3)[seed % 3]
    img = torch.rand(3, H, W, generator=g)
    gt = (img + 0.15 * torch.randn(3, H, W, generator=g)).clamp(0, 1)
    alpha = torch.rand(1, H, W, generator=g)
    alpha = torch.where(torch.rand(1, H, W, generator=g) < 0.3, torch.zeros_like(alpha), alpha)       # holes: alpha == 0
    alpha = torch.where(torch.rand(1, H, W, generator=g) < 0.05, torch.full_like(alpha, 1e-7), alpha)  # nearly empty pixels
    depth = 0.5 + 19.5 * torch.rand(1, H, W, generator=g)
    nrm = torch.nn.functional.normalize(torch.randn(3, H, W, generator=g), dim=0) * alpha
    am = torch.cat([alpha * depth, alpha, nrm, torch.where(alpha > 0, depth * (0.8 + 0.4 * torch.rand(1, H, W, generator=g)),
                                                           torch.zeros_like(depth)), alpha * torch.rand(1, H, W, generator=g)])
    outs = []
    for defer in (False, True):
        ih, ah = img.to(dev).requires_grad_(True), am.to(dev).requires_grad_(True)
        total, parts = training_objective(ih, ah, gt.to(dev), cam, lam, ln, ld, ratio, defer_value=defer)
        total.backward()
        torch.cuda.synchronize()
        outs.append((float(total.detach()), parts.detach().cpu().double(), ih.grad.cpu().double(),
                     None if ah.grad is None else ah.grad.cpu().double()))
    # the deferred form writes the same five scalars and the same gradients
    assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    assert (outs[0][3] is None) == (outs[1][3] is None) and (outs[0][3] is None or torch.equal(outs[0][3], outs[1][3]))
    tot, parts, gi, ga = outs[0]
    io, ao = img.double().requires_grad_(True), am.double().requires_grad_(True)
    lo, l1o, so = loss_ref.photometric_loss(io, gt.double(), lam)
    ro, nmo, dmo = R.regularizer_loss(ao, cam.world_view_transform.cpu().double(), cam.full_proj_transform.cpu().double(), ratio, ln, ld)
    (lo + ro).backward()
    use_reg = ln > 0 or ld > 0
    want = float(lo + ro) if use_reg else float(lo)
    line = f"seed {seed}: {W}x{H} lambda_dssim {lam} lambda_normal {ln} lambda_dist {ld} depth_ratio {ratio}: total {tot:.6f} vs {want:.6f}"
    assert math.isfinite(tot) and bool(torch.isfinite(gi).all()) and (ga is None or bool(torch.isfinite(ga).all())), line
    assert abs(tot - want) <= 3e-5 * max(abs(want), 1e-3), line
    assert abs(float(parts[0]) - float(l1o)) <= 3e-5 * max(float(l1o), 1e-3) and abs(float(parts[1]) - float(so)) <= 3e-5, line
    gio = io.grad
    assert float((gi - gio).abs().max()) <= 3e-4 * float(gio.abs().max()) + 1e-12, line
    if use_reg:
        # Compared where alpha > 0.  At a hole the reference's formulation leaves 0 / 0 on the depth and alpha channels and,
        # where both x- or both y-neighbours of a pixel are holes (their points coincide at the camera centre, the cross product
        # is exactly 0, F.normalize's backward is g / 1e-12), numbers of order 1e9 on the median channel -- on HOLE pixels only
        # (d cross / d dy = dx x . = 0), which the rasterizer's backward never reads.  The fused kernel writes zeros there.
        lit = (am[1] > 0)
        gao = torch.nan_to_num(ao.grad, 0.0, 0.0, 0.0)
        for c in range(7):
            if not bool(lit.any()):
                break
            sc = float(gao[c][lit].abs().max())
            err = float((ga[c] - gao[c])[lit].abs().max())
            assert err <= 3e-3 * sc + 1e-10, (line, c, err, sc)
    else:
        assert ga is None or float(ga.abs().max()) == 0.0
    print("\n   " + line)


# ---- the optimiser step: FusedAdam against torch.optim.Adam over drawn shapes, step counts, learning rates and gradient scales ----
@pytest.mark.parametrize("seed", list(range(400, 416)))
def test_random_adam_against_torch(gpu_device, seed):
    from gaussmart_amd.fused_adam import FusedAdam
    dev = gpu_device
    g = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    n = [1, 2, 63, 64, 65, 257, 1000, 4097, 20011][ri(0, 8)]
    shapes = [(n, 3), (n, 1, 3), (n, [0, 3, 8, 15][ri(0, 3)], 3), (n, 1), (n, 2), (n, 4)]
    shapes = [s for s in shapes if all(d > 0 for d in s)][: ri(1, 6)]
    lrs = [float(10.0 ** (-6 + 5 * torch.rand(1, generator=g))) if ri(0, 5) else 0.0 for _ in shapes]
    gexp = [float(-30 + 50 * torch.rand(1, generator=g)) for _ in shapes]            # gradient scale 1e-30 ... 1e20 per tensor
    steps = ri(1, 40)
    init = [torch.randn(s, generator=g) * float(10.0 ** (-3 + 6 * torch.rand(1, generator=g))) for s in shapes]
    pa = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    pb = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    oa = FusedAdam([{"params": [p], "lr": lr, "name": str(i)} for i, (p, lr) in enumerate(zip(pa, lrs))], lr=0.0, eps=1e-15)
    ob = torch.optim.Adam([{"params": [p], "lr": lr, "name": str(i)} for i, (p, lr) in enumerate(zip(pb, lrs))], lr=0.0, eps=1e-15,
                          foreach=False, fused=False)
    for it in range(steps):
        for i, (a, b) in enumerate(zip(pa, pb)):
            gr = torch.randn(a.shape, generator=g) * 10.0 ** gexp[i]
            mode = ri(0, 5)
            if mode == 0:
                gr[::2] = 0                                   # invisible rows: zero gradient, the moments still decay
            elif mode == 1:
                gr.zero_()
            a.grad, b.grad = gr.to(dev), gr.to(dev)
            if mode == 2 and i == 0:                          # a parameter that sits this step out (replaced by a densification)
                a.grad = b.grad = None
        if ri(0, 3) == 0:
            k = ri(0, len(shapes) - 1)
            oa.param_groups[k]["lr"] = ob.param_groups[k]["lr"] = lrs[k] * float(torch.rand(1, generator=g))
        oa.step(); ob.step()
    for i, (a, b) in enumerate(zip(pa, pb)):
        assert bool(torch.isfinite(a).all()) == bool(torch.isfinite(b).all())
        tag = (seed, i, tuple(a.shape), lrs[i], gexp[i], steps)
        torch.testing.assert_close(a, b, rtol=3e-6, atol=1e-7 * float(b.detach().abs().max()) + 1e-30, msg=lambda m: f"{tag}: {m}")
        if a in oa.state and b in ob.state and "exp_avg" in ob.state[b]:
            ea, eb = oa.state[a]["exp_avg"], ob.state[b]["exp_avg"]
            torch.testing.assert_close(ea, eb, rtol=3e-6, atol=2e-6 * float(eb.abs().max()) + 1e-37, msg=lambda m: f"{tag} exp_avg: {m}")
            va, vb = oa.state[a]["exp_avg_sq"], ob.state[b]["exp_avg_sq"]
            torch.testing.assert_close(va, vb, rtol=3e-6, atol=2e-6 * float(vb.abs().max()) + 1e-37, msg=lambda m: f"{tag} exp_avg_sq: {m}")
            assert float(oa.state[a]["step"]) == float(ob.state[b]["step"])


# ---- binning: tile lists, ranges and the reconstructed 64-bit keys bit for bit against NumPy's stable sort, on adversarial draws ----
@pytest.mark.parametrize("seed", list(range(500, 520)))
def test_random_binning_bit_exact(gpu_device, seed):
    import numpy as np
    from gaussmart_amd.synthetic import activate, make_scene
    from test_gpu_rasterizer import _debug, _numpy_binning
    g = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    w, h = SIZES[ri(0, len(SIZES) - 1)] if seed % 2 else [(640, 360), (333, 211), (1000, 64), (64, 1000), (512, 512)][ri(0, 4)]
    n = [1, 7, 300, 4097, 20000, 60001][ri(0, 5)]
    p, cam = make_scene(n, w, h, seed=seed, radius_px=float(2 + 12 * torch.rand(1, generator=g)))
    a = activate(p)
    kind = seed % 5
    if kind == 0:                                   # every depth equal: both sort stages must be stable
        a["means3D"][:, 2] = 5.0
    elif kind == 1:                                 # a handful of depth values
        a["means3D"][:, 2] = torch.round(a["means3D"][:, 2])
    elif kind == 2:                                 # everything in one spot: one tile holds (nearly) every instance
        a["means3D"][:, :2] = a["means3D"][:1, :2] + 1e-3 * torch.randn(n, 2, generator=g)
    elif kind == 3 and n >= 7:                      # a few screen-filling splats among the others
        a["scales"][: max(1, n // 50)] = 30.0
    dbg = _debug(a, cam, gpu_device)
    keys, plist, ranges, _ = _numpy_binning(dbg, w, h)
    assert dbg["num_rendered"] == keys.size, (seed, dbg["num_rendered"], keys.size)
    np.testing.assert_array_equal(dbg["point_list"].cpu().numpy().astype(np.uint32)[: keys.size], plist)
    np.testing.assert_array_equal(dbg["ranges"].cpu().numpy().astype(np.uint32), ranges)
    if keys.size:
        pl = dbg["point_list"].cpu().numpy().astype(np.int64)[: keys.size]
        tile_of = np.repeat(np.arange(ranges.shape[0]), (ranges[:, 1] - ranges[:, 0]).astype(np.int64))
        dk = dbg["depth_key"].cpu().numpy().astype(np.uint32).astype(np.uint64)
        np.testing.assert_array_equal((tile_of.astype(np.uint64) << np.uint64(32)) | dk[pl], keys)
        rows = dbg["inst_row"].cpu().numpy().astype(np.int64)[: keys.size]
        assert np.array_equal(np.sort(rows), np.arange(keys.size))


# ---- simple_knn.distCUDA2: mean squared distance to the three nearest neighbours against a k-d tree, on drawn point clouds ----
@pytest.mark.parametrize("seed", list(range(600, 616)))
def test_random_point_cloud_knn(gpu_device, seed):
    import numpy as np
    from scipy.spatial import cKDTree
    from simple_knn._C import distCUDA2
    rng = np.random.default_rng(seed)
    n = int([4, 5, 63, 64, 65, 1000, 1025, 30000, 100003][rng.integers(0, 9)])
    scale = float(10.0 ** rng.uniform(-4, 4))
    kind = seed % 6
    if kind == 0:
        pts = rng.normal(size=(n, 3))
    elif kind == 1:       # a line (Morton boxes degenerate to one axis)
        pts = np.c_[rng.uniform(-1, 1, size=n), np.zeros(n), np.zeros(n)]
    elif kind == 2:       # many exact duplicates
        pts = rng.integers(0, max(2, n // 8), size=(n, 3)).astype(np.float64)
    elif kind == 3:       # two far-apart clusters of very different density
        pts = np.r_[rng.normal(size=(n // 2, 3)) * 1e-3, rng.normal(size=(n - n // 2, 3)) + 50.0]
    elif kind == 4:       # a plane plus outliers
        pts = np.c_[rng.uniform(-1, 1, size=(n, 2)), np.zeros(n)]
        pts[:: max(1, n // 7)] += rng.normal(size=pts[:: max(1, n // 7)].shape) * 20
    else:                 # a thin shell
        v = rng.normal(size=(n, 3)); pts = v / np.linalg.norm(v, axis=1, keepdims=True)
    pts = (pts * scale).astype(np.float32)
    out = distCUDA2(torch.from_numpy(pts).to(gpu_device)).cpu().numpy().astype(np.float64)
    p64 = pts.astype(np.float64)
    d, _ = cKDTree(p64).query(p64, k=4)
    ref = (d[:, 1:] ** 2).mean(1)
    # fp32 differences of coordinates ~scale: absolute floor from the rounding of (a - b)^2 at that magnitude
    floor = 3 * (float(np.abs(pts).max()) * 2.0 ** -23) ** 2 * 8
    assert np.all(np.isfinite(out))
    np.testing.assert_allclose(out, ref, rtol=5e-5, atol=floor, err_msg=f"seed {seed} n {n} kind {kind} scale {scale:.2e}")
