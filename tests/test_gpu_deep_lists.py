"""Oracle parity of render_fwd / render_bwd in the regime real scenes and the benchmark run in: tile lists of
several hundred to several thousand entries, i.e. MANY 64-entry batches per wave (recursion state carried across
batches, two-deep id / record prefetch, touch words and dense gradient-row slots of later batches, quads that
saturate in the middle of a list).  Call sites protected: train.py:144 (total_loss.backward()),
gaussian_renderer/__init__.py:97-141.

Three kinds of evidence, all HIP (through the C ABI) against the fp64 oracle:
  * whole-frame gradient comparisons on small frames with very deep lists (every list entry of every tile);
  * the north_star bar stated honestly: <= 1e-4 relative on every Gaussian that blends into no pixel holding a
    decision an fp32 evaluation could take differently (the oracle reports those), statistics for the rest;
  * the 1M-Gaussian 1920x1080 benchmark frame itself: forward state (colour, allmap, final_T / M1 / M2, n_contrib,
    median contributor) and the gradients of a backward restricted to sampled tiles, against the oracle run on exactly
    the Gaussians those tiles list.
PARITY UNPINNED against upstream (oracle/surfel_ref.py header): this pins HIP == oracle."""
import math

import numpy as np
import pytest
import torch

from conftest import oracle_settings, hip_settings, facing_scene
from gaussmart_amd.synthetic import make_scene, activate
from oracle import surfel_ref as O
from oracle_farm import FARM, spec, row_stats, check_against_committed_checksums
from test_gpu_rasterizer import _hip_gradients

pytestmark = pytest.mark.gpu

NAMES = ("means3D", "opacities", "shs", "scales", "rotations")


def _hip_grads(a, cam, dev, flags, wc, wa, bg, deg=3):
    from gaussmart_amd.rasterizer import GaussianRasterizer
    N = a["means3D"].shape[0]
    hin = {k: a[k].clone().to(dev).requires_grad_(True) for k in NAMES}
    m2d = torch.zeros(N, 3, device=dev, requires_grad=True)
    c, r, am = GaussianRasterizer(hip_settings(cam, deg, bg, dev), flags=flags)(
        means3D=hin["means3D"], means2D=m2d, shs=hin["shs"], opacities=hin["opacities"], scales=hin["scales"],
        rotations=hin["rotations"])
    ((c * wc.to(dev)).sum() + (am * wa.to(dev)).sum()).backward()
    torch.cuda.synchronize()
    g = {k: hin[k].grad.cpu().double() for k in NAMES}
    g["means2D"] = m2d.grad.cpu().double()
    return g, c.detach().cpu().double(), am.detach().cpu().double(), r.cpu()


def _oracle_grads(a, cam, flags, wc, wa, bg, deg=3, tiles=None, sel=None, dtype=torch.float64):
    """The oracle (fp64 unless told otherwise) on the Gaussians `sel` (all if None); gradients come back indexed like
    the selection, as float64."""
    S = oracle_settings(cam, deg, dtype, bg)
    oin = {k: (a[k] if sel is None else a[k][sel]).clone().to(dtype).requires_grad_(True) for k in NAMES}
    n = oin["means3D"].shape[0]
    m2d = torch.zeros(n, 3, dtype=dtype, requires_grad=True)
    c, r, am = O.rasterize(oin["means3D"], m2d, oin["opacities"], oin["shs"], None, oin["scales"], oin["rotations"], None,
                           settings=S, flags=flags, tiles=tiles)
    ((c * wc.to(dtype)).sum() + (am * wa.to(dtype)).sum()).backward()
    g = {k: oin[k].grad.double() for k in NAMES}
    g["means2D"] = m2d.grad.double()
    return g, c.detach().double(), am.detach().double(), r, S


def _row_stats(gh, go, N):
    d = (gh - go).abs().reshape(N, -1).amax(1)
    sc = float(go.abs().max())
    rown = go.reshape(N, -1).abs().amax(1)
    rel = d / (rown + 1e-6 * sc)
    act = rown > 1e-4 * sc
    return rel, act, float(d.max()) / max(sc, 1e-30)


def _deep_scene(n, w, h, radius_px, seed, opa_shift=0.0, opa_const=None):
    p, cam = facing_scene(n, w, h, seed=seed, radius_px=radius_px)
    p["opacity"] = p["opacity"] + opa_shift
    a = activate(p)
    if opa_const == "mixed":     # faint haze with a near-opaque splat every 24th: pixels saturate deep inside the list
        a["opacities"] = torch.full_like(a["opacities"], 0.03)
        a["opacities"][::24] = 0.97
    elif opa_const is not None:
        a["opacities"] = torch.full_like(a["opacities"], opa_const)
    return a, cam


# The oracle side of every whole-frame case below runs in a worker process from the start of the session
# (tests/oracle_farm.py); the weights of the differentiated scalar come from seed + 11 everywhere, so a scene that two
# tests look at is computed once.  sens_tols: the margins at which the oracle names the flip-sensitive Gaussians.
def _deep_spec(n, w, h, radius_px, seed, opa_shift, opa_const, flags, want32=False, sens_tols=(1e-3,)):
    return spec("facing", n, w, h, seed, radius_px=radius_px, flags=flags, opa_shift=opa_shift, opa_const=opa_const,
                wseed=seed + 11, want32=want32, sens_tols=sens_tols)


DEEP = [
    # name, n, w, h, radius_px, seed, opacity shift (logit), constant opacity, quirk flags
    ("8k@64x64 r14", 8000, 64, 64, 14.0, 0, -2.0, None, 3),
    ("8k@64x64 r14 exact-derivative", 8000, 64, 64, 14.0, 0, -2.0, None, 0),
    ("8k@64x64 r14 faint (walks far down the list)", 8000, 64, 64, 14.0, 1, -3.5, None, 3),
    ("20k@128x128 r16", 20000, 128, 128, 16.0, 2, -2.5, None, 3),
    ("6k@96x80 r24 haze + near-opaque splats (quads saturate mid-list)", 6000, 96, 80, 24.0, 3, 0.0, "mixed", 3),
    ("12k@70x50 r20 very faint (every pixel reaches the list end region)", 12000, 70, 50, 20.0, 4, 0.0, 0.02, 3),
    # second seeds (round 4)
    ("8k@64x64 r14, second seed", 8000, 64, 64, 14.0, 10, -2.0, None, 3),
    ("6k@96x80 r24 haze + near-opaque splats, second seed", 6000, 96, 80, 24.0, 13, 0.0, "mixed", 3),
]
SHARED_1E4 = {0: "deep64", 3: "deep128"}      # scenes test_gradients_1e4_on_decision_stable_gaussians looks at as well
DEEP_KEYS = []
for i, d in enumerate(DEEP):
    shared = i in SHARED_1E4
    DEEP_KEYS.append(FARM.register("deep-" + d[0].split(" ")[0] + f"-{i}",
                                   _deep_spec(*d[1:], want32=shared, sens_tols=(2e-4, 1e-3) if shared else (1e-3,))))
SHALLOW_1E4 = FARM.register("deep-shallow-2k@256x256", _deep_spec(2000, 256, 256, 6.0, 0, 0.0, None, 3, want32=True,
                                                                   sens_tols=(2e-4, 1e-3)))
SCREEN_FILLING = FARM.register("deep-screen-filling-400", _deep_spec(400, 176, 144, 45.0, 9, 0.0, 0.05, 3, sens_tols=(1e-3,)))
FARM.specs[SCREEN_FILLING]["wseed"] = 21


@pytest.mark.parametrize("case", DEEP_KEYS)
def test_backward_parity_deep_lists(gpu_device, case):
    """Whole-frame gradients (all seven allmap channels carry gradient: depth, alpha, normal, MEDIAN depth and DISTORTION
    terms are live) on frames whose tile lists are thousands of entries long."""
    sp = FARM.specs[case]
    n, flags = sp["n"], sp["flags"]
    gh, c_h, _, _ = _hip_gradients(sp, gpu_device)
    res = FARM.get(case)
    check_against_committed_checksums(case, res)
    go, c_o, L = res["grads"], torch.from_numpy(res["color"]), res["lists"]
    walked = L["walked"]
    batches = math.ceil(walked / 64)
    print(f"\n[{case}] tile lists: mean {L['mean']:.0f}, max {L['max']} entries; deepest entry any pixel blends: "
          f"{walked} (= {batches} batches of 64); mean per-pixel depth {L['mean_depth']:.0f}")
    assert walked >= 256 and batches >= 4
    assert float((c_h - c_o).abs().max()) < 5e-3 and float((c_h - c_o).abs().median()) < 1e-5
    # Gaussians blended into a pixel that holds a decision with a margin below 1e-3 (the oracle names them): a flip of
    # such a decision between fp32 and fp64 -- e.g. WHICH splat is a pixel's median-depth contributor, which receives the
    # whole dL/dmedian of that pixel -- moves their gradients by a finite amount.  Everything else must agree tightly;
    # flip outliers must be few and must all be among the Gaussians the oracle named.
    sens = torch.from_numpy(res["sens"][1e-3][0])
    for k in gh:
        rel, act, normwise = row_stats(gh[k], go[k], n)
        med, p99 = float(rel[act].median()), float(rel[act].quantile(0.99))
        d = (gh[k] - go[k]).abs().reshape(n, -1).amax(1)
        sc = float(go[k].abs().max())
        outl = d > 2e-3 * sc
        print(f"    {k:10s} normwise {normwise:.2e} (decision-stable rows {float(d[~sens].max()) / sc:.2e})  median {med:.2e}  "
              f"p99 {p99:.2e}  ({int(act.sum())} active rows, {int(outl.sum())} flip outliers)")
        assert med < 1e-4 and p99 < 2e-3, (case, k, med, p99)
        assert float(d[~sens].max()) < 1e-4 * sc, (case, k)
        assert int((outl & ~sens).sum()) == 0 and int(outl.sum()) <= max(2, int(act.sum()) // 500), (case, k, int(outl.sum()))


@pytest.mark.parametrize("case", [SHALLOW_1E4, DEEP_KEYS[0], DEEP_KEYS[3]], ids=["shallow", "deep64", "deep128"])
def test_gradients_1e4_on_decision_stable_gaussians(gpu_device, case):
    """north_star's bar -- every gradient within 1e-4 relative -- on the set where it is well defined: Gaussians that
    blend into no pixel whose walk holds a discrete decision (alpha >= 1/255, rho3d <= rho2d, T(1-alpha) < 1e-4,
    T > 0.5, depth >= near, alpha clamp) with relative margin < TOL.  Such a decision can come out differently in fp32
    and fp64 and then moves the gradient of every Gaussian of that pixel by a finite amount -- those Gaussians keep the
    statistical bars of test_backward_parity_*; the excluded fraction is printed.
    On the stable set every gradient row must satisfy |hip - fp64| <= 1e-4 |row|_inf + 1e-5 |tensor|_inf, except rows on
    which the oracle ITSELF, evaluated in fp32, is off by at least a quarter as much (sub-pixel and edge-on splats: the
    conditioning of the ray-splat intersection, which any fp32 implementation -- upstream's included -- shares); those
    rows must stay under 1 % of the set, and the rows outside the bar that the fp32 oracle does NOT share under 0.5 %, each
    below 2e-3 of the tensor's scale."""
    TOL = 2e-4
    sp = FARM.specs[case]
    n, w, h = sp["n"], sp["w"], sp["h"]
    gh, _, radii_h, _ = _hip_gradients(sp, gpu_device)
    res = FARM.get(case)
    go, d32s = res["grads"], res["d32"]
    radii_o, radii_32 = torch.from_numpy(res["radii"]), torch.from_numpy(res["radii32"])
    sens, n_px = res["sens"][TOL]
    # a radius that rounds up differently changes the tile rect, i.e. which tiles may blend the splat at all
    sens = torch.from_numpy(sens) | torch.from_numpy(res["ext_margin_small"]) | (radii_h != radii_o) | (radii_32 != radii_o)
    # the same with five times the margin: where sub-pixel or edge-on splats make rho3d itself uncertain to ~1e-4 in
    # fp32, a decision with a margin slightly above TOL can still flip -- such rows must at least be explained by THIS set
    sens_wide = torch.from_numpy(res["sens"][5 * TOL][0])
    visible = radii_o > 0
    stable = visible & ~sens
    frac_excl = float((visible & sens).sum()) / max(int(visible.sum()), 1)
    print(f"\n[{case}] pixels holding a decision with margin < {TOL:g}: {n_px} of {w * h} ({n_px / (w * h):.2%}); "
          f"Gaussians blending into one of them: {int((visible & sens).sum())} of {int(visible.sum())} visible "
          f"({frac_excl:.1%} excluded from the 1e-4 bar)")
    for k in gh:
        d = (gh[k] - go[k]).abs().reshape(n, -1).amax(1)
        d32 = d32s[k]
        sc = float(go[k].abs().max())
        rown = go[k].reshape(n, -1).abs().amax(1)
        ok = d <= 1e-4 * rown + 1e-5 * sc                  # allclose(rtol = 1e-4, atol = 1e-5 of the tensor's scale)
        cond = ~ok & (d <= 4.0 * d32 + 1e-6 * sc)          # ... or fp32 arithmetic itself cannot do better on that row
        bad = stable & ~ok & ~cond
        n_st = int(stable.sum())
        print(f"    {k:10s} stable rows {n_st}: within rtol 1e-4 / atol 1e-5: {int((stable & ok).sum())}; ill-conditioned (the "
              f"fp32 ORACLE is off by >= a quarter as much): {int((stable & cond).sum())}; flips with margin in [TOL, 5 TOL): "
              f"{int(bad.sum())} | max "
              f"d/scale: HIP {float(d[stable].max()) / sc:.1e}, fp32 oracle {float(d32[stable].max()) / sc:.1e} | excluded rows "
              f"outside the bar: {int((visible & sens & ~ok).sum())} of {int((visible & sens).sum())}")
        # Rows outside the bar that the fp32 oracle does not share: flips with a margin between TOL and 5 TOL (named by
        # `sens_wide`), or rows on which the kernels' evaluation order happens to round worse than the oracle's by more than
        # the factor 4 above (tests/explain_flips.py shows the oracle's own worst rows on these scenes are rounding on
        # surfels seen 65-75 degrees off their normal, not flips).  Both kinds are rare and SMALL: at most 0.5 % of the rows
        # (0.33 % unexplained by sens_wide), none beyond 2e-3 of the tensor's scale.  (Round 3 asserted "at most one, all
        # explained" and passed it by the luck of one weight seed.)
        unexplained = bad & ~sens_wide
        if int(bad.sum()):
            print(f"               rows outside the bar not shared by the fp32 oracle: {int(bad.sum())} ({int(unexplained.sum())} outside "
                  f"sens_wide), largest {float(d[bad].max()) / sc:.1e} of the scale")
            assert float(d[bad].max()) <= 2e-3 * sc, (case, k)
        assert int(bad.sum()) <= max(3, n_st // 200) and int(unexplained.sum()) <= max(2, n_st // 300), (case, k, int(bad.sum()))
        assert int((stable & cond).sum()) <= max(2, n_st // 100), (case, k)
        # and HIP is as accurate as an fp32 evaluation of the oracle's own formulas, in bulk
        assert float(d[stable].median()) <= 3.0 * float(d32[stable].median()) + 1e-7 * sc, (case, k)


def _benchmark_frame(dev):
    N, W, H = 1_000_000, 1920, 1080
    p, cam = make_scene(N, W, H, seed=0)
    return activate(p), cam, N, W, H


def test_benchmark_frame_forward_state_vs_oracle(gpu_device):
    """1M / 1080p: colour, all seven allmap channels, final_T / M1 / M2, n_contrib and the median contributor of 12
    tiles (the longest lists included) against fp64 math on the HIP geometry and the HIP tile lists."""
    from test_gpu_rasterizer import _debug
    a, cam, N, W, H = _benchmark_frame(gpu_device)
    dbg = _debug(a, cam, gpu_device, bg=(0.1, 0.2, 0.3))
    ranges = dbg["ranges"].cpu().numpy().astype(np.int64)
    lens = ranges[:, 1] - ranges[:, 0]
    tiles = sorted(set([int(t) for t in np.argsort(lens)[-4:]] +
                       [int(t) for t in np.linspace(0, ranges.shape[0] - 1, 8).round()]))
    s = dbg["splat"].cpu().double()
    pl = dbg["point_list"].cpu().numpy().astype(np.int64)
    S = oracle_settings(cam, 3, torch.float64, (0.1, 0.2, 0.3))
    out = O.render_tiles(s[:, 0:9].reshape(-1, 3, 3).contiguous(), s[:, 9:11].contiguous(), s[:, 11:14].contiguous(),
                         s[:, 14].contiguous(), s[:, 15:18].contiguous(), torch.from_numpy(pl), ranges, S, margins=True,
                         tiles=tiles)
    gx = (W + 15) // 16
    col, am = dbg["color"].cpu().double(), dbg["allmap"].cpu().double()
    fT, nc = dbg["final_T"].cpu().double(), dbg["n_contrib"].cpu().to(torch.int64)
    med_h = torch.where((nc[1] == 0xFFFFFFFF) | (nc[1] < 0), torch.full_like(nc[1], -1), nc[1])
    walked, n_stable, n_px = 0, 0, 0
    for t in tiles:
        ty, tx = divmod(t, gx)
        ys, xs = slice(ty * 16, min(ty * 16 + 16, H)), slice(tx * 16, min(tx * 16 + 16, W))
        m = {k: v[ys, xs] for k, v in out.margins.items()}
        st = (m["m_alpha"] > 1e-3) & (m["m_term"] > 1e-3) & (m["m_rho"] > 1e-3)
        n_stable += int(st.sum()); n_px += st.numel()
        walked = max(walked, int(out.n_contrib[0][ys, xs].max()))
        assert float((col[:, ys, xs] - out.color[:, ys, xs]).abs()[:, st].max()) < 1e-4
        for c in range(7):
            mk = st & (m["m_med"] > 1e-4) if c == 5 else st
            ref = out.allmap[c][ys, xs]
            tol = (5e-4 if c == 6 else 1e-4) * max(1.0, float(ref.abs().max()))
            assert float((am[c][ys, xs] - ref).abs()[mk].max()) < tol, (t, c)
        assert float((fT[:, ys, xs] - out.final_T[:, ys, xs]).abs()[:, st].max()) < 1e-4
        assert int(((nc[0][ys, xs] != out.n_contrib[0][ys, xs]) & st).sum()) == 0
        assert int(((med_h[ys, xs] != out.n_contrib[1][ys, xs]) & st & (m["m_med"] > 1e-4)).sum()) == 0
    print(f"\n[1M/1080p forward] {len(tiles)} tiles, list lengths {[int(lens[t]) for t in tiles]}, deepest blended entry "
          f"{walked}; decision-stable pixels {n_stable}/{n_px}")
    assert walked >= 256 and n_stable > 0.97 * n_px


def test_benchmark_frame_tile_restricted_backward_vs_oracle(gpu_device):
    """1M / 1080p: the image gradients are zero outside K sampled tiles, gsr_backward runs on the whole frame, and the
    result is compared with the oracle -- preprocess, binning, compositing forward and backward in fp64 -- run on
    exactly the Gaussians those tiles list and restricted to those tiles.  Gaussians in none of the tiles must receive
    exactly zero."""
    from test_gpu_rasterizer import _debug
    dev = gpu_device
    a, cam, N, W, H = _benchmark_frame(dev)
    bg = (0.1, 0.2, 0.3)
    dbg = _debug(a, cam, dev, bg=bg)
    ranges = dbg["ranges"].cpu().numpy().astype(np.int64)
    lens = ranges[:, 1] - ranges[:, 0]
    pl = dbg["point_list"].cpu().numpy().astype(np.int64)
    gx = (W + 15) // 16
    tiles = sorted(set([int(t) for t in np.argsort(lens)[-3:]] +
                       [int(t) for t in np.linspace(gx + 1, ranges.shape[0] - gx - 2, 9).round()]))
    del dbg
    g = torch.Generator().manual_seed(3)
    wc, wa = torch.zeros(3, H, W), torch.zeros(7, H, W)
    for t in tiles:
        ty, tx = divmod(t, gx)
        ys, xs = slice(ty * 16, min(ty * 16 + 16, H)), slice(tx * 16, min(tx * 16 + 16, W))
        wc[:, ys, xs] = torch.randn(3, ys.stop - ys.start, xs.stop - xs.start, generator=g)
        wa[:, ys, xs] = torch.randn(7, ys.stop - ys.start, xs.stop - xs.start, generator=g)
    flags = 3
    gh, c_h, am_h, radii_h = _hip_grads(a, cam, dev, flags, wc, wa, bg)
    sel = np.unique(np.concatenate([pl[ranges[t, 0]:ranges[t, 1]] for t in tiles]))
    sel_t = torch.from_numpy(sel)
    g32, _, _, _, _ = _oracle_grads(a, cam, flags, wc, wa, bg, tiles=tiles, sel=sel_t, dtype=torch.float32)
    go, c_o, am_o, radii_o, S = _oracle_grads(a, cam, flags, wc, wa, bg, tiles=tiles, sel=sel_t)
    L = O.LAST
    # the oracle's own binning of the selection reproduces the lists of the sampled tiles (ids mapped back)
    for t in tiles:
        mine = sel[L["point_list"][int(L["ranges"][t, 0]):int(L["ranges"][t, 1])].numpy()]
        assert np.array_equal(mine, pl[ranges[t, 0]:ranges[t, 1]]), t
    walked = int(L["n_contrib"][0].max())
    print(f"\n[1M/1080p tile-restricted backward] {len(tiles)} tiles, list lengths {[int(lens[t]) for t in tiles]}, "
          f"{sel.size} Gaussians involved, deepest blended entry {walked} (= {math.ceil(walked / 64)} batches)")
    assert walked >= 256
    # forward inside the tiles
    for t in tiles:
        ty, tx = divmod(t, gx)
        ys, xs = slice(ty * 16, min(ty * 16 + 16, H)), slice(tx * 16, min(tx * 16 + 16, W))
        assert float((c_h[:, ys, xs] - c_o[:, ys, xs]).abs().median()) < 1e-5
    outside = torch.ones(N, dtype=torch.bool); outside[sel_t] = False
    n_sel = sel.size
    # Pixel coordinates of ~1e3 and unconstrained orientations (edge-on surfels included) make fp32 itself lose three to
    # four digits here: the oracle evaluated in fp32 is printed beside the kernels, and the kernels may not be worse
    for k in gh:
        assert float(gh[k][outside].abs().max()) == 0.0, k        # nothing leaks out of the masked tiles
        rel, act, normwise = _row_stats(gh[k][sel_t], go[k], n_sel)
        rel32, _, normwise32 = _row_stats(g32[k], go[k], n_sel)
        med, p99 = float(rel[act].median()), float(rel[act].quantile(0.99))
        med32, p99_32 = float(rel32[act].median()), float(rel32[act].quantile(0.99))
        print(f"    {k:10s} normwise {normwise:.2e}  median {med:.2e}  p99 {p99:.2e}  ({int(act.sum())} active rows) | oracle in "
              f"fp32: normwise {normwise32:.2e}  median {med32:.2e}  p99 {p99_32:.2e}")
        assert normwise < 5e-3 and med < 1e-4 and p99 < 5e-3, (k, normwise, med, p99)
        assert med <= 2.0 * med32 + 1e-6 and p99 <= 3.0 * p99_32 + 1e-5, (k, med, med32, p99, p99_32)


@pytest.mark.oracle_cases(SCREEN_FILLING)
def test_backward_parity_screen_filling_splats(gpu_device):
    """Splats that cover a large part of the frame own thousands of gradient rows each (one per 4x4 block they are
    blended into): the row reduction hands every Gaussian with more than 192 rows to its whole workgroup
    -- checked against the fp64 oracle, with faint splats so that every one of them is blended
    far down the lists."""
    sp = FARM.specs[SCREEN_FILLING]
    n = sp["n"]
    gh, c_h, _, _ = _hip_gradients(sp, gpu_device)
    res = FARM.get(SCREEN_FILLING)
    go, c_o = res["grads"], torch.from_numpy(res["color"])
    # rows per Gaussian ~ blocks it is blended into: count the blocks its rect covers as a lower-bound proxy
    rect = res["rect"].astype(np.int64)
    blocks = ((rect[:, 2] - rect[:, 0]) * (rect[:, 3] - rect[:, 1])) * 16
    print(f"\n[screen-filling splats] tile-rect blocks per Gaussian: median {int(np.median(blocks))}, max {int(blocks.max())}; "
          f"{int((blocks > 192).sum())} of {n} Gaussians above the 192-row hand-over")
    assert int((blocks > 1000).sum()) > 50
    assert float((c_h - c_o).abs().max()) < 5e-3
    sens = torch.from_numpy(res["sens"][1e-3][0])
    for k in gh:
        rel, act, normwise = row_stats(gh[k], go[k], n)
        d = (gh[k] - go[k]).abs().reshape(n, -1).amax(1)
        sc = float(go[k].abs().max())
        print(f"    {k:10s} normwise {normwise:.2e}  median {float(rel[act].median()):.2e}  p99 {float(rel[act].quantile(0.99)):.2e}")
        assert float(rel[act].median()) < 1e-4 and float(rel[act].quantile(0.99)) < 2e-3, k
        assert int(((d > 2e-3 * sc) & ~sens).sum()) == 0, k
