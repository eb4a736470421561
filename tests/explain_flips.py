"""Which (pixel, splat) decision does an fp32 evaluation take differently?  CPU-only diagnostic behind the gradient bars.

The tail of every fp32-vs-fp64 gradient comparison (tests/test_gpu_rasterizer.py, tests/test_gpu_deep_lists.py) is made of
discrete decisions of the compositing walk -- alpha >= 1/255, rho3d <= rho2d, T (1 - alpha) < 1e-4, T > 0.5 (median depth),
depth >= near, the 0.99 clamp -- that land on the other side of their threshold in fp32.  This script evaluates the ORACLE in
fp64 and in fp32 on a registered oracle-farm case, finds the Gaussian whose gradient row differs most for a tensor, and lists
the pairs of its tiles on which the two evaluations decide differently, with the fp64 margin of each.  HIP computes in fp32
too: its outliers are events of the same kind (round 3's `k1_tests.log`: a K1 variant with hoisted loads moved `scales`
normwise from 0.96e-3 to 1.045e-3 on this scene, across a fixed 1e-3 bar -- the fp32 oracle itself sits at 0.64e-3 there).

    python tests/explain_flips.py facing2k-s0-flags3 scales        (test infrastructure: imports oracle/)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

import oracle_farm as F                                    # noqa: E402
from oracle import surfel_ref as O                         # noqa: E402


def decisions(px, py, Tm, xy, opa):
    """The discrete decisions of one tile's walk, per (list entry, pixel), and the quantities they threshold."""
    Tu, Tv, Tw = Tm[:, 0, :], Tm[:, 1, :], Tm[:, 2, :]
    pxb, pyb = px[None, :], py[None, :]
    k = [pxb * Tw[:, i:i + 1] - Tu[:, i:i + 1] for i in range(3)]
    l = [pyb * Tw[:, i:i + 1] - Tv[:, i:i + 1] for i in range(3)]
    p0, p1, p2 = k[1] * l[2] - k[2] * l[1], k[2] * l[0] - k[0] * l[2], k[0] * l[1] - k[1] * l[0]
    ok = p2 != 0
    p2s = torch.where(ok, p2, torch.ones_like(p2))
    sx, sy = p0 / p2s, p1 / p2s
    rho3d = sx * sx + sy * sy
    dx, dy = xy[:, 0:1] - pxb, xy[:, 1:2] - pyb
    rho2d = O.FILTER_INV_SQUARE * (dx * dx + dy * dy)
    use3d = rho3d <= rho2d
    rho = torch.where(use3d, rho3d, rho2d)
    depth = torch.where(use3d, sx * Tw[:, 0:1] + sy * Tw[:, 1:2] + Tw[:, 2:3], Tw[:, 2:3].expand_as(sx))
    ok = ok & (depth >= O.NEAR_N)
    a_raw = opa[:, None] * torch.exp(-0.5 * rho)
    alpha = torch.clamp_max(a_raw, O.ALPHA_MAX)
    blends = ok & (alpha >= O.ALPHA_MIN)
    a_eff = torch.where(blends, alpha, torch.zeros_like(alpha))
    cum = torch.cumprod(1 - a_eff, dim=0)
    term = blends & (cum < O.T_EPS)
    L, P = alpha.shape
    first = torch.where(term.any(0), term.to(torch.uint8).argmax(0), torch.full((P,), L))
    contrib = blends & (torch.arange(L)[:, None] < first[None, :])
    a_c = torch.where(contrib, alpha, torch.zeros_like(alpha))
    T_i = torch.cat([torch.ones(1, P, dtype=alpha.dtype), torch.cumprod(1 - a_c, dim=0)[:-1]], 0)
    med = contrib & (T_i > 0.5)
    return dict(blends=blends, use3d=use3d & contrib, contrib=contrib, median=med, clamped=contrib & (a_raw > O.ALPHA_MAX),
                alpha=alpha, a_raw=a_raw, cum=cum, T_i=T_i, rho3d=rho3d, rho2d=rho2d)


def main(key, tensor):
    import test_gpu_rasterizer, test_gpu_deep_lists, test_gpu_wide_payload      # noqa: F401  (register the cases)
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    sp = dict(F.FARM.specs[key])
    a, cam, bg, wc, wa = F.build_inputs(sp)
    n = a["means3D"].shape[0]
    g32, _, _, _, _ = F._oracle_once(sp, a, cam, bg, wc, wa, torch.float32)
    L32 = dict(O.LAST)
    g64, _, _, _, _ = F._oracle_once(sp, a, cam, bg, wc, wa, torch.float64)
    L64 = dict(O.LAST)
    d = (g32[tensor] - g64[tensor]).abs().reshape(n, -1).amax(1)
    sc = float(g64[tensor].abs().max())
    worst = torch.argsort(d, descending=True)[:3]
    print(f"{key}: {tensor}: fp32 oracle vs fp64 oracle, worst rows (error / tensor scale): "
          + ", ".join(f"g{int(g)} {float(d[g]) / sc:.2e}" for g in worst))
    same_lists = np.array_equal(L32["point_list"].numpy(), L64["point_list"].numpy()) and np.array_equal(L32["ranges"], L64["ranges"])
    print(f"tile lists identical in both precisions: {same_lists}")
    W, H = sp["w"], sp["h"]
    gx = (W + O.TILE - 1) // O.TILE
    for g in worst[:2]:
        g = int(g)
        nz = float(L64["full_geom"][2][g][2])
        print(f"-- Gaussian {g}: row error {float(d[g]) / sc:.2e} of the tensor's scale; {tensor} gradient row fp64 "
              f"{[round(float(v), 4) for v in g64[tensor][g].flatten()[:6]]} vs fp32 {[round(float(v), 4) for v in g32[tensor][g].flatten()[:6]]}")
        print(f"   conditioning: view-space normal z = {nz:+.3f} (the surfel is seen {np.degrees(np.arccos(min(1.0, abs(nz)))):.0f} deg off its "
              f"normal), scales {[round(float(v), 4) for v in a['scales'][g]] if 'scales' in a else None}, screen radius "
              f"{int(L64['geom'].radii[g]) if L64['geom'].radii.shape[0] == n else '?'} px, centre {[round(float(v), 1) for v in L64['full_geom'][1][g]]}")
        n_diff = 0
        for t in range(L64["ranges"].shape[0]):
            ids = L64["point_list"][int(L64["ranges"][t, 0]):int(L64["ranges"][t, 1])]
            if not bool((ids == g).any()):
                continue
            yy, xx = O._tile_pixels(t, gx, W, H, torch.float64)
            dec = {}
            for name, Lx, dt in (("fp64", L64, torch.float64), ("fp32", L32, torch.float32)):
                T_, xy_, _, opa_, _ = Lx["full_geom"]
                dec[name] = decisions(xx.to(dt), yy.to(dt), T_[ids].to(dt), xy_[ids].to(dt), opa_[ids].to(dt))
            # pixels this Gaussian contributes to (in either evaluation)
            j = int(torch.nonzero(ids == g)[0])
            px_g = dec["fp64"]["contrib"][j] | dec["fp32"]["contrib"][j]
            for kind in ("blends", "use3d", "contrib", "median", "clamped"):
                diff = (dec["fp64"][kind] != dec["fp32"][kind]) & px_g[None, :]
                for e, p in torch.nonzero(diff).tolist():
                    q = dec["fp64"]
                    margin = {"blends": abs(float(q["alpha"][e, p]) - O.ALPHA_MIN) / O.ALPHA_MIN,
                              "use3d": abs(float(q["rho3d"][e, p] - q["rho2d"][e, p])) / (float(torch.minimum(q["rho3d"][e, p], q["rho2d"][e, p])) + 1e-12),
                              "contrib": abs(float(q["cum"][e, p]) - O.T_EPS) / O.T_EPS,
                              "median": abs(float(q["T_i"][e, p]) - 0.5),
                              "clamped": abs(float(q["a_raw"][e, p]) - O.ALPHA_MAX) / O.ALPHA_MAX}[kind]
                    n_diff += 1
                    print(f"   tile {t} pixel ({int(xx[p])}, {int(yy[p])}): decision `{kind}` for list entry {e} (Gaussian {int(ids[e])}) "
                          f"differs: fp64 {bool(dec['fp64'][kind][e, p])}, fp32 {bool(dec['fp32'][kind][e, p])}; fp64 margin {margin:.2e}; "
                          f"Gaussian {g} blends into this pixel with alpha {float(q['alpha'][j, p]):.4f}")


        if n_diff == 0:
            print("   no decision of any pixel this Gaussian blends into differs between fp32 and fp64: the row error is rounding, "
                  "amplified by the conditioning of the ray-splat intersection (not a flip)")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "facing2k-s0-flags3", sys.argv[2] if len(sys.argv) > 2 else "scales")
