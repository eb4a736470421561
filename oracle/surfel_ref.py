"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the 2D-Gaussian-surfel rasterizer hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (gaussmart_amd/, diff_surfel_rasterization/, simple_knn/) never does.

PARITY UNPINNED: the reference's arithmetic for this path lives in the un-vendored, un-pinned
submodule hbb1/diff-surfel-rasterization (/root/reference/.gitmodules:1-3; directory empty,
no commit SHA recoverable) and the reference holds no tests or golden vectors for it.  This file
restates the published 2DGS algorithm (Huang et al., SIGGRAPH 2024, eq. 8-11 + appendix) and is
anchored on what IS in the tree:
  * T-matrix construction and the (W-1)/2 pixel convention  gaussian_renderer/__init__.py:64-75
  * splat->world matrix (two scales, third axis = normal)    scene/gaussian_model.py:29-35
  * quaternion (w,x,y,z) -> rotation                         utils/general_utils.py:78-99
  * SH basis, +0.5 and clamp_min 0                           utils/sh_utils.py:57-112,
                                                             gaussian_renderer/__init__.py:86-91
  * camera matrices (row-vector / transposed convention)     scene/cameras.py:56-59
  * output channel layout of `allmap`                        gaussian_renderer/__init__.py:117-141
Everything tagged [U] is the surveyor's recollection of the submodule and lives in the constants
block below (mirrored by gaussmart_amd/csrc/gsr_constants.h).

The oracle is pure PyTorch (any float dtype, CPU), vectorised per 16x16 tile; backward comes from
autograd on a per-tile recomputation, so its gradients are exact derivatives of its own forward
(verified by torch.autograd.gradcheck in tests/test_oracle.py).  Two optional "quirk" flags
reproduce the two places where the [U] backward is not the derivative of the forward.
"""
from __future__ import annotations

import math
from typing import NamedTuple, Optional

import numpy as np
import torch

# ----------------------------------------------------------------------------- [U] constants
TILE = 16                 # BLOCK_X = BLOCK_Y
NEAR_N = 0.2              # near plane for frustum cull and per-pixel depth reject
FAR_N = 100.0             # far plane used by the distortion depth mapping
CUTOFF = 3.0              # AABB cutoff in sigmas
FILTER_SIZE = 0.707106    # sqrt(2)/2 truncated, as the literal upstream uses
FILTER_INV_SQUARE = 2.0   # 1 / FILTER_SIZE^2
ALPHA_MAX = 0.99
ALPHA_MIN = 1.0 / 255.0
T_EPS = 1e-4
AABB_MIN_EXTENT2 = 1e-4

# quirk flags (bit field, same values as GSR_FLAG_* in include/gsr.h)
QUIRK_CLAMP_PASSTHROUGH = 1   # d(min(0.99, o*G)) treated as identity even when clamped
QUIRK_FILTER_DEPTH_GRAD = 2   # low-pass branch: dL/dz also flows to Tw.x, Tw.y scaled by s
QUIRKS_UPSTREAM = QUIRK_CLAMP_PASSTHROUGH | QUIRK_FILTER_DEPTH_GRAD
QUIRK_AABB_GRAD_CUTOFF1 = 512  # screen-space centre: forward value with weights (9, 9, -1), gradient of the (1, 1, -1) form

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
         -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


class Settings(NamedTuple):
    """Same 12 fields as GaussianRasterizationSettings (gaussian_renderer/__init__.py:37-51)."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool = False
    debug: bool = False


# ----------------------------------------------------------------------------- per-Gaussian
def quat_to_rotmat(q: torch.Tensor) -> torch.Tensor:
    """(w,x,y,z) -> R, utils/general_utils.py:78-99.  The normalisation factor is detached:
    callers pass unit quaternions (scene/gaussian_model.py:109) and the [U] backward returns the
    gradient w.r.t. the normalised quaternion."""
    s = torch.rsqrt((q * q).sum(-1, keepdim=True)).detach()
    qn = q * s
    r, x, y, z = qn[:, 0], qn[:, 1], qn[:, 2], qn[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.reshape(-1, 3, 3)


def eval_sh_rgb(deg: int, shs: torch.Tensor, means3D: torch.Tensor, campos: torch.Tensor):
    """shs [N,K,3] (coefficient-major, as GaussianModel.get_features) -> rgb [N,3] and the
    clamp mask.  utils/sh_utils.py:57-112 + gaussian_renderer/__init__.py:88-91."""
    d = means3D - campos[None, :]
    d = d / d.norm(dim=1, keepdim=True)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    res = SH_C0 * shs[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * shs[:, 1] + SH_C1 * z * shs[:, 2] - SH_C1 * x * shs[:, 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * shs[:, 4] + SH_C2[1] * yz * shs[:, 5]
                   + SH_C2[2] * (2.0 * zz - xx - yy) * shs[:, 6]
                   + SH_C2[3] * xz * shs[:, 7] + SH_C2[4] * (xx - yy) * shs[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3 * xx - yy) * shs[:, 9]
                       + SH_C3[1] * xy * z * shs[:, 10]
                       + SH_C3[2] * y * (4 * zz - xx - yy) * shs[:, 11]
                       + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * shs[:, 12]
                       + SH_C3[4] * x * (4 * zz - xx - yy) * shs[:, 13]
                       + SH_C3[5] * z * (xx - yy) * shs[:, 14]
                       + SH_C3[6] * x * (xx - 3 * yy) * shs[:, 15])
    res = res + 0.5
    clamped = res < 0
    return torch.clamp_min(res, 0.0), clamped


def world2pix3(S: Settings, dtype) -> torch.Tensor:
    """projmatrix @ ndc2pix, columns (x, y, w); gaussian_renderer/__init__.py:66-75."""
    W, H = float(S.image_width), float(S.image_height)
    ndc2pix = torch.tensor([[W / 2, 0, 0, (W - 1) / 2],
                            [0, H / 2, 0, (H - 1) / 2],
                            [0, 0, 0, 1]], dtype=dtype).T          # [4,3]
    return S.projmatrix.to(dtype) @ ndc2pix


class Geom(NamedTuple):
    vis_idx: torch.Tensor      # int64 [V] indices of Gaussians that survive every cull
    Tm: torch.Tensor           # [V,3,3] rows Tu, Tv, Tw (coefficients of u, v, 1)
    xy: torch.Tensor           # [V,2] AABB centre in pixels
    normal: torch.Tensor       # [V,3] view-space normal, facing the camera
    depth: torch.Tensor        # [V] view-space z of the centre
    rgb: Optional[torch.Tensor]  # [V,C] (None when colors_precomp is used by the caller)
    radii: torch.Tensor        # int32 [N]
    rect: torch.Tensor         # int32 [N,4] (minx, miny, maxx, maxy) in tiles
    clamped: Optional[torch.Tensor]
    ext_margin: torch.Tensor   # [N] distance of the un-ceiled radius to the next integer (test aid)


def preprocess(means3D, scales, rotations, opacities, shs, colors_precomp, transmat_precomp,
               S: Settings, flags: int = 0) -> Geom:
    """[U] preprocess: cull, T matrix, AABB, tile rect, normal, SH colour.  `flags`: only QUIRK_AABB_GRAD_CUTOFF1 matters
    here (the centre keeps its forward VALUE, its gradient is the one of the weights (1, 1, -1))."""
    N = means3D.shape[0]
    dt = means3D.dtype
    V = S.viewmatrix.to(dt)
    W, H = S.image_width, S.image_height
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE

    radii = torch.zeros(N, dtype=torch.int32)
    rect = torch.zeros(N, 4, dtype=torch.int32)
    ext_margin = torch.full((N,), float("inf"), dtype=torch.float64)

    p_view_all = means3D @ V[:3, :3] + V[3, :3]
    idx0 = torch.nonzero(p_view_all[:, 2].detach() > NEAR_N).squeeze(1)      # in_frustum
    p = means3D[idx0]
    p_view = p_view_all[idx0]
    if transmat_precomp is None:
        R = quat_to_rotmat(rotations[idx0])
        mod = S.scale_modifier
        tu = R[:, :, 0] * (scales[idx0, 0:1] * mod)
        tv = R[:, :, 1] * (scales[idx0, 1:2] * mod)
        tn = R[:, :, 2]
        zeros = torch.zeros(p.shape[0], 1, dtype=dt)
        ones = torch.ones(p.shape[0], 1, dtype=dt)
        Hm = torch.stack([torch.cat([tu, zeros], 1), torch.cat([tv, zeros], 1),
                          torch.cat([p, ones], 1)], dim=1)                    # [n,3,4]
        M = Hm @ world2pix3(S, dt)                                            # [n,(u,v,1),(x,y,w)]
        Tm = M.permute(0, 2, 1)                                               # rows Tu,Tv,Tw
        normal = tn @ V[:3, :3]
    else:
        Tm = transmat_precomp[idx0].reshape(-1, 3, 3)
        normal = torch.zeros(p.shape[0], 3, dtype=dt)
        normal[:, 2] = 1.0
    Tu, Tv, Tw = Tm[:, 0], Tm[:, 1], Tm[:, 2]

    cosv = -(p_view * normal).sum(-1)
    ok = cosv.detach() != 0
    normal = torch.where(cosv.detach()[:, None] > 0, normal, -normal)

    t = torch.tensor([CUTOFF * CUTOFF, CUTOFF * CUTOFF, -1.0], dtype=dt)
    d = (t * Tw * Tw).sum(-1)
    ok = ok & (d.detach() != 0)
    dsafe = torch.where(d.detach() != 0, d, torch.ones_like(d))
    f = t[None, :] / dsafe[:, None]
    cx = (f * Tu * Tw).sum(-1)
    cy = (f * Tv * Tw).sum(-1)
    h0x = cx * cx - (f * Tu * Tu).sum(-1)
    h0y = cy * cy - (f * Tv * Tv).sum(-1)
    hx = torch.sqrt(torch.clamp_min(h0x.detach(), AABB_MIN_EXTENT2))
    hy = torch.sqrt(torch.clamp_min(h0y.detach(), AABB_MIN_EXTENT2))
    rad_f = torch.maximum(torch.maximum(hx, hy), torch.tensor(CUTOFF * FILTER_SIZE, dtype=dt))
    radius = torch.ceil(rad_f)
    xy = torch.stack([cx, cy], -1)
    if flags & QUIRK_AABB_GRAD_CUTOFF1:
        t1 = torch.tensor([1.0, 1.0, -1.0], dtype=dt)
        d1 = (t1 * Tw * Tw).sum(-1)
        d1safe = torch.where(d1.detach() != 0, d1, torch.ones_like(d1))
        f1 = torch.where((d1.detach() != 0)[:, None], t1[None, :] / d1safe[:, None], torch.zeros_like(f))
        xy1 = torch.stack([(f1 * Tu * Tw).sum(-1), (f1 * Tv * Tw).sum(-1)], -1)
        xy = xy1 + (xy - xy1).detach()

    cxd, cyd = cx.detach(), cy.detach()
    def tdiv(v):  # (int)(v / TILE): truncation toward zero, like the C cast
        return torch.trunc(v / TILE).to(torch.int64)
    minx = tdiv(cxd - radius).clamp(0, gx)
    miny = tdiv(cyd - radius).clamp(0, gy)
    maxx = tdiv(cxd + radius + (TILE - 1)).clamp(0, gx)
    maxy = tdiv(cyd + radius + (TILE - 1)).clamp(0, gy)
    tiles = (maxx - minx) * (maxy - miny)
    ok = ok & (tiles > 0) & torch.isfinite(cxd) & torch.isfinite(cyd)

    sel = torch.nonzero(ok).squeeze(1)
    vis_idx = idx0[sel]
    radii[vis_idx] = radius[sel].to(torch.int32)
    rect[vis_idx] = torch.stack([minx, miny, maxx, maxy], -1)[sel].to(torch.int32)
    ext_margin[idx0] = (radius - rad_f).to(torch.float64)

    rgb = clamped = None
    if colors_precomp is None and shs is not None:
        rgb, clamped = eval_sh_rgb(S.sh_degree, shs[vis_idx], means3D[vis_idx], S.campos.to(dt))
    return Geom(vis_idx, Tm[sel], xy[sel], normal[sel], p_view[sel, 2], rgb, radii, rect,
                clamped, ext_margin)


# ----------------------------------------------------------------------------- binning
def bin_tiles(xy_unused, radii: np.ndarray, rect: np.ndarray, depth_all: np.ndarray, grid_x: int):
    """[U] duplicateWithKeys + stable sort + identifyTileRanges, in NumPy.

    radii int32 [N], rect int32 [N,4], depth_all float32 [N].  Emission order: Gaussian index
    ascending, tiles row-major inside the rect.  key = (tile << 32) | float32 bits of depth.
    Returns keys_sorted u64 [D], point_list u32 [D], ranges u32 [tiles,2] (zeros when empty).
    """
    dbits = depth_all.astype(np.float32).view(np.uint32).astype(np.uint64)
    rect = rect.astype(np.int64)
    w = np.where(radii > 0, rect[:, 2] - rect[:, 0], 0)
    h = np.where(radii > 0, rect[:, 3] - rect[:, 1], 0)
    cnt = np.maximum(w, 0) * np.maximum(h, 0)
    total = int(cnt.sum())
    if total == 0:
        return np.zeros(0, np.uint64), np.zeros(0, np.uint32)
    gid = np.repeat(np.arange(radii.shape[0]), cnt)              # Gaussian of each instance
    first = np.repeat(np.cumsum(cnt) - cnt, cnt)
    local = np.arange(total) - first                             # row-major index inside the rect
    wy = np.repeat(w, cnt)
    ty = rect[gid, 1] + local // wy
    tx = rect[gid, 0] + local % wy
    keys = ((ty * grid_x + tx).astype(np.uint64) << np.uint64(32)) | dbits[gid]
    vals = gid.astype(np.uint32)
    order = np.argsort(keys, kind="stable")
    return keys[order], vals[order]


def tile_ranges(keys_sorted: np.ndarray, n_tiles: int) -> np.ndarray:
    ranges = np.zeros((n_tiles, 2), dtype=np.uint32)
    if keys_sorted.size == 0:
        return ranges
    tid = (keys_sorted >> np.uint64(32)).astype(np.int64)
    starts = np.nonzero(np.r_[True, tid[1:] != tid[:-1]])[0]
    ends = np.r_[starts[1:], tid.size]
    ranges[tid[starts], 0] = starts
    ranges[tid[starts], 1] = ends
    return ranges


# ----------------------------------------------------------------------------- per-tile
def _tile_eval(px, py, Tm, xy, nrm, opa, rgb, flags, margins=False, flip_tol=None):
    """All (Gaussian, pixel) pairs of one tile.  Tm [L,3,3], xy [L,2], nrm [L,3], opa [L],
    rgb [L,C]; px, py [P].  Returns the per-pixel accumulators (differentiable).
    `flip_tol`: also return "sensitive" [L] -- list entries that blend (or nearly blend) into a pixel where ANY
    discrete decision of the walk (alpha >= 1/255, rho3d <= rho2d, T(1-alpha) < 1e-4, T > 0.5, depth >= near,
    alpha clamp) has a relative margin below flip_tol, i.e. could be taken differently in fp32."""
    L = Tm.shape[0]
    dt = Tm.dtype
    P = px.shape[0]
    if L == 0:
        z = torch.zeros(P, dtype=dt)
        zi = torch.zeros(P, dtype=torch.int64)
        inf = torch.full((P,), float("inf"), dtype=dt)
        out = dict(C=torch.zeros(P, rgb.shape[1], dtype=dt), T=torch.ones(P, dtype=dt), D=z,
                   N=torch.zeros(P, 3, dtype=dt), med=z, dist=z, M1=z, M2=z, n_contrib=zi,
                   med_contrib=zi - 1, first=zi)
        if margins:
            out.update(m_alpha=inf, m_term=inf, m_med=inf, m_rho=inf)
        if flip_tol is not None:
            out["sensitive"] = torch.zeros(0, dtype=torch.bool)
        return out
    Tu, Tv, Tw = Tm[:, 0, :], Tm[:, 1, :], Tm[:, 2, :]
    pxb, pyb = px[None, :], py[None, :]
    k0 = pxb * Tw[:, 0:1] - Tu[:, 0:1]
    k1 = pxb * Tw[:, 1:2] - Tu[:, 1:2]
    k2 = pxb * Tw[:, 2:3] - Tu[:, 2:3]
    l0 = pyb * Tw[:, 0:1] - Tv[:, 0:1]
    l1 = pyb * Tw[:, 1:2] - Tv[:, 1:2]
    l2 = pyb * Tw[:, 2:3] - Tv[:, 2:3]
    p0 = k1 * l2 - k2 * l1
    p1 = k2 * l0 - k0 * l2
    p2 = k0 * l1 - k1 * l0
    valid = p2.detach() != 0
    p2s = torch.where(valid, p2, torch.ones_like(p2))
    sx_v, sy_v = p0.detach() / p2s.detach(), p1.detach() / p2s.detach()       # the values (IEEE division, as the reference's)
    dx = xy[:, 0:1] - pxb
    dy = xy[:, 1:2] - pyb
    rho2d = FILTER_INV_SQUARE * (dx * dx + dy * dy)
    rho3d_v = sx_v * sx_v + sy_v * sy_v
    use3d = rho3d_v <= rho2d.detach()
    # The reference differentiates s = p / p.z only inside its `rho3d <= rho2d` branch.  Autograd through torch.where would
    # evaluate the quotient's derivative (-p / p.z^2) on the other branch too and multiply it by zero: for a surfel whose
    # scales collapsed (p.z ~ 1e-33 and below in fp32) that is inf x 0 = NaN in rows the reference leaves finite.  So the
    # differentiable quotient is formed only where the branch is taken (same values, same gradients there).
    one, zero = torch.ones_like(p2s), torch.zeros_like(p2s)
    den = torch.where(use3d, p2s, one)
    sx, sy = torch.where(use3d, p0, zero) / den, torch.where(use3d, p1, zero) / den
    rho3d = sx * sx + sy * sy
    rho = torch.where(use3d, rho3d, rho2d)
    z3d = sx * Tw[:, 0:1] + sy * Tw[:, 1:2] + Tw[:, 2:3]
    if flags & QUIRK_FILTER_DEPTH_GRAD:
        zq = sx_v * Tw[:, 0:1] + sy_v * Tw[:, 1:2]
        z2d = Tw[:, 2:3] + (zq - zq.detach())
    else:
        z2d = Tw[:, 2:3].expand_as(z3d)
    depth = torch.where(use3d, z3d, z2d)
    valid = valid & (depth.detach() >= NEAR_N)
    power = -0.5 * rho
    valid = valid & ~(power.detach() > 0)
    G = torch.exp(power)
    a_raw = opa[:, None] * G
    a_cl = torch.clamp_max(a_raw, ALPHA_MAX)
    alpha = a_raw + (a_cl - a_raw).detach() if flags & QUIRK_CLAMP_PASSTHROUGH else a_cl
    pre_alpha_valid = valid
    valid = valid & (alpha.detach() >= ALPHA_MIN)

    a_eff = torch.where(valid, alpha.detach(), torch.zeros_like(alpha))
    cum = torch.cumprod(1 - a_eff, dim=0)                     # test_T for every valid pair
    term = valid & (cum < T_EPS)
    ar = torch.arange(L)[:, None]
    first = torch.where(term.any(0), term.to(torch.uint8).argmax(0), torch.full((P,), L))
    contrib = valid & (ar < first[None, :])

    a_c = torch.where(contrib, alpha, torch.zeros_like(alpha))
    cumc = torch.cumprod(1 - a_c, dim=0)
    T_i = torch.cat([torch.ones(1, P, dtype=dt), cumc[:-1]], 0)
    T_final = cumc[-1]
    w = a_c * T_i

    zsafe = torch.where(contrib, depth, torch.ones_like(depth))
    m = FAR_N / (FAR_N - NEAR_N) * (1 - NEAR_N / zsafe)
    mw = m * w
    mmw = m * mw
    M1_i = torch.cumsum(mw, 0) - mw
    M2_i = torch.cumsum(mmw, 0) - mmw
    A_i = 1 - T_i
    dist = ((m * m * A_i + M2_i - 2 * m * M1_i) * w).sum(0)
    D = (zsafe * w).sum(0)
    M1 = mw.sum(0)
    M2 = mmw.sum(0)
    Nrm = (nrm[:, None, :] * w[:, :, None]).sum(0)            # [P,3]
    C = (rgb[:, None, :] * w[:, :, None]).sum(0)              # [P,C]

    med_ok = contrib & (T_i.detach() > 0.5)
    has_med = med_ok.any(0)
    last_med = (L - 1) - torch.flip(med_ok, [0]).to(torch.uint8).argmax(0)
    med_depth = torch.where(has_med, torch.gather(zsafe, 0, last_med[None, :])[0],
                            torch.zeros_like(T_final))
    has_c = contrib.any(0)
    last_c = (L - 1) - torch.flip(contrib, [0]).to(torch.uint8).argmax(0)
    n_contrib = torch.where(has_c, last_c + 1, torch.zeros_like(last_c))
    med_contrib = torch.where(has_med, last_med + 1, torch.full_like(last_med, -1))

    out = dict(C=C, T=T_final, D=D, N=Nrm, med=med_depth, dist=dist, M1=M1, M2=M2,
               n_contrib=n_contrib, med_contrib=med_contrib, first=first)
    if margins:
        big = torch.full_like(alpha, float("inf")).detach()
        live = ar <= first[None, :]
        a_d = alpha.detach()
        out["m_alpha"] = torch.where(pre_alpha_valid & live, (a_d - ALPHA_MIN).abs() / ALPHA_MIN, big).amin(0)
        out["m_term"] = torch.where(valid & live, (cum - T_EPS).abs() / T_EPS, big).amin(0)
        out["m_med"] = torch.where(contrib, (T_i.detach() - 0.5).abs(), big).amin(0)
        out["m_rho"] = torch.where(contrib, (rho3d_v - rho2d.detach()).abs()
                                   / (rho.detach() + 1e-12), big).amin(0)
    if flip_tol is not None:
        big = torch.full_like(alpha, float("inf")).detach()
        live = ar <= first[None, :]
        a_d, araw_d, dep_d = alpha.detach(), a_raw.detach(), depth.detach()
        near_alpha = pre_alpha_valid & live
        pair_m = torch.where(near_alpha, (a_d - ALPHA_MIN).abs() / ALPHA_MIN, big)
        pair_m = torch.minimum(pair_m, torch.where(valid & live, (cum - T_EPS).abs() / T_EPS, big))
        pair_m = torch.minimum(pair_m, torch.where(contrib, (T_i.detach() - 0.5).abs(), big))
        pair_m = torch.minimum(pair_m, torch.where(contrib, (rho3d_v - rho2d.detach()).abs()
                                                   / (rho.detach() + 1e-12), big))
        pair_m = torch.minimum(pair_m, torch.where(contrib, (araw_d - ALPHA_MAX).abs() / ALPHA_MAX, big))
        pair_m = torch.minimum(pair_m, torch.where(live & (p2.detach() != 0) & (a_d >= ALPHA_MIN * (1 - flip_tol)),
                                                   (dep_d - NEAR_N).abs() / NEAR_N, big))
        unstable_px = pair_m.amin(0) < flip_tol                                   # [P]
        blends = near_alpha & (a_d >= ALPHA_MIN * (1 - flip_tol))                 # [L,P]
        out["sensitive"] = (blends & unstable_px[None, :]).any(1)
        out["unstable_px"] = unstable_px
    return out


class RenderOut(NamedTuple):
    color: torch.Tensor      # [C,H,W]
    allmap: torch.Tensor     # [7,H,W]
    final_T: torch.Tensor    # [3,H,W]  (T, M1, M2)
    n_contrib: torch.Tensor  # int64 [2,H,W] (last contributor, median contributor; -1 = none)
    luse: list               # per tile: how much of the list any pixel reaches
    margins: Optional[dict]


def _tile_pixels(t, gx, W, H, dt):
    ty, tx = divmod(t, gx)
    ys = torch.arange(ty * TILE, min((ty + 1) * TILE, H))
    xs = torch.arange(tx * TILE, min((tx + 1) * TILE, W))
    yy, xx = torch.meshgrid(ys, xs, indexing="ij")
    return yy.reshape(-1), xx.reshape(-1)


def _assemble(o, bg):
    color = o["C"] + o["T"][:, None] * bg[None, :]
    allmap = torch.stack([o["D"], 1 - o["T"], o["N"][:, 0], o["N"][:, 1], o["N"][:, 2],
                          o["med"], o["dist"]], 0)
    return color.T, allmap


def render_tiles(geom_T, geom_xy, geom_nrm, geom_opa, geom_rgb, point_list, ranges, S: Settings,
                 flags=QUIRKS_UPSTREAM, margins=False, tiles=None) -> RenderOut:
    """Forward over all (or the listed) tiles, no autograd graph kept.  geom_* are indexed by the
    ids stored in point_list."""
    W, H = S.image_width, S.image_height
    gx = (W + TILE - 1) // TILE
    dt = geom_T.dtype
    C = geom_rgb.shape[1]
    bg = S.bg.to(dt)
    color = torch.zeros(C, H, W, dtype=dt)
    allmap = torch.zeros(7, H, W, dtype=dt)
    final_T = torch.zeros(3, H, W, dtype=dt)
    n_contrib = torch.zeros(2, H, W, dtype=torch.int64)
    marg = {k: torch.full((H, W), float("inf"), dtype=dt) for k in ("m_alpha", "m_term", "m_med", "m_rho")} \
        if margins else None
    n_tiles = ranges.shape[0]
    luse = [0] * n_tiles
    with torch.no_grad():
        for t in (range(n_tiles) if tiles is None else tiles):
            yy, xx = _tile_pixels(t, gx, W, H, dt)
            ids = point_list[int(ranges[t, 0]):int(ranges[t, 1])]
            o = _tile_eval(xx.to(dt), yy.to(dt), geom_T[ids], geom_xy[ids], geom_nrm[ids],
                           geom_opa[ids], geom_rgb[ids], flags, margins)
            c, a = _assemble(o, bg)
            color[:, yy, xx] = c
            allmap[:, yy, xx] = a
            final_T[0, yy, xx] = o["T"]; final_T[1, yy, xx] = o["M1"]; final_T[2, yy, xx] = o["M2"]
            n_contrib[0, yy, xx] = o["n_contrib"]; n_contrib[1, yy, xx] = o["med_contrib"]
            luse[t] = int(min(ids.shape[0], int(o["first"].max()) + 1)) if ids.shape[0] else 0
            if margins:
                for k in marg:
                    marg[k][yy, xx] = o[k]
    return RenderOut(color, allmap, final_T, n_contrib, luse, marg)


def flip_sensitive_gaussians(geom_T, geom_xy, geom_nrm, geom_opa, geom_rgb, point_list, ranges, S: Settings,
                             flags=QUIRKS_UPSTREAM, tol=1e-4, tiles=None):
    """Bool [N]: Gaussians that blend into at least one pixel whose walk holds a decision with margin < tol (see
    _tile_eval).  A fp32 evaluation may take such a decision the other way, which changes the gradient of EVERY
    Gaussian blended into that pixel by a finite amount; all other Gaussians see the same decisions in fp32 and fp64,
    so their gradients differ by rounding only.  Also returns the number of unstable pixels."""
    W, H = S.image_width, S.image_height
    gx = (W + TILE - 1) // TILE
    dt = geom_T.dtype
    mask = torch.zeros(geom_T.shape[0], dtype=torch.bool)
    n_px = 0
    with torch.no_grad():
        for t in (range(ranges.shape[0]) if tiles is None else tiles):
            ids = point_list[int(ranges[t, 0]):int(ranges[t, 1])]
            if ids.shape[0] == 0:
                continue
            yy, xx = _tile_pixels(t, gx, W, H, dt)
            o = _tile_eval(xx.to(dt), yy.to(dt), geom_T[ids], geom_xy[ids], geom_nrm[ids], geom_opa[ids],
                           geom_rgb[ids], flags, flip_tol=tol)
            mask[ids[o["sensitive"]]] = True
            n_px += int(o["unstable_px"].sum())
    return mask, n_px


def render_tiles_backward(geom_T, geom_xy, geom_nrm, geom_opa, geom_rgb, point_list, ranges, luse,
                          S: Settings, dL_dcolor, dL_dallmap, flags=QUIRKS_UPSTREAM, tiles=None):
    """Per-tile recompute + autograd.  Returns gradients w.r.t. the five geom arrays."""
    W, H = S.image_width, S.image_height
    gx = (W + TILE - 1) // TILE
    dt = geom_T.dtype
    bg = S.bg.to(dt)
    g = [torch.zeros_like(x) for x in (geom_T, geom_xy, geom_nrm, geom_opa, geom_rgb)]
    for t in (range(ranges.shape[0]) if tiles is None else tiles):
        n = luse[t]
        if n == 0:
            continue
        yy, xx = _tile_pixels(t, gx, W, H, dt)
        ids = point_list[int(ranges[t, 0]):int(ranges[t, 0]) + n]
        leaves = [x[ids].detach().requires_grad_(True)
                  for x in (geom_T, geom_xy, geom_nrm, geom_opa, geom_rgb)]
        with torch.enable_grad():
            o = _tile_eval(xx.to(dt), yy.to(dt), *leaves, flags)
            c, a = _assemble(o, bg)
            scalar = (c * dL_dcolor[:, yy, xx]).sum() + (a * dL_dallmap[:, yy, xx]).sum()
        grads = torch.autograd.grad(scalar, leaves, allow_unused=True)
        for acc, gr in zip(g, grads):
            if gr is not None:
                acc.index_add_(0, ids, gr)
    return g


# ----------------------------------------------------------------------------- operator
LAST = {}     # intermediate results of the most recent rasterize() call (binning, list use, image state): test aid


class _OracleRasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, opacities, shs, colors_precomp, scales, rotations,
                cov3D_precomp, S, flags, tiles=None):
        N = means3D.shape[0]
        inputs = dict(means3D=means3D, opacities=opacities, shs=shs, colors_precomp=colors_precomp,
                      scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
        leaves = {k: (v.detach().requires_grad_(True) if v is not None else None)
                  for k, v in inputs.items()}
        with torch.enable_grad():
            geom = preprocess(leaves["means3D"], leaves["scales"], leaves["rotations"],
                              leaves["opacities"], leaves["shs"], leaves["colors_precomp"],
                              leaves["cov3D_precomp"], S, flags)
            rgb_v = geom.rgb if geom.rgb is not None else leaves["colors_precomp"][geom.vis_idx]
            opa_v = leaves["opacities"][geom.vis_idx, 0]
        dt = means3D.dtype
        # scatter to N-indexed arrays so point_list can hold Gaussian ids
        def full(x):
            out = torch.zeros((N,) + tuple(x.shape[1:]), dtype=x.dtype)
            out[geom.vis_idx] = x.detach()
            return out
        depth_all = np.zeros(N, np.float32)
        depth_all[geom.vis_idx.numpy()] = geom.depth.detach().to(torch.float32).numpy()
        gx = (S.image_width + TILE - 1) // TILE
        gy = (S.image_height + TILE - 1) // TILE
        keys, plist = bin_tiles(None, geom.radii.numpy(), geom.rect.numpy(), depth_all, gx)
        ranges = tile_ranges(keys, gx * gy)
        plist_t = torch.from_numpy(plist.astype(np.int64))
        full_geom = [full(geom.Tm), full(geom.xy), full(geom.normal), full(opa_v), full(rgb_v)]
        out = render_tiles(*full_geom, plist_t, ranges, S, flags, tiles=tiles)
        ctx.tiles = tiles
        LAST.clear()
        LAST.update(full_geom=full_geom, point_list=plist_t, ranges=ranges, luse=out.luse, n_contrib=out.n_contrib,
                    final_T=out.final_T, geom=geom)
        ctx.S, ctx.flags, ctx.geom, ctx.leaves = S, flags, geom, leaves
        ctx.stage1 = (geom.Tm, geom.xy, geom.normal, opa_v, rgb_v)
        ctx.full_geom, ctx.plist, ctx.ranges, ctx.luse = full_geom, plist_t, ranges, out.luse
        ctx.keys = keys
        ctx.aux = out
        ctx.mark_non_differentiable(geom.radii)
        return out.color, geom.radii, out.allmap

    @staticmethod
    def backward(ctx, dL_dcolor, _dradii, dL_dallmap):
        S, geom = ctx.S, ctx.geom
        gT, gxy, gn, go, gc = render_tiles_backward(*ctx.full_geom, ctx.plist, ctx.ranges, ctx.luse,
                                                    S, dL_dcolor, dL_dallmap, ctx.flags, tiles=ctx.tiles)
        N = gT.shape[0]
        vi = geom.vis_idx
        # [U] densification "hack": means2D.grad = (dL/dTu.z * Tw.z * W/2, dL/dTv.z * Tw.z * H/2, 0)
        # taken from the RAW render-backward dL/dT, consumer scene/gaussian_model.py:551-553
        g2d = torch.zeros(N, 3, dtype=gT.dtype)
        tw_z = ctx.full_geom[0][:, 2, 2]
        g2d[:, 0] = gT[:, 0, 2] * tw_z * 0.5 * S.image_width
        g2d[:, 1] = gT[:, 1, 2] * tw_z * 0.5 * S.image_height
        vis_mask = torch.zeros(N, dtype=torch.bool); vis_mask[vi] = True
        g2d[~vis_mask] = 0
        names = ["means3D", "opacities", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp"]
        wrt = [ctx.leaves[k] for k in names if ctx.leaves[k] is not None]
        outs = [o for o in ctx.stage1]
        gouts = [gT[vi], gxy[vi], gn[vi], go[vi], gc[vi]]
        keep = [(o, g) for o, g in zip(outs, gouts) if o.requires_grad]
        res = torch.autograd.grad([o for o, _ in keep], wrt, [g for _, g in keep], allow_unused=True,
                                  retain_graph=True)
        res = [r if r is not None else torch.zeros_like(w) for r, w in zip(res, wrt)]
        gmap = dict(zip([k for k in names if ctx.leaves[k] is not None], res))
        return (gmap.get("means3D"), g2d, gmap.get("opacities"), gmap.get("shs"),
                gmap.get("colors_precomp"), gmap.get("scales"), gmap.get("rotations"),
                gmap.get("cov3D_precomp"), None, None, None)


def rasterize(means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None,
              rotations=None, cov3D_precomp=None, *, settings: Settings, flags=QUIRKS_UPSTREAM, tiles=None):
    """Oracle with the operator signature of GaussianRasterizer.forward
    (call site gaussian_renderer/__init__.py:97-106).  Returns (color, radii, allmap).
    `tiles` (test aid): composite, and backpropagate through, only these tiles; the images stay zero elsewhere."""
    if (shs is None) == (colors_precomp is None):
        raise Exception("Please provide excatly one of either SHs or precomputed colors!")
    if ((scales is None or rotations is None) and cov3D_precomp is None) or \
            ((scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
    return _OracleRasterize.apply(means3D, means2D, opacities, shs, colors_precomp, scales,
                                  rotations, cov3D_precomp, settings, flags, tiles)


class OracleRasterizer(torch.nn.Module):
    """Duck-type of diff_surfel_rasterization.GaussianRasterizer backed by the oracle."""

    def __init__(self, raster_settings, flags=QUIRKS_UPSTREAM):
        super().__init__()
        self.raster_settings = raster_settings
        self.flags = flags

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None,
                rotations=None, cov3D_precomp=None):
        rs = self.raster_settings
        S = Settings(*[getattr(rs, f) for f in Settings._fields])
        return rasterize(means3D, means2D, opacities, shs, colors_precomp, scales, rotations,
                         cov3D_precomp, settings=S, flags=self.flags)
