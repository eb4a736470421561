"""The CPU-oracle side of the GPU gradient-parity tests, as jobs in worker processes.

The fp64 (and fp32) oracle gradients of a case depend only on its seeded inputs, not on anything the HIP path produces,
and they are what the GPU suite spent most of its wall time on (round 3: ~250 s of 437 s, single-process, per-tile
Python loops that do not use the host's other cores).  Every case is therefore described by a small picklable `spec`,
registered when its test module is imported, and -- when a `-m gpu` session starts on a box with a device -- all cases of
the selected tests are computed CONCURRENTLY in (at most four) spawned worker processes (no GPU work, one torch thread
each, longest first) while the GPU tests run; a test asks for its case with FARM.get(key) and blocks only if it is not done yet.
Nothing is dropped and nothing is stale: the oracle still runs live for every case, every session.  (Committing the
fp64 gradients as fixtures instead would be ~7 MB of incompressible floats next to 250 KB of existing fixtures; what IS
committed is tests/golden/oracle_grad_checksums.json -- a few sums per case, written by
tests/golden/make_oracle_checksums.py -- which pins the live oracle against silent drift.)

Without a pool (CPU sessions, a spawn failure, FARM_WORKERS=0) FARM.get computes the case in-process.
The oracle is test infrastructure: nothing under gaussmart_amd/ imports this file or oracle/.
"""
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

GRAD_NAMES = ("means3D", "opacities", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp")
DEFAULT_BG = (0.2, 0.4, 0.6)


def spec(scene="facing", n=2000, w=256, h=256, seed=0, *, radius_px=6.0, flags=3, deg=3, scale_modifier=1.0, view=0,
         bg=None, wseed=1, scaling_shift=0.0, opa_shift=0.0, opa_const=None, precomp=False, wide=None, want32=True,
         sens_tols=(), tilt=0.35):
    """One oracle case.  scene: "facing" (surfels within a moderate tilt of the camera) | "random" (any orientation).
    wide = (C, generator seed): colors_precomp [N,C] + a C-channel background instead of SH colours."""
    return dict(scene=scene, n=int(n), w=int(w), h=int(h), seed=int(seed), radius_px=float(radius_px), flags=int(flags),
                deg=int(deg), scale_modifier=float(scale_modifier), view=int(view), bg=None if bg is None else tuple(bg),
                wseed=int(wseed), scaling_shift=float(scaling_shift), opa_shift=float(opa_shift), opa_const=opa_const,
                precomp=bool(precomp), wide=None if wide is None else tuple(wide), want32=bool(want32),
                sens_tols=tuple(float(t) for t in sens_tols), tilt=float(tilt))


def build_inputs(sp):
    """-> (a, cam, bg, wc, wa): the operator inputs (f32, CPU), the camera, the background and the image weights of the
    scalar the gradients are taken of.  Deterministic: the HIP side of a test and the oracle worker both call this."""
    from conftest import facing_scene, oracle_settings
    from gaussmart_amd.synthetic import make_scene, activate, jittered_cameras
    from oracle import surfel_ref as O
    n, w, h, seed = sp["n"], sp["w"], sp["h"], sp["seed"]
    if sp["scene"] == "facing":
        p, cam = facing_scene(n, w, h, seed=seed, tilt=sp["tilt"], radius_px=sp["radius_px"])
    elif sp["scene"] == "random":
        p, cam = make_scene(n, w, h, seed=seed, radius_px=sp["radius_px"])
    else:
        raise ValueError(sp["scene"])
    p = dict(p)
    if sp["scaling_shift"]:
        p["scaling"] = p["scaling"] + sp["scaling_shift"]
    if sp["opa_shift"]:
        p["opacity"] = p["opacity"] + sp["opa_shift"]
    a = activate(p)
    if sp["opa_const"] == "mixed":     # faint haze with a near-opaque splat every 24th: pixels saturate deep inside the list
        a["opacities"] = torch.full_like(a["opacities"], 0.03)
        a["opacities"][::24] = 0.97
    elif sp["opa_const"] is not None:
        a["opacities"] = torch.full_like(a["opacities"], float(sp["opa_const"]))
    if sp["view"]:
        cam = jittered_cameras(sp["view"] + 1, w, h, seed=4, amount=0.25)[sp["view"]]
    bg = sp["bg"] if sp["bg"] is not None else DEFAULT_BG
    if sp["precomp"]:                  # precomputed colours + precomputed T (cov3D_precomp): the reference's alternates
        # (T is built in fp64 and rounded once: an fp32 evaluation differs in the last bit between CPU generations, and the
        # inputs of a case must be the same bits on the box that wrote the checksums and on the box that checks them)
        S = oracle_settings(cam, 3, torch.float64)
        geom = O.preprocess(a["means3D"].double(), a["scales"].double(), a["rotations"].double(), a["opacities"].double(),
                            a["shs"].double(), None, None, S)
        T = torch.tensor([1., 0, 0, 0, 1, 0, 0, 0, 1]).repeat(n, 1)
        T[geom.vis_idx] = geom.Tm.reshape(-1, 9).float()
        a = dict(means3D=a["means3D"], opacities=a["opacities"],
                 colors_precomp=torch.rand(n, 3, generator=torch.Generator().manual_seed(1000 + seed)), cov3D_precomp=T)
    if sp["wide"] is not None:
        C, gseed = sp["wide"]
        g = torch.Generator().manual_seed(int(gseed))
        a = dict(means3D=a["means3D"], opacities=a["opacities"], scales=a["scales"], rotations=a["rotations"],
                 colors_precomp=torch.randn(n, int(C), generator=g))
        bg = tuple(float(x) for x in torch.rand(int(C), generator=g))
    n_ch = a["colors_precomp"].shape[1] if a.get("colors_precomp") is not None else 3
    g = torch.Generator().manual_seed(sp["wseed"])
    wc, wa = torch.randn(n_ch, h, w, generator=g), torch.randn(7, h, w, generator=g)
    return a, cam, bg, wc, wa


def row_stats(gh, go, n):
    """Per-Gaussian relative error of a gradient tensor against the fp64 oracle -> (rel [n], active [n], normwise)."""
    d = (gh - go).abs().reshape(n, -1).amax(1)
    sc = float(go.abs().max())
    rown = go.reshape(n, -1).abs().amax(1)
    rel = d / (rown + 1e-6 * sc)
    act = rown > 1e-4 * sc
    return rel, act, float(d.max()) / max(sc, 1e-30)


def summarize(gh, go, n, rows=None, d=None, trim=0):
    """Statistics of the per-Gaussian relative error over `rows` (bool [n]; all if None).  `d`: the per-row absolute error
    if it is already known (the fp32 oracle's, computed in the worker) instead of |gh - go|.  `trim`: the norm-wise figure
    (a maximum) leaves out the `trim` worst rows and reports the worst of them as `trimmed_max`."""
    sc = float(go.abs().max())
    rown = go.reshape(n, -1).abs().amax(1)
    if d is None:
        d = (gh - go).abs().reshape(n, -1).amax(1)
    rel = d / (rown + 1e-6 * sc)
    act = rown > 1e-4 * sc
    sel = act if rows is None else act & rows
    keep = torch.ones(n, dtype=torch.bool) if rows is None else rows
    dk = torch.sort(d[keep], descending=True).values
    trim = min(int(trim), max(int(dk.numel()) - 1, 0))
    return dict(normwise=(float(dk[trim]) if dk.numel() else 0.0) / max(sc, 1e-30),
                trimmed_max=(float(dk[0]) if dk.numel() and trim else 0.0) / max(sc, 1e-30),
                median=float(rel[sel].median()) if sel.any() else 0.0,
                p99=float(rel[sel].quantile(0.99)) if sel.any() else 0.0, active=int(sel.sum()))


def _oracle_once(sp, a, cam, bg, wc, wa, dtype):
    from conftest import oracle_settings
    from oracle import surfel_ref as O
    S = oracle_settings(cam, sp["deg"], dtype, bg, scale_modifier=sp["scale_modifier"])
    names = [k for k in GRAD_NAMES if a.get(k) is not None]
    oin = {k: a[k].clone().to(dtype).requires_grad_(True) for k in names}
    n = a["means3D"].shape[0]
    m2d = torch.zeros(n, 3, dtype=dtype, requires_grad=True)
    c, r, am = O.rasterize(oin["means3D"], m2d, oin["opacities"], oin.get("shs"), oin.get("colors_precomp"),
                           oin.get("scales"), oin.get("rotations"), oin.get("cov3D_precomp"), settings=S, flags=sp["flags"])
    ((c * wc.to(dtype)).sum() + (am * wa.to(dtype)).sum()).backward()
    g = {k: oin[k].grad.double() for k in names}
    g["means2D"] = m2d.grad.double()
    return g, c.detach().double(), am.detach().double(), r, S


def run_case(sp):
    """The oracle side of one case -> dict of NumPy arrays / plain numbers (picklable)."""
    torch.set_num_threads(int(os.environ.get("FARM_TORCH_THREADS", "1")))
    from oracle import surfel_ref as O
    a, cam, bg, wc, wa = build_inputs(sp)
    n = a["means3D"].shape[0]
    out = {}
    if sp["want32"]:
        g32, _, _, r32, _ = _oracle_once(sp, a, cam, bg, wc, wa, torch.float32)
        out["radii32"] = r32.numpy().copy()
    go, c_o, am_o, r_o, S = _oracle_once(sp, a, cam, bg, wc, wa, torch.float64)      # (last: O.LAST describes THIS run)
    L = O.LAST
    out["grads"] = {k: v.numpy().copy() for k, v in go.items()}
    out["color"], out["allmap"], out["radii"] = c_o.numpy().copy(), am_o.numpy().copy(), r_o.numpy().copy()
    if sp["want32"]:
        out["d32"] = {k: (g32[k] - go[k]).abs().reshape(n, -1).amax(1).numpy().copy() for k in go}
        out["stats32"] = {k: summarize(g32[k], go[k], n) for k in go}
    lens = (L["ranges"][:, 1].astype(np.int64) - L["ranges"][:, 0].astype(np.int64))
    out["lists"] = dict(mean=float(lens.mean()), max=int(lens.max()), walked=int(L["n_contrib"][0].max()),
                        mean_depth=float(L["n_contrib"][0].double().mean()))
    out["ext_margin_small"] = (L["geom"].ext_margin < 1e-3).numpy().copy() if L["geom"].ext_margin.shape[0] == n else None
    if out["ext_margin_small"] is None:      # ext_margin is indexed by visible Gaussian: scatter to N
        m = np.zeros(n, bool)
        m[L["geom"].vis_idx.numpy()] = (L["geom"].ext_margin < 1e-3).numpy()
        out["ext_margin_small"] = m
    out["rect"] = L["geom"].rect.numpy().copy()
    out["sens"] = {}
    for tol in sp["sens_tols"]:
        mask, n_px = O.flip_sensitive_gaussians(*L["full_geom"], L["point_list"], L["ranges"], S, flags=sp["flags"], tol=tol)
        out["sens"][tol] = (mask.numpy().copy(), int(n_px))
    out["checksum"] = {k: [float(v.sum()), float(v.abs().sum())] for k, v in go.items()}
    return out


def _worker(sp):
    try:
        return run_case(sp)
    except BaseException as e:          # surfaces in the test that asks for the case
        import traceback
        return {"error": f"{type(e).__name__}: {e}\n{traceback.format_exc()}"}


class Farm:
    def __init__(self):
        self.specs, self.costs, self.futures, self.cache, self.done = {}, {}, {}, {}, {}
        self.pool = None

    def register(self, key, sp):
        """`key` names the case; cost estimate (longest first) = pixels x Gaussians-per-pixel proxy."""
        self.specs[key] = sp
        passes = (2 if sp["want32"] else 1) + 0.5 * len(sp["sens_tols"])
        self.costs[key] = passes * sp["n"] * max(sp["radius_px"], 1.0) ** 2 * (4.0 if sp["wide"] else 1.0)
        return key

    def start(self, keys=None):
        keys = [k for k in (self.specs if keys is None else keys) if k in self.specs and k not in self.futures]
        workers = int(os.environ.get("FARM_WORKERS", "-1"))
        if workers < 0:
            try:
                cores = len(os.sched_getaffinity(0))
            except AttributeError:
                cores = os.cpu_count() or 2
            # At most FOUR: torch's autograd engine starts a thread per visible GPU at a process's first backward(), which
            # opens the device files even for CPU-only work (no environment variable prevents it: measured,
            # scripts/dev_farm_probe.py), and the GPU boxes allow six processes on the card at once -- the test process,
            # four workers, one spare.  Tests that start GPU ranks of their own call FARM.drain() first.
            workers = max(1, min(4, cores - 3, len(keys)))
        if workers == 0 or not keys:
            return 0
        try:
            import multiprocessing as mp
            from concurrent.futures import ProcessPoolExecutor
            if self.pool is None:
                self.pool = ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("spawn"))
            for k in sorted(keys, key=lambda k: -self.costs[k]):
                self.futures[k] = self.pool.submit(_worker, self.specs[k])
        except Exception as e:          # no pool: every case is computed where it is asked for
            print(f"[oracle farm] no worker pool ({e}); cases run in-process", file=sys.stderr)
            self.pool, self.futures = None, {}
            return 0
        return workers

    def get(self, key):
        if key in self.cache:
            return self.cache[key]
        fut = self.futures.pop(key, None)
        res = self.done.pop(key, None)
        if fut is not None:
            try:
                res = fut.result()
            except Exception as e:      # a worker died (e.g. out of memory): fall back to this process
                print(f"[oracle farm] worker failed for {key} ({e}); computing in-process", file=sys.stderr)
        if res is None:
            res = run_case(self.specs[key])
        if "error" in res:
            raise RuntimeError(f"oracle case {key} failed in its worker:\n{res['error']}")
        res["grads"] = {k: torch.from_numpy(v) for k, v in res["grads"].items()}
        if "d32" in res:
            res["d32"] = {k: torch.from_numpy(v) for k, v in res["d32"].items()}
        self.cache = {key: res}         # (one case at a time: a 64-channel image is 15 MB)
        if self.pool is not None and not self.futures:
            self.shutdown()             # nothing left to compute: the idle workers need not keep the device files open
        return res

    def drain(self):
        """Wait for every outstanding case, keep the results, and end the worker processes (they hold the GPU's device
        files open): for tests that are about to start GPU processes of their own."""
        for key in list(self.futures):
            fut = self.futures.pop(key)
            try:
                self.done[key] = fut.result()
            except Exception as e:
                print(f"[oracle farm] worker failed for {key} ({e}); it will be computed in-process", file=sys.stderr)
        self.shutdown()

    def shutdown(self):
        if self.pool is not None:
            procs = list(getattr(self.pool, "_processes", {}).values())
            self.pool.shutdown(wait=False, cancel_futures=True)
            for p in procs:             # cases nobody asked for (a -k run) need not finish
                try:
                    p.kill()
                except Exception:
                    pass
            self.pool, self.futures = None, {}


FARM = Farm()
CHECKSUMS = os.path.join(ROOT, "tests", "golden", "oracle_grad_checksums.json")
_checksums = None


def check_against_committed_checksums(key, res, rtol=1e-6):
    """The live oracle result against the sums tests/golden/make_oracle_checksums.py committed (drift guard: a change of
    the oracle, of a scene builder or of torch's CPU kernels that moves the fp64 gradients shows up here, not as a
    mysteriously shifted parity statistic).  Cases absent from the file are reported, not failed."""
    global _checksums
    if _checksums is None:
        _checksums = json.load(open(CHECKSUMS)) if os.path.exists(CHECKSUMS) else {}
    want = _checksums.get(key)
    if want is None:
        return False
    for k, (s, sa) in want.items():
        got_s, got_sa = res["checksum"][k]
        assert math.isclose(got_sa, sa, rel_tol=rtol, abs_tol=1e-300), (key, k, got_sa, sa)
        assert abs(got_s - s) <= rtol * max(sa, 1e-300), (key, k, got_s, s)
    return True


# ------------------------------------------------------------------------------------------------ the parity bars
# north_star: gradients within 1e-4 relative in fp32.  What an fp32 implementation CAN reach on a given scene is set by the
# scene's conditioning (pixel coordinates ~1e3, sub-pixel and edge-on splats) and by discrete decisions (alpha >= 1/255,
# rho3d <= rho2d, T(1-alpha) < 1e-4, T > 0.5) that fp32 and fp64 may take differently on a few pairs.  Both effects hit the
# oracle's own formulas evaluated in fp32 just as hard, so the bars are stated RELATIVE to that evaluation on the same
# scene, plus absolute caps:
#     median  <= 1e-4                              (north_star's figure, for the typical Gaussian)
#     median  <= K_MED * median_fp32 + 1e-6
#     p99     <= K_TAIL * p99_fp32 + 2e-4          and <= CAP_P99
#     normwise<= K_TAIL * normwise_fp32 + 2e-4     and <= CAP_NORMWISE
# (round 3 asserted fixed bars -- normwise < 1e-3, p99 < 2e-3 -- and the shipped build passed one of them by 4 %.)
# These bars hold on the DECISION-STABLE rows: Gaussians that blend into no pixel holding a decision with a relative margin
# below 1e-3 (the oracle names them; radii that round differently count too).  On the others one flipped decision moves a
# row by a finite amount whatever the arithmetic (round 4's second seed of the 2k scene: a flip HIP takes and the fp32
# oracle does not moved one `means3D` row by 1.3e-3 of the scale): those rows only have to stay under FLIP_CAP, and at most
# max(2, 0.2 %) of them may exceed FLIP_BIG.  The norm-wise figure is a MAXIMUM over ~2,000 rows: it leaves out the worst row per
# thousand (at least one), which is held to FLIP_CAP instead -- the oracle's margin test at 1e-3 cannot name every pair an
# fp32 implementation with fast reciprocals / exponentials can flip (measured over 40 (case, tensor) entries: the kernels'
# median and p99 errors are 0.8-1.05 x the fp32 oracle's, the maxima within 1.2 x, except ONE row of one scene at 8.8 x).
def TRIM(n_rows):
    return max(1, n_rows // 1000)


K_MED, K_TAIL = 3.0, 4.0
CAP_P99, CAP_NORMWISE = 5e-3, 1e-2
FLIP_CAP, FLIP_BIG = 2e-2, 2e-3
REPORT = []      # rows (case, tensor, hip stats, fp32-oracle stats, worst bar usage): printed at session end, kept under profiles/


def check_gradient_bars(case, hip_stats, f32_stats, tensors=None, flips=None):
    """Asserts the bars above for every tensor (statistics over the decision-stable rows) and records the measured figures;
    `flips`: per tensor (largest error / scale on the flip-sensitive rows, how many of them exceed FLIP_BIG, how many there
    are).  Returns the worst fraction of a bar used."""
    worst = 0.0
    for k, s in hip_stats.items():
        if tensors is not None and k not in tensors:
            continue
        r = f32_stats[k]
        if flips is not None and k in flips:
            f_max, f_big, f_n = flips[k]
            assert f_max <= FLIP_CAP, f"{case}: {k}: a flip-sensitive row is off by {f_max:.2e} of the scale"
            assert f_big <= max(2, f_n // 500), f"{case}: {k}: {f_big} of {f_n} flip-sensitive rows beyond {FLIP_BIG:g}"
            s = dict(s, flip_rows=f_n, flip_max=f_max, flip_big=f_big)
        assert s.get("trimmed_max", 0.0) <= FLIP_CAP, f"{case}: {k}: a decision-stable row is off by {s['trimmed_max']:.2e} of the scale"
        bars = dict(median=min(1e-4, K_MED * r["median"] + 1e-6),
                    p99=min(CAP_P99, K_TAIL * r["p99"] + 2e-4),
                    normwise=min(CAP_NORMWISE, K_TAIL * r["normwise"] + 2e-4))
        used = {m: s[m] / bars[m] for m in bars}
        REPORT.append((case, k, s, r, used))
        worst = max(worst, max(used.values()))
        for m in bars:
            assert s[m] <= bars[m], (f"{case}: {k} {m} {s[m]:.3e} exceeds its bar {bars[m]:.3e} "
                                     f"(fp32 oracle on the same scene: {r[m]:.3e})")
    return worst


def format_report():
    lines = ["case | tensor | decision-stable rows, HIP vs fp64: normwise median p99 | same rows, fp32 oracle vs fp64: normwise "
             "median p99 | worst bar usage | flip-sensitive rows: count, largest error / scale, how many beyond 2e-3"]
    for case, k, s, r, used in REPORT:
        lines.append(f"{case} | {k} | {s['normwise']:.2e} {s['median']:.2e} {s['p99']:.2e} | "
                     f"{r['normwise']:.2e} {r['median']:.2e} {r['p99']:.2e} | "
                     f"{max(used.values()):.2f} ({max(used, key=used.get)}) | "
                     + (f"{s['flip_rows']} {s['flip_max']:.2e} {s['flip_big']}" if "flip_rows" in s else "-")
                     + f" | worst stable row(s) left out of normwise: {s.get('trimmed_max', 0.0):.2e}")
    return "\n".join(lines)
