"""The reference's whole optimisation schedule, end to end through train(), on a synthetic stand-in of a BASELINE config.

What runs is train.py:90-216 as gaussmart_amd/trainer.py restates it, with OptimizationParams defaults
(arguments/__init__.py:76-95): SH degree +1 every 1,000 iterations, densification statistics every iteration and
densify / prune every 100 from 500 to 15,000, size threshold 20 px after the first opacity reset, opacity reset every
3,000, lambda_dist from 3,000 (0 by default), lambda_normal from 7,000, no optimiser step on the very last iteration.
The scene is synthetic (there is no dataset in this environment): a target model rendered from `--views` jittered
cameras gives the ground-truth images, the trained model starts from a perturbed copy at SH degree 0.

    python scripts/full_schedule_train.py --preset scan24 --iterations 30000 --out gpurun_out/full_scan24.json

Prints one line per `--log-every` iterations (iteration, N, it/s since the last line, loss, peak memory) and writes a
JSON summary: it/s of the densification phase and of the rest SEPARATELY, N over time, PSNR before / after (training
views and two held-out views), peak HBM, and whether any parameter or Adam moment ever was non-finite.

Under torch.distributed.run (WORLD_SIZE > 1) the same schedule runs view-parallel: each rank renders its shard of every
epoch's shuffle, gradients and densification statistics are exchanged, and every log point also checks that
the replicas are BIT-identical.  GSR_BENCH_BACKEND=gloo GSR_BENCH_SINGLE_DEVICE=1 puts all ranks on one GPU (rehearsal).
"""
import argparse
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussmart_amd import rasterizer as R                                     # noqa: E402
from gaussmart_amd.gaussian_model import GaussianModel                       # noqa: E402
from gaussmart_amd.gaussian_renderer import render                           # noqa: E402
from gaussmart_amd.losses import psnr                                        # noqa: E402
from gaussmart_amd.params import OptimizationParams, PipelineParams          # noqa: E402
from gaussmart_amd.synthetic import make_scene, perturb, jittered_cameras    # noqa: E402
from gaussmart_amd.trainer import train, TrainState                          # noqa: E402
from gaussmart_amd.view_parallel import ViewParallel                         # noqa: E402

PRESETS = {   # name: (N, W, H, mean projected 3-sigma radius in px) -- bench.py's presets
    "c1": (2_000, 256, 256, 6.0),
    "small": (30_000, 400, 300, 6.0),
    "scan24": (300_000, 1600, 1200, 17.0),
    "bicycle": (5_000_000, 1237, 822, 9.0),
    "headline": (1_000_000, 1920, 1080, 6.0),
}


def all_finite(model, report=None):
    names = ["xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation"]
    ts = [(n, p.detach()) for n, p in zip(names, model.parameters())]
    for n, p in zip(names, model.parameters()):
        st = model.optimizer.state.get(p, {})
        ts += [(f"{n}.{k}", st[k]) for k in ("exp_avg", "exp_avg_sq") if k in st]
    ok = bool(torch.stack([torch.isfinite(t).all() for _, t in ts if t.numel()]).all().item())
    if not ok and report is not None:        # which tensors, how many rows, and the first offending Gaussian
        for n, t in ts:
            bad = ~torch.isfinite(t.reshape(t.shape[0], -1)).all(1)
            if bool(bad.any()):
                i = int(torch.nonzero(bad)[0])
                report(f"    non-finite: {n}: {int(bad.sum())} rows, first Gaussian {i}: xyz {model._xyz[i].tolist()} scaling "
                       f"{model._scaling[i].tolist()} opacity {model._opacity[i].tolist()} rotation {model._rotation[i].tolist()}")
    return ok


def run(preset, iterations, views, log_every, seed, extent, schedule_iterations=None, quiet=False, model_out=None,
        start_fraction=1.0, grad_threshold=None, watch_from=None,
        lambda_normal=None, lambda_dist=None, depth_ratio=None, white_background=False, mode="fused", profile_tail=0):
    """-> summary dict.  `schedule_iterations`: opt.iterations (the lr schedule's horizon and the one iteration that
    takes no optimiser step); default = `iterations`, i.e. the run IS the whole schedule."""
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    single = os.environ.get("GSR_BENCH_SINGLE_DEVICE", "0") == "1"
    dev = torch.device("cuda:0" if (world == 1 or single) else f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}")
    torch.cuda.set_device(dev)
    if world > 1 and not torch.distributed.is_initialized():
        torch.distributed.init_process_group(os.environ.get("GSR_BENCH_BACKEND", "nccl"))
    quiet = quiet or rank != 0
    n, w, h, radius = PRESETS[preset]
    params, _ = make_scene(n, w, h, seed=seed, radius_px=radius)
    cams = jittered_cameras(views + 2, w, h, seed=seed, device=dev, amount=0.3)
    pipe, opt = PipelineParams(), OptimizationParams()
    bg = torch.ones(3, device=dev) if white_background else torch.zeros(3, device=dev)
    opt.iterations = int(schedule_iterations or iterations)
    # the reference's other documented settings (README: --lambda_dist 100 / 1000, --depth_ratio 1 for bounded scenes;
    # scripts/dtu_eval.py:45 trains with lambda_normal = lambda_dist = 0)
    if lambda_normal is not None:
        opt.lambda_normal = float(lambda_normal)
    if lambda_dist is not None:
        opt.lambda_dist = float(lambda_dist)
    if depth_ratio is not None:
        pipe.depth_ratio = float(depth_ratio)
    dropin = mode == "dropin"
    if dropin:
        # what a PYTHONPATH swap under the reference's train.py runs (bench.py --mode dropin): the
        # reference-signature operator behind torch activations, render()'s torch post-processing, torch L1 + SSIM and the
        # regularizers as train.py:132-143, torch.optim.Adam over the six parameter groups
        pipe.fused_activations = pipe.factored_sh_grad = False
        pipe.reference_objective = True
    if grad_threshold is not None:
        opt.densify_grad_threshold = float(grad_threshold)
    target = GaussianModel(3, device=dev)
    target.create_from_params(params)
    with torch.no_grad():
        for c in cams:
            c.original_image = render(c, target, pipe, bg, surface_maps=False)["render"].clamp(0, 1).contiguous()
    del target
    train_cams, held_out = cams[:views], cams[views:]
    m = GaussianModel(3, device=dev)
    # the reference starts at SH degree 0 and raises it every 1,000 iterations (scene/gaussian_model.py:125-127)
    start = perturb(params, pos=0.02, log_scale=0.2, opa=0.5, color=0.3)
    if start_fraction < 1.0:
        # an SfM-like sparse start: every k-th Gaussian, the survivors widened so that they still cover the frame; the
        # densification phase has to grow the model back (N over time is in the trace)
        k = max(int(round(1.0 / start_fraction)), 1)
        start = {name: t[::k].contiguous() for name, t in start.items()}
        start["scaling"] = start["scaling"] + 0.5 * math.log(k)
    m.create_from_params(start, active_sh_degree=0)
    if dropin:
        m.use_fused_adam = False
    m.training_setup(opt)
    vp = ViewParallel(m, overlap_local=not dropin)

    def mean_psnr(cs):
        with torch.no_grad():
            vp.finish()
            return float(torch.stack([psnr(render(c, m, pipe, bg, surface_maps=False)["render"].clamp(0, 1)[None],
                                           c.original_image[None]).mean() for c in cs]).mean())

    before = (mean_psnr(train_cams), mean_psnr(held_out))
    replicas = []         # per log point: are the ranks' parameters bit-identical (always True on one rank)
    trace = []            # (iteration, N, seconds since start, loss, SH degree, finite)
    phase_t = {}          # iteration -> wall seconds at that point (synchronised)
    marks = sorted({min(opt.densify_until_iter, iterations), iterations})
    state = TrainState(seed)
    last_losses = {}
    torch.cuda.reset_peak_memory_stats()
    torch.cuda.synchronize()
    t0 = time.time()
    tl = [t0, 0]

    prev = {}

    def watch(it):
        """Diagnostic (--watch-from): after every iteration from `watch_from`, look for the first non-finite parameter row
        and print that Gaussian's parameters and Adam moments as they were ONE iteration earlier and as they are now."""
        vp.finish()
        torch.cuda.synchronize()
        names = ["xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation"]
        cur = {}
        for nm, p in zip(names, m.parameters()):
            st = m.optimizer.state.get(p, {})
            cur[nm] = p.detach().clone()
            for k in ("exp_avg", "exp_avg_sq"):
                if k in st:
                    cur[f"{nm}.{k}"] = st[k].clone()
        bad = torch.zeros(cur["xyz"].shape[0], dtype=torch.bool, device=dev)
        for nm, t in cur.items():
            bad |= ~torch.isfinite(t.reshape(t.shape[0], -1)).all(1)
        if bool(bad.any()) and prev:
            i = int(torch.nonzero(bad)[0])
            print(f"[watch] first non-finite row after iteration {it}: Gaussian {i} ({int(bad.sum())} rows in all)", flush=True)
            for nm in cur:
                if nm.startswith("features_rest"):
                    continue
                print(f"   {nm:22s} before {prev[nm][i].flatten().tolist()}  after {cur[nm][i].flatten().tolist()}", flush=True)
            # which view's gradient does it: the plain (non-pipelined) path at the parameters of one iteration earlier
            from gaussmart_amd.trainer import training_losses
            probe = GaussianModel(3, device=dev)
            probe.create_from_params({k: prev[k] for k in names}, active_sh_degree=int(m.active_sh_degree))
            for ci, c in enumerate(train_cams):
                for q in probe.parameters():
                    q.grad = None
                pkg = render(c, probe, pipe, bg)
                total, _ = training_losses(pkg, c.original_image, opt, it, c, pipe)
                total.backward()
                torch.cuda.synchronize()
                rows = {nm: q.grad[i].flatten().tolist() for nm, q in zip(names, probe.parameters()) if nm != "features_rest" and q.grad is not None}
                nonf = {nm: int((~torch.isfinite(q.grad.reshape(q.shape[0], -1)).all(1)).sum()) for nm, q in zip(names, probe.parameters()) if q.grad is not None}
                print(f"   view {ci}: radius {int(pkg['radii'][i])}  loss {float(total):.5f}  non-finite gradient rows {nonf}  grads of Gaussian {i}: {rows}", flush=True)
            raise SystemExit(3)
        prev.clear()
        prev.update(cur)

    prof = {}

    def on_iteration(it):
        if profile_tail and it == iterations - profile_tail:      # per-kernel averages of the last iterations (HIP events of the
            from gaussmart_amd import _lib                        # library's opt-in profiler: ~5 us of timeline per bracket)
            vp.finish(); torch.cuda.synchronize()
            _lib.profile_reset(); _lib.profile_enable(True)
        if profile_tail and it == iterations:
            from gaussmart_amd import _lib
            vp.finish(); torch.cuda.synchronize()
            _lib.profile_enable(False)
            prof.update({k: [round(ms / n, 4), n] for k, (ms, n) in _lib.profile_read().items() if n})
            if not quiet:
                print("   kernel ms per launch over the last %d iterations: " % profile_tail +
                      ", ".join(f"{k} {v[0]} (x{v[1] / profile_tail:g})" for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0] * kv[1][1])), flush=True)
        if watch_from is not None and it >= watch_from:
            watch(it)
        if it % log_every == 0 or it in marks:
            vp.finish()
            torch.cuda.synchronize()
            now = time.time()
            ok = all_finite(m, report=None if quiet else (lambda msg: print(msg, flush=True)))
            same = vp.replicas_identical()           # (every rank calls it: a collective)
            replicas.append(bool(same))
            loss = float(last_losses["l"]["loss"]) if "l" in last_losses else float("nan")
            trace.append((it, int(m.get_xyz.shape[0]), round(now - t0, 3), loss, int(m.active_sh_degree), ok))
            if not quiet:
                print(f"[it {it:6d}] N {m.get_xyz.shape[0]:8d}  sh {m.active_sh_degree}  "
                      f"{(it - tl[1]) / max(now - tl[0], 1e-9):7.1f} it/s  peak {torch.cuda.max_memory_allocated() / 2**30:5.1f} GiB"
                      f"  finite {ok}" + (f"  replicas identical {same}" if world > 1 else ""), flush=True)
            tl[0], tl[1] = time.time(), it
            if it in marks:
                phase_t[it] = now - t0

    # train() hands the losses of the last iteration back only at the end; keep a reference for the log lines
    import gaussmart_amd.trainer as T
    orig_step = T.training_step

    def step_and_remember(*a, **k):
        pkg, parts = orig_step(*a, **k)
        last_losses["l"] = parts
        return pkg, parts

    T.training_step = step_and_remember
    try:
        train(m, train_cams, opt, pipe, bg, cameras_extent=extent, first_iter=0, iterations=iterations,
              view_parallel=None if dropin else vp, seed=seed, state=state, on_iteration=on_iteration,
              white_background=white_background)
    finally:
        T.training_step = orig_step
    torch.cuda.synchronize()
    total_s = time.time() - t0
    after = (mean_psnr(train_cams), mean_psnr(held_out))
    dens_end = min(opt.densify_until_iter, iterations)
    dens_s = phase_t.get(dens_end, total_s)
    summary = {
        "mode": mode, "preset": preset, "start_points": int(start["xyz"].shape[0]), "target_points": n, "densify_grad_threshold": opt.densify_grad_threshold,
        "width": w, "height": h, "radius_px": radius, "views": views, "lambda_normal": opt.lambda_normal, "lambda_dist": opt.lambda_dist,
        "depth_ratio": pipe.depth_ratio, "white_background": bool(white_background),
        "iterations": iterations, "schedule_iterations": opt.iterations, "cameras_extent": extent, "seed": seed,
        "densification_phase": {"iterations": dens_end, "seconds": round(dens_s, 2),
                                "it_per_s": round(dens_end / dens_s, 1)},
        "rest": ({"iterations": iterations - dens_end, "seconds": round(total_s - dens_s, 2),
                  "it_per_s": round((iterations - dens_end) / max(total_s - dens_s, 1e-9), 1)}
                 if iterations > dens_end else None),
        "whole_run_it_per_s": round(iterations / total_s, 1),
        "psnr_train_before_after_db": [round(before[0], 3), round(after[0], 3)],
        "psnr_held_out_before_after_db": [round(before[1], 3), round(after[1], 3)],
        "final_points": int(m.get_xyz.shape[0]),
        "max_points": max([t[1] for t in trace] + [int(start["xyz"].shape[0])]),
        "pool_buffers_created": R.STATS["pool_buffers_created"], "pool_gib_created": round(R.STATS["pool_bytes_created"] / 2**30, 2),
        "final_sh_degree": int(m.active_sh_degree),
        "peak_hbm_gib": round(torch.cuda.max_memory_allocated() / 2**30, 2),
        "reserved_hbm_gib": round(torch.cuda.max_memory_reserved() / 2**30, 2),
        "all_finite_at_every_log_point": all(t[5] for t in trace),
        "world_size": world, "backend": (torch.distributed.get_backend() if world > 1 else None),
        "replicas_identical_at_every_log_point": all(replicas),
        # sums of the raw bit patterns of the six parameter tensors: two runs of the same command must print the same six numbers
        "kernel_ms_per_launch_tail": prof or None,
        "final_parameter_checksum": [int(v) for v in vp.replica_checksum().tolist()],
        "row_scans_carried": R.STATS["row_scans_carried"],
        "trace_columns": ["iteration", "points", "seconds", "loss", "sh_degree", "finite"],
        "trace": trace,
        "data": "synthetic: target model rendered from jittered views; trained model = perturbed copy, SH degree 0",
    }
    if model_out is not None:
        model_out.append(m)
    return summary


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="scan24", choices=sorted(PRESETS))
    ap.add_argument("--iterations", type=int, default=30000)
    ap.add_argument("--schedule-iterations", type=int, default=None)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--log-every", type=int, default=500)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--extent", type=float, default=5.0, help="cameras_extent handed to densify_and_prune")
    ap.add_argument("--start-fraction", type=float, default=1.0, help="start from every k-th Gaussian of the perturbed scene")
    ap.add_argument("--grad-threshold", type=float, default=None, help="opt.densify_grad_threshold (default: the reference's 0.0002)")
    ap.add_argument("--mode", choices=("fused", "dropin"), default="fused", help="dropin: the reference-shaped loop around the "
                    "reference-signature operator (torch activations / post-processing / L1 + SSIM / Adam)")
    ap.add_argument("--lambda-normal", type=float, default=None)
    ap.add_argument("--lambda-dist", type=float, default=None)
    ap.add_argument("--depth-ratio", type=float, default=None)
    ap.add_argument("--white-background", action="store_true")
    ap.add_argument("--profile-tail", type=int, default=0, help="per-kernel launch averages over the last K iterations")
    ap.add_argument("--watch-from", type=int, default=None, help="diagnostic: from this iteration on, check every iteration and "
                    "print the first Gaussian that turns non-finite, before and after")
    ap.add_argument("--out", default=None)
    a = ap.parse_args(argv)
    s = run(a.preset, a.iterations, a.views, a.log_every, a.seed, a.extent, a.schedule_iterations,
            start_fraction=a.start_fraction, grad_threshold=a.grad_threshold, watch_from=a.watch_from,
            lambda_normal=a.lambda_normal, lambda_dist=a.lambda_dist, depth_ratio=a.depth_ratio, white_background=a.white_background, mode=a.mode, profile_tail=a.profile_tail)
    line = {k: v for k, v in s.items() if k != "trace"}
    if int(os.environ.get("RANK", "0")) == 0:
        print(json.dumps(line))
    if a.out and int(os.environ.get("RANK", "0")) == 0:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(s, f, indent=1)
    assert s["all_finite_at_every_log_point"], "non-finite parameter or moment"
    assert s["replicas_identical_at_every_log_point"], "the replicas of a view-parallel run diverged"
    assert math.isfinite(s["psnr_train_before_after_db"][1])


if __name__ == "__main__":
    main()
