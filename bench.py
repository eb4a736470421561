#!/usr/bin/env python3
"""Headline benchmark: train iters/sec @ 1M Gaussians, 1920x1080 (BASELINE.json `metric`).

One step = one full training iteration of the hot path over one synthetic view:
  lr update -> render() forward (HIP) -> 0.8*L1 + 0.2*(1-SSIM) + 0.05*normal loss -> backward (HIP)
  -> [N>1: RCCL exchange of the per-Gaussian gradients] -> Adam step.
Iteration index is fixed in the 7k..30k regime of the reference schedule (train.py:132-133:
normal loss active, SH degree 3, no densification inside the timed region).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
  python bench.py --preset scan24|bicycle|truck        (the other BASELINE.json configs' shapes; synthetic stand-ins)
  python bench.py --radius-px 12                       (list-length sensitivity of the headline scene)

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the dominant
kernel (timed live with HIP events on the launch stream) and `cpu_baseline` (the pure-PyTorch
oracle on a bounded sample of the same frame, on this box's host cores).
"""
import argparse
import gc
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec

# The other BASELINE.json configurations as synthetic stand-ins (no datasets offline): Gaussian count and frame size of
# the named scene; the mean projected 3-sigma radius is chosen so the instance count D lands in the range SURVEY 8(a) A4
# states for it (scan24: D = 2-4 M at ~10 tiles per Gaussian; bicycle: D = 20-40 M).  "headline" is BASELINE's metric.
PRESETS = {
    "headline": dict(gaussians=1_000_000, width=1920, height=1080, radius_px=6.0,
                     what="synthetic 1M-Gaussian 1920x1080 scene (SURVEY 8(d) recipe)"),
    "scan24": dict(gaussians=300_000, width=1600, height=1200, radius_px=17.0,
                   what="DTU scan24-like: 300k Gaussians at 1600x1200, ~10 tiles per Gaussian"),
    "bicycle": dict(gaussians=5_000_000, width=1237, height=822, radius_px=9.0,
                    what="Mip-NeRF360 bicycle-like: 5M Gaussians at 1237x822, D = 20-40 M"),
    "truck": dict(gaussians=1_000_000, width=979, height=543, radius_px=6.0,
                  what="Tanks&Temples truck-like: 1M Gaussians at 979x543 (identification/camera_loader.py:125), one view per GPU"),
}


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cores():
    """CPU share of this container (cgroup quota / affinity), not the host's core count."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 32))


def algorithmic_bytes(N, D, P):
    """algorithmic HBM bytes per launch of each big kernel."""
    return {
        "preprocess_fwd": N * (232 + 87),
        "render_fwd": D * 76 + P * 60,
        "render_bwd": D * (76 + 72) + P * 60,
        "preprocess_bwd": N * (72 + 232 + 232),
    }


def iteration_bytes(N, D, P, tiles, D_walked=None):
    """Whole-iteration algorithmic bytes.  `D_walked`: charge the compositing kernels only the instances
    they can have touched (the binning sorts all D either way)."""
    npass = math.ceil((32 + math.ceil(math.log2(max(tiles, 2)))) / 8)
    b = algorithmic_bytes(N, D, P)
    if D_walked is not None:
        bw = algorithmic_bytes(N, D_walked, P)
        b["render_fwd"], b["render_bwd"] = bw["render_fwd"], bw["render_bwd"]
    return sum(b.values()) + D * 12 * (1 + 2 * npass) + N * 58 * 28


def _pmc_entry(kernel):
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)["kernels"].get(kernel)
    except (OSError, KeyError, TypeError, ValueError):
        return None


# FETCH_SIZE on gfx950 reports HALF the bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM section) and says
# nothing about other shapes; scripts/microbench/fetch_calib.hip measured this library's own patterns on known byte
# counts (profiles/r02_fetch_calib.json): the factor to multiply FETCH_SIZE by is 2 for 16-B-per-lane streams AND for
# the per-lane 80-byte record gathers / dword gathers of the compositing and binning kernels only where the calibration
# says so.  Kernels are classed by what dominates their reads.
FETCH_FACTOR = {"stream": 2.0, "gather": 2.0}          # overwritten from profiles/r02_fetch_calib.json when present
KERNEL_READ_CLASS = {"render_fwd": "gather", "render_bwd": "gather", "finalize_bins": "gather", "slot_count": "gather",
                     "reduce_rows": "stream", "preprocess_fwd": "stream", "preprocess_bwd": "stream", "adam": "stream"}


def _load_fetch_calibration():
    path = os.path.join(ROOT, "profiles", "r02_fetch_calib.json")
    try:
        with open(path) as f:
            c = json.load(f)["fetch_factor"]
        FETCH_FACTOR.update({k: float(v) for k, v in c.items() if k in FETCH_FACTOR})
    except (OSError, KeyError, TypeError, ValueError):
        pass


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC summary (separate rocprofv3 --pmc passes; counters
    cannot be read inside this process): raw FETCH_SIZE / WRITE_SIZE and the corrected total.  None if not covered."""
    k = _pmc_entry(kernel)
    if not k:
        return None, None
    cls = KERNEL_READ_CLASS.get(kernel, "stream")
    raw = {"fetch_bytes_raw": int(k["fetch_kb"] * 1024), "write_bytes_raw": int(k["write_kb"] * 1024),
           "fetch_factor": FETCH_FACTOR[cls], "read_class": cls}
    return int(FETCH_FACTOR[cls] * k["fetch_kb"] * 1024 + k["write_kb"] * 1024), raw


# Vector-issue accounting of the dominant kernel (round 4).  The bound is MI355X_MICROARCH.md's: a SIMD issues one wave64
# vector instruction per 2 cycles (1,024 SIMDs), at the clock the chip HOLDS while the kernel runs -- measured inside the
# kernels with s_memtime / s_memrealtime (make PROBES=1, scripts/dev_clock_probe.py, profiles/r04_notes/clock_probe_k6_k7.json:
# 2.37 GHz under render_fwd / render_bwd; only an all-FMA loop is power-limited, to 1.86-1.89 GHz, which is what round 3's
# "764 G wave-instructions/s at 2.4 GHz" really was).  Priced kind by kind on this chip (scripts/microbench/valu_rate.hip,
# profiles/r04_notes/valu_rate_kinds.txt: plain 2.4 cycles, DPP 4.3-5.5, SGPR-writing compares 4.4, transcendentals 8.3) the
# instruction mix of render_bwd's loop costs 2.9-3.1 cycles per instruction, i.e. it could reach ~0.68 of the 2-cycle bound.
N_SIMD = 1024
VALU_BOUND_CYCLES = 2.0
VALU_MIX_CYCLES = {"render_bwd": 3.0, "render_fwd": 3.0}     # per instruction, from the per-kind rates x the loop's mix


def measured_clock_mhz(kernel):
    try:
        with open(os.path.join(ROOT, "profiles", "r04_notes", "clock_probe_k6_k7.json")) as f:
            return float(json.load(f)[kernel]["clock_mhz"])
    except (OSError, KeyError, TypeError, ValueError):
        return None


def pmc_valu_insts(kernel):
    k = _pmc_entry(kernel)
    try:
        return float(k["valu_insts"]) if k else None
    except (KeyError, TypeError, ValueError):
        return None


def cpu_baseline(params, cam, n_tiles_sample, n_gauss_sample, dbg, W, H, full=False):
    """Oracle (pure PyTorch fp32 + NumPy binning) on the host cores, every stage of the iteration: preprocess forward +
    backward, binning (duplicate + stable 64-bit sort + ranges), compositing forward + recompute-backward, L1 + SSIM loss
    forward + backward, Adam.  By default the two expensive stages are SAMPLED (preprocess on n_gauss_sample Gaussians,
    compositing on n_tiles_sample evenly spread tiles of the frame's own tile lists) and scaled -- about 20-30 s of host
    work, as the bench contract asks; `full=True` (--cpu-full) runs the whole frame un-sampled (~2 min at 1M / 1080p)."""
    import numpy as np
    from oracle import surfel_ref as O
    from gaussmart_amd.synthetic import activate
    from gaussmart_amd.losses import l1_loss, ssim
    torch.set_num_threads(host_cores())
    cores = torch.get_num_threads()
    a = {k: v.cpu() for k, v in activate(params).items()}
    N = a["means3D"].shape[0]
    S = O.Settings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3), 1.0,
                   cam.world_view_transform.cpu(), cam.full_proj_transform.cpu(), 3, cam.camera_center.cpu())
    # (1) per-Gaussian stage, forward + backward through autograd
    ns = N if full else min(n_gauss_sample, N)
    sub = {k: v[:ns].clone().requires_grad_(True) for k, v in a.items()}
    t0 = time.perf_counter()
    geom = O.preprocess(sub["means3D"], sub["scales"], sub["rotations"], sub["opacities"], sub["shs"], None, None, S)
    (geom.Tm.sum() + geom.xy.sum() + geom.normal.sum() + geom.rgb.sum()).backward()
    t_pre = (time.perf_counter() - t0) * (N / ns)
    # (2) binning of the whole frame (NumPy: duplicate with keys, stable sort of 64-bit keys, tile ranges)
    spl = dbg["splat"].cpu()
    radii = dbg["radii"].cpu().numpy()
    gx, gy = (W + 15) // 16, (H + 15) // 16
    cx, cy, rf = spl[:, 9].numpy(), spl[:, 10].numpy(), radii.astype(np.float32)
    t0 = time.perf_counter()
    with np.errstate(all="ignore"):
        tr = lambda v: np.nan_to_num(np.trunc(v / np.float32(16)), nan=0, posinf=1e9, neginf=-1e9).astype(np.int64)
        rect = np.stack([np.clip(tr(cx - rf), 0, gx), np.clip(tr(cy - rf), 0, gy),
                         np.clip(tr(cx + rf + np.float32(15)), 0, gx), np.clip(tr(cy + rf + np.float32(15)), 0, gy)], 1)
    rect[radii <= 0] = 0
    depth = dbg["depth_key"].cpu().numpy().view(np.float32)
    keys, plist_np = O.bin_tiles(None, radii, rect.astype(np.int32), depth, gx)
    ranges = O.tile_ranges(keys, gx * gy).astype(np.int64)
    t_bin = time.perf_counter() - t0
    # (3) compositing, forward + recompute-backward, on evenly spread tiles (all of them with full=True)
    gT, gxy = spl[:, 0:9].reshape(-1, 3, 3).contiguous(), spl[:, 9:11].contiguous()
    gn, go, gc = spl[:, 11:14].contiguous(), spl[:, 14].contiguous(), spl[:, 15:18].contiguous()
    plist = torch.from_numpy(plist_np.astype(np.int64))
    total_tiles = ranges.shape[0]
    tiles = list(range(total_tiles)) if full else [int(i) for i in np.linspace(0, total_tiles - 1, n_tiles_sample).round()]
    dc, da = torch.ones(3, H, W), torch.ones(7, H, W)
    t0 = time.perf_counter()
    out = O.render_tiles(gT, gxy, gn, go, gc, plist, ranges, S, tiles=tiles)
    O.render_tiles_backward(gT, gxy, gn, go, gc, plist, ranges, out.luse, S, dc, da, tiles=tiles)
    t_tiles = time.perf_counter() - t0
    inst_sample = int(sum(ranges[t, 1] - ranges[t, 0] for t in tiles))
    inst_total = int((ranges[:, 1] - ranges[:, 0]).sum())
    t_render = t_tiles * (inst_total / max(inst_sample, 1))
    # (4) photometric loss forward + backward (utils/loss_utils.py formulation) and (5) Adam over 58 floats per Gaussian
    img = out.color.clone().requires_grad_(True)
    gt = torch.rand(3, H, W)
    t0 = time.perf_counter()
    (0.8 * l1_loss(img, gt) + 0.2 * (1.0 - ssim(img, gt))).backward()
    t_loss = time.perf_counter() - t0
    p = torch.zeros(N, 58, requires_grad=True)
    adam = torch.optim.Adam([p], lr=1e-3, eps=1e-15)
    p.grad = torch.ones_like(p)
    adam.step()
    t0 = time.perf_counter()
    adam.step()
    t_adam = time.perf_counter() - t0
    total = t_pre + t_bin + t_render + t_loss + t_adam
    how = "whole frame, un-sampled" if full else \
        (f"preprocess on {ns} of {N} Gaussians and compositing on {len(tiles)} of {total_tiles} tiles holding {inst_sample} of "
         f"{inst_total} instances, both scaled to the frame; binning, loss and Adam on the whole frame")
    return {"value": 1.0 / total, "unit": "iters/s", "cores": cores, "kind": "port",
            "seconds_per_iteration": round(total, 2),
            "stage_seconds": {"preprocess_fwd_bwd": round(t_pre, 2), "binning": round(t_bin, 2),
                              "composite_fwd_bwd": round(t_render, 2), "loss_fwd_bwd": round(t_loss, 2), "adam": round(t_adam, 2)},
            "sample": f"oracle/surfel_ref.py fp32 + NumPy stable sort, one full training iteration ({how}); measured host time "
                      f"{t_pre * ns / N + t_bin + t_tiles + t_loss + t_adam:.1f} s"}


def wide_payload_bench(args, preset, N, W, H, radius_px):
    """BASELINE.json config 5 ("extra per-Gaussian feature channels rendered / backpropped"; build-defined: SURVEY section 0
    fact 5): the operator with colors_precomp [N,C] -- forward + backward through the C-ABI, per-kernel times from the
    library's event profiler, peak HBM.  One "iteration" = one forward + one backward of the rasterizer (no loss kernels,
    no optimiser: the reference has no C-channel training loop to time).  One GPU."""
    import torch as _t
    from gaussmart_amd import _lib
    from gaussmart_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    from gaussmart_amd.synthetic import activate, make_scene
    C = args.channels
    dev = _t.device("cuda:0")
    _lib.lib()
    log(f"wide payload: {N} Gaussians, {W}x{H}, radius {radius_px} px, {C} channels")
    params, cam = make_scene(N, W, H, seed=0, device="cpu", radius_px=radius_px)
    a = {k: v.to(dev) for k, v in activate(params).items() if k in ("means3D", "opacities", "scales", "rotations")}
    g = _t.Generator().manual_seed(0)
    col = _t.rand(N, C, generator=g).to(dev).requires_grad_(True)
    ins = {k: a[k].clone().requires_grad_(True) for k in a}
    rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), _t.zeros(C, device=dev), 1.0,
                                       cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), 3,
                                       cam.camera_center.to(dev), False, False)
    rast = GaussianRasterizer(rs)
    wc, wa = _t.randn(C, H, W, device=dev), _t.randn(7, H, W, device=dev)
    m2d = _t.zeros(N, 3, device=dev, requires_grad=True)

    def step():
        c, r, am = rast(means3D=ins["means3D"], means2D=m2d, colors_precomp=col, opacities=ins["opacities"],
                        scales=ins["scales"], rotations=ins["rotations"])
        ((c * wc).sum() + (am * wa).sum()).backward()
        for t in (col, m2d, *ins.values()):
            t.grad = None
    for _ in range(max(args.warmup, 2)):
        step()
    _t.cuda.synchronize()
    _t.cuda.reset_peak_memory_stats(dev)
    _lib.profile_reset()
    _lib.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    _t.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    _lib.profile_enable(False)
    prof = {k: round(ms / max(n, 1), 4) for k, (ms, n) in _lib.profile_read().items() if n}
    out = {"metric": f"rasterizer fwd+bwd iters/sec @{N} Gaussians {W}x{H}, {C}-channel payload", "value": args.steps / elapsed,
           "unit": "iters/s", "n_gpus": 1, "steps": args.steps, "warmup": max(args.warmup, 2), "ms_per_step": elapsed / args.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"{preset['what']}; colors_precomp [{N},{C}] through GaussianRasterizer forward + backward "
                                  f"(event-timed per kernel, so the wall figure carries ~5 us per bracket)",
                      "preset": args.preset, "gaussians": N, "width": W, "height": H, "radius_px": radius_px, "channels": C},
           "kernel_ms": prof,
           "hbm_peak_gb": {"allocated": round(_t.cuda.max_memory_allocated(dev) / 2**30, 2),
                           "reserved": round(_t.cuda.max_memory_reserved(dev) / 2**30, 2)}}
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)    # >= 200 timed iterations after 20 warm-up
    ap.add_argument("--warmup", type=int, default=20)    # (0.6 s of GPU time at 1M / 1080p)
    ap.add_argument("--preset", choices=sorted(PRESETS), default="headline")
    ap.add_argument("--gaussians", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--radius-px", type=float, default=None, help="mean projected 3-sigma radius of the synthetic splats")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-tiles", type=int, default=1200)   # ~15 % of the frame: ~15 s of host work
    ap.add_argument("--cpu-gaussians", type=int, default=250_000)
    ap.add_argument("--cpu-full", action="store_true", help="CPU baseline on the whole frame, un-sampled (~2 min)")
    ap.add_argument("--forward-frames", type=int, default=50, help="inference frames (no_grad render) timed after the run")
    ap.add_argument("--dropin-render", choices=("torch", "hip"), default="torch",
                    help="--mode dropin: `torch` = render() as the reference writes it (torch activations, torch post-processing of "
                         "allmap); `hip` = gaussmart_amd.gaussian_renderer.render with the activations fused into the operator and "
                         "the derived maps from fused_surface_maps (same signature, same dictionary)")
    ap.add_argument("--dropin-adam", choices=("torch", "hip"), default="torch",
                    help="--mode dropin: `torch` = torch.optim.Adam as scene/gaussian_model.py:295 builds it; `hip` = "
                         "gaussmart_amd.fused_adam.FusedAdam (same class interface and state, one launch per step)")
    ap.add_argument("--dropin-loss", choices=("torch", "hip"), default="torch",
                    help="--mode dropin: `torch` = the reference's utils/loss_utils.py formulation (five grouped convolutions through "
                         "MIOpen); `hip` = gaussmart_amd.loss_utils, the same two functions on the fused kernel")
    ap.add_argument("--mode", choices=("fused", "dropin"), default="fused",
                    help="fused (default, the headline): the build's own trainer -- raw-parameter operator, fused objective, "
                         "factored SH Adam.  dropin: what INTEGRATION.md section 1 delivers under the reference's own loop -- "
                         "the reference-signature operator from diff_surfel_rasterization with torch activations "
                         "(gaussian_renderer/__init__.py:55-106), the torch post-processing of render(), torch L1 + SSIM "
                         "(utils/loss_utils.py), torch.optim.Adam and one .item() per step (train.py:112-152,214)")
    ap.add_argument("--eval-flags", action="store_true",
                    help="the flags the reference's evaluation scripts train with (scripts/dtu_eval.py:45: "
                         "--lambda_normal 0 --lambda_dist 0): no gradient reaches the surface channels of allmap")
    ap.add_argument("--channels", type=int, default=3,
                    help="wide per-pixel payload (BASELINE.json config 5): time the operator's forward + backward with "
                         "colors_precomp [N,C], C = 4..64 (multiple of 4), on the chosen preset instead of the training step")
    args = ap.parse_args()
    preset = PRESETS[args.preset]
    N = args.gaussians or preset["gaussians"]
    W, H = args.width or preset["width"], args.height or preset["height"]
    radius_px = args.radius_px if args.radius_px is not None else preset["radius_px"]
    headline = (N, W, H, radius_px) == (1_000_000, 1920, 1080, 6.0)
    if args.channels != 3:
        return wide_payload_bench(args, preset, N, W, H, radius_px)

    # Rank 0 must print exactly ONE line on stdout, but RCCL writes a version banner to fd 1 when the first
    # communicator is created: park the real stdout and send everything else (library chatter included) to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    # the host driver of these boxes only supports dmabuf IPC; RCCL fails with hipIpcGetMemHandle otherwise
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    if os.environ.get("GSR_BENCH_SINGLE_DEVICE"):      # rehearsal aid: every rank on cuda:0 (if the RCCL build allows it)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    under_launcher = "RANK" in os.environ and "MASTER_PORT" in os.environ
    if world > 1 or under_launcher:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("GSR_BENCH_BACKEND", "nccl")      # "nccl" is RCCL on ROCm; "gloo" only to rehearse the
        if backend == "nccl":                                      # N > 1 code path with several ranks on ONE GPU
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from gaussmart_amd import _lib
    from gaussmart_amd.synthetic import make_scene, perturb, activate, jittered_cameras
    from gaussmart_amd.gaussian_model import GaussianModel
    from gaussmart_amd.gaussian_renderer import render
    from gaussmart_amd.params import OptimizationParams, PipelineParams
    from gaussmart_amd.trainer import training_step
    from gaussmart_amd.view_parallel import ViewParallel
    from gaussmart_amd.rasterizer import rasterize_debug, GaussianRasterizationSettings

    _lib.lib()   # fail loudly if the HIP extension is missing
    _load_fetch_calibration()
    log(f"rank {rank}/{world} on {torch.cuda.get_device_name(dev)}; building scene ({args.preset}: {N} Gaussians, {W}x{H}, "
        f"radius {radius_px} px)")
    params, _ = make_scene(N, W, H, seed=0, device="cpu", radius_px=radius_px)
    cam = jittered_cameras(world, W, H, seed=0, device=dev)[rank]   # one view per rank
    bg = torch.zeros(3, device=dev)
    pipe, opt = PipelineParams(), OptimizationParams()
    if os.environ.get("GSR_BENCH_PLAIN_ACTIVATIONS"):      # A/B aid: torch activations + reference-signature operator
        pipe.fused_activations = False
    if os.environ.get("GSR_BENCH_UNFACTORED"):             # A/B aid: explicit SH gradient tensors (58 floats per Gaussian)
        pipe.factored_sh_grad = False
    if args.eval_flags:                                    # scripts/dtu_eval.py:45
        opt.lambda_normal, opt.lambda_dist = 0.0, 0.0
    dropin = args.mode == "dropin"
    if dropin:
        if world > 1:
            raise SystemExit("--mode dropin is the reference's single-GPU loop (it has no distributed code)")
        pipe.fused_activations = False
        pipe.factored_sh_grad = False
        if args.dropin_render == "hip":
            # a third optional line: `from gaussmart_amd.gaussian_renderer import render` -- the same
            # signature and dictionary, the activations inside the operator's kernels and the five derived maps from one launch
            pipe.fused_activations = True
            pipe.fused_surface_maps = True

    target = GaussianModel(3, device=dev)
    target.create_from_params(perturb(params))
    with torch.no_grad():
        gt = render(cam, target, pipe, bg)["render"].clamp(0, 1).contiguous()
    del target
    model = GaussianModel(3, device=dev)
    model.create_from_params(params)
    if dropin:
        # torch.optim.Adam over the six parameter groups (scene/gaussian_model.py:282-295); --dropin-adam hip: the second optional
        # one-line swap of INTEGRATION.md section 1 (fused_adam.FusedAdam: a torch.optim.Adam subclass, same state layout)
        model.use_fused_adam = args.dropin_adam == "hip"
    model.training_setup(opt)
    force_dp = bool(os.environ.get("GSR_BENCH_FORCE_DP")) and dist.is_initialized()   # rehearsal on one GPU
    vp = ViewParallel(model, force=force_dp) if (world > 1 or force_dp) else None
    if dropin:
        vp = None
    elif vp is None and os.environ.get("GSR_BENCH_LOCAL_OVERLAP", "1") != "0":
        # N = 1: the same step pipeline as N > 1, minus the exchange -- the HBM-bound SH update (and the next forward's SH
        # colour pass behind it) run on a side stream beside the next forward's latency-bound depth sort / binning.  Same
        # kernels, same arithmetic, bit-identical parameters (tests/test_gpu_view_parallel.py); everything is joined by
        # the synchronisation that ends the timed region.  GSR_BENCH_LOCAL_OVERLAP=0: strictly serial step
        # (same box, 200 steps, three alternating runs each: 470.4 / 474.9 / 477.3 vs 482.9 / 483.2 / 478.8 it/s).
        vp = ViewParallel(model, overlap_local=True)

    # how many ranks the collective backend itself sees (an all-reduce of ones), for the top level of the JSON line
    ranks_counted = None
    if dist.is_initialized():
        ones = torch.ones(1, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(ones)
        ranks_counted = int(ones.item())

    base_iter = 10_000
    # the view of the next step is known (this rank always renders its own view): the SH optimiser step leaves its colours
    # (gsr_adam_sh_factored_next), the next forward skips the SH colour pass.  GSR_BENCH_COLOR_CACHE=0: off (A/B aid)
    next_cam = cam if os.environ.get("GSR_BENCH_COLOR_CACHE", "1") != "0" else None
    def step(i):
        training_step(model, cam, gt, opt, pipe, bg, base_iter + i, view_parallel=vp, next_cam=next_cam)

    if dropin:
        # One iteration of the reference's loop (train.py:93-216) around the drop-in operator: render() with torch
        # activations and the torch post-processing (expected depth, depth_to_normal, ...), torch L1 + SSIM, the two
        # regularizers, backward, one .item() (the reference reads four for its progress bar), torch Adam.
        if args.dropin_loss == "hip":       # the one-line swap of INTEGRATION.md section 1: same names, the fused kernel underneath
            from gaussmart_amd.loss_utils import l1_loss, ssim
        else:
            from gaussmart_amd.losses import l1_loss, ssim
        ema = [0.0]

        def step(i):        # noqa: F811
            it = base_iter + i
            model.update_learning_rate(it)
            pkg = render(cam, model, pipe, bg)
            image = pkg["render"]
            Ll1 = l1_loss(image, gt)
            loss = (1.0 - opt.lambda_dssim) * Ll1 + opt.lambda_dssim * (1.0 - ssim(image, gt))
            lambda_normal = opt.lambda_normal if it > 7000 else 0.0
            lambda_dist = opt.lambda_dist if it > 3000 else 0.0
            normal_error = (1 - (pkg["rend_normal"] * pkg["surf_normal"]).sum(dim=0))[None]
            total = loss + lambda_dist * pkg["rend_dist"].mean() + lambda_normal * normal_error.mean()
            total.backward()
            ema[0] = 0.4 * loss.item() + 0.6 * ema[0]
            model.optimizer.step()
            model.optimizer.zero_grad(set_to_none=True)

    # Python's cyclic garbage collector runs when allocation counts cross a threshold, i.e. at arbitrary steps, and a full
    # collection stalls the host for a millisecond or more -- in a 20-step timed region that is several per cent of noise
    # that has nothing to do with the device.  Collect now, keep it off while stepping (reference counting still frees
    # everything the steps allocate), turn it back on after the measurements.
    gc.collect()
    gc.disable()
    log("target rendered; warm-up")
    if (vp is not None and os.environ.get("GSR_BENCH_HIGH_PRIORITY", "1") != "0") or os.environ.get("GSR_BENCH_HIGH_PRIORITY") == "1":
        # view-parallel step: the step itself runs on a HIGH-priority stream, so its short latency-bound kernels (sort
        # passes, scans, emission) are dispatched ahead of the bandwidth-bound SH update that shares the GPU with them
        # from the normal-priority side stream (size-1 rehearsal, same box: 446 -> 457 it/s)
        hp = torch.cuda.Stream(device=dev, priority=-1)
        hp.wait_stream(torch.cuda.current_stream(dev))      # scene, target and model were set up on the default stream
        torch.cuda.set_stream(hp)
    # HIP events around a kernel cost ~5 us of GPU timeline each, so the timed region brackets ONLY the dominant
    # kernel; which one that is is measured here, during the (untimed) warm-up, with all four big kernels bracketed.
    big = ("preprocess_fwd", "render_fwd", "render_bwd", "preprocess_bwd")
    _lib.profile_reset()
    _lib.profile_enable(big)
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    warm = {k: (ms / n if n else 0.0) for k, (ms, n) in _lib.profile_read().items() if k in big}
    dom_warm = max(warm, key=warm.get) if args.warmup > 0 and any(warm.values()) else "render_bwd"

    # Replica equality of the view-parallel step (every rank must hold bit-identical parameters).  The default N > 1 step
    # is the pipelined one (collectives + SH update on side streams); should it ever leave the replicas different, the
    # run falls back to the blocking exchange (the path the two-rank tests cover), re-synchronised from rank 0, and says so.
    dp_path, replicas_ok_warm = None, None
    if vp is not None and (world > 1 or force_dp):
        dp_path = "pipelined" if vp.pipelined and dist.get_backend() == "nccl" else "blocking"
        replicas_ok_warm = vp.replicas_identical()
        if not replicas_ok_warm and dp_path == "pipelined":
            log("replicas differ after the warm-up with the pipelined exchange: falling back to the blocking exchange")
            vp.pipelined = False
            vp.resync_from_rank0(model.optimizer)
            dp_path = "blocking (fallback: the pipelined step left the replicas different)"
            for i in range(max(2, args.warmup // 4)):
                step(i)
            torch.cuda.synchronize()

    log(f"timing (dominant kernel in warm-up: {dom_warm})")
    timed = _lib.KERNEL_NAMES if os.environ.get("GSR_BENCH_PROFILE_ALL") else (dom_warm,)   # all: adds event overhead
    _lib.profile_reset()
    _lib.profile_enable(timed)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    local_elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    _lib.profile_enable(False)
    prof = _lib.profile_read()
    log(f"timed {args.steps} steps in {elapsed:.3f}s")
    per_rank_ms = [local_elapsed / args.steps * 1e3]
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor([local_elapsed / args.steps * 1e3], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank_ms = [float(x.item()) for x in allr]

    # ---- after the timed region (so none of this perturbs `value`) -------------------------------------------------
    # (a) per-step times with device events: median step, and -- view-parallel runs -- how long the step's own stream
    #     stalls on collectives (exposed communication), per step
    # Per-step device events would be the obvious tool, but a timing event on the step's stream is not free on this
    # stack: at D = 18 M (exact-row path: one host wait per backward) steps bracketed by events took 4.3-11.8 ms against a
    # 3.06 ms mean of the un-instrumented region.  So the distribution comes from CHUNKS: 10 chunks of n_post / 10 steps,
    # host clock, one synchronisation per chunk boundary (<= 1 % perturbation); the median chunk gives the median step.
    n_post = min(max(args.steps, 10), 100) // 10 * 10
    if vp is not None:
        vp.probe = True
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    chunk = n_post // 10
    step_ms = []
    for c in range(10):
        tc = time.perf_counter()
        for i in range(chunk):
            step(args.warmup + args.steps + c * chunk + i)
        torch.cuda.synchronize()
        step_ms.append((time.perf_counter() - tc) / chunk * 1e3)
    step_ms_chrono = list(step_ms)
    step_ms.sort()
    median_ms = 0.5 * (step_ms[4] + step_ms[5])
    comm_exposed_ms = comm_waits = None
    if vp is not None:
        exp_ms, n_waits = vp.exposed_ms()
        comm_exposed_ms, comm_waits = exp_ms / n_post, n_waits / n_post
        vp.probe = False
    replicas_ok = vp.replicas_identical() if (vp is not None and (world > 1 or force_dp)) else None
    # (a') the same step with the forward building ALL seven allmap channels.  By default the fused trainer does not accumulate
    # the channels its objective cannot read in this configuration -- distortion and median depth at lambda_dist = 0 and
    # depth_ratio = 0, the reference's defaults -- which changes no loss value, gradient or parameter by a single bit
    # (tests/test_gpu_rasterizer.py::test_forward_without_distortion_and_median_equals_the_general_one_elsewhere); the line
    # carries both rates so that nobody has to take that on trust.
    full_maps = None
    if not dropin and getattr(pipe, "color_only_when_unregularized", False):
        pipe.color_only_when_unregularized = False
        for i in range(5):
            step(args.warmup + args.steps + n_post + i)
        torch.cuda.synchronize()
        tf0 = time.perf_counter()
        for i in range(n_post):
            step(args.warmup + args.steps + n_post + 5 + i)
        torch.cuda.synchronize()
        full_maps = {"ms_per_step": (time.perf_counter() - tf0) / n_post * 1e3, "steps": n_post}
        full_maps["value"] = world * 1e3 / full_maps["ms_per_step"]
        pipe.color_only_when_unregularized = True

    # (b) the reference's own `iter_time` bracket (train.py:91,145: forward + loss + backward, no optimiser step) and
    # (c) inference frames the way render.py / view.py produce them (render() under no_grad: forward-only kernels)
    gc.enable()
    ref_iter_ms = fwd_fps = None
    if rank == 0:
        if vp is not None:
            vp.finish()
        n_ref = min(20, max(args.steps, 1))
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ref)]
        for a, b in ev:
            a.record()
            training_step(model, cam, gt, opt, pipe, bg, base_iter + args.warmup + args.steps, step_optimizer=False)
            b.record()
            model.optimizer.zero_grad(set_to_none=True)
            if hasattr(model.optimizer, "take_pending_sh"):
                model.optimizer.take_pending_sh()
        torch.cuda.synchronize()
        ref_iter_ms = sorted(a.elapsed_time(b) for a, b in ev)[n_ref // 2]
        if args.forward_frames > 0:
            with torch.no_grad():
                for _ in range(3):
                    render(cam, model, pipe, bg, surface_maps=False)
                torch.cuda.synchronize()
                tf = time.perf_counter()
                for _ in range(args.forward_frames):
                    render(cam, model, pipe, bg, surface_maps=False)
                torch.cuda.synchronize()
                fwd_fps = args.forward_frames / (time.perf_counter() - tf)
    if world > 1:
        dist.barrier()

    if rank == 0:
        # measured instance count of this frame (plugs into the algorithmic byte model)
        a = activate({k: p.detach() for k, p in zip(("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation"),
                                                    (model._xyz, model._features_dc, model._features_rest, model._opacity,
                                                     model._scaling, model._rotation))})
        rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), bg, 1.0,
                                           cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
        dbg = rasterize_debug(a["means3D"], a["opacities"], a["shs"], None, a["scales"], a["rotations"], None, raster_settings=rs)
        D, P = dbg["num_rendered"], W * H
        tiles = ((W + 15) // 16) * ((H + 15) // 16)
        nc = dbg["n_contrib"][0].float()
        # what K6 / K7 actually walk: per tile the longest prefix of its list any quad staged before its pixels saturated
        # (the contractual byte model charges all D instances; on occluded scenes most of them are never touched)
        D_walked = int(dbg["covered"].long().max(dim=1).values.sum()) if "covered" in dbg and dbg["covered"].numel() else D
        per_kernel = {k: (ms / n if n else 0.0) for k, (ms, n) in prof.items() if k in big and n}
        dom = max(per_kernel, key=per_kernel.get)
        ab = algorithmic_bytes(N, D, P)
        achieved = ab[dom] / (per_kernel[dom] * 1e-3) / 1e9 if per_kernel[dom] > 0 else 0.0
        ab_walked = algorithmic_bytes(N, D_walked, P)
        achieved_walked = ab_walked[dom] / (per_kernel[dom] * 1e-3) / 1e9 if per_kernel[dom] > 0 else 0.0
        ms_per_step = elapsed / args.steps * 1e3
        iter_b = iteration_bytes(N, D, P, tiles)
        traffic, traffic_raw = pmc_traffic(dom) if headline else (None, None)    # the committed counters are this workload's
        metric = "train iters/sec @1M Gaussians 1080p" if (N, W, H) == (1_000_000, 1920, 1080) else \
            f"train iters/sec @{N} Gaussians {W}x{H}"
        out = {
            "metric": metric, "value": world * args.steps / elapsed, "unit": "iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{preset['what'] if args.preset != 'headline' or headline else 'synthetic scene'}; "
                                   f"{N} Gaussians, {W}x{H}, mean projected radius {radius_px} px, seed 0, SH degree 3, "
                                   f"L1+SSIM+normal loss, Adam; one view per GPU per step",
                       "preset": args.preset, "gaussians": N, "width": W, "height": H, "radius_px": radius_px, "instances_D": D,
                       "tile_list_mean": round(D / tiles, 1), "entries_walked_per_pixel_mean": round(float(nc.mean()), 1),
                       "entries_walked_per_pixel_max": int(nc.max()),
                       "parallelism": f"view-parallel dp{world}" if world > 1 else "single GPU",
                       "mode": args.mode + (" (reference-signature operator under a reference-shaped loop: " +
                                            ("activations in the operator, derived maps from one HIP launch, " if args.dropin_render == "hip"
                                             else "torch activations, torch post-processing, ") + ("gaussmart_amd.loss_utils l1_loss / ssim (HIP)" if args.dropin_loss == "hip"
                                                                        else "torch L1 + SSIM") +
                                            ", " + ("fused_adam.FusedAdam" if args.dropin_adam == "hip" else "torch.optim.Adam") + ", one .item() per step)"
                                            if dropin else " (raw-parameter operator, fused objective, factored SH Adam)"),
                       "loss": "L1 + SSIM only (--eval-flags: lambda_normal 0, lambda_dist 0, scripts/dtu_eval.py:45)"
                               if args.eval_flags else "L1 + SSIM + normal consistency (lambda_normal 0.05, lambda_dist 0)",
                       "step_pipeline": ("SH Adam update + next colour pass on a side stream beside the next forward's binning; "
                                         "every step renders the SAME view, so the optimiser step always leaves the next "
                                         "forward's SH colours (colour cache hit on every step); the forward accumulates only the "
                                         "allmap channels this objective reads (no distortion / median depth at lambda_dist = 0, "
                                         "depth_ratio = 0; none at all with --eval-flags)" if next_cam is not None else
                                         "SH Adam update on a side stream beside the next forward's binning")
                                        if (vp is not None and (vp.overlap_local or world > 1 or force_dp)) else "serial"},
            "ms_per_step_median": median_ms,
            # the step with every allmap channel accumulated in the forward (see (a') above); `value` is the default step
            "with_all_seven_allmap_channels": full_maps,
            "ms_per_step_chunks": [round(x, 4) for x in step_ms_chrono],      # 10 chunks of consecutive steps, in order
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_raw": traffic_raw,
                         "traffic_source": "profiles/pmc_traffic.json (separate rocprofv3 --pmc passes, same scene): "
                                           "fetch_factor x FETCH_SIZE + WRITE_SIZE, factor per read class calibrated on known "
                                           "byte counts (profiles/r02_fetch_calib.json, scripts/microbench/fetch_calib.hip)",
                         "algorithmic_bytes_per_launch": ab[dom], "avg_launch_ms": per_kernel[dom],
                         # the same fraction with D_walked = sum over tiles of the longest staged prefix instead of D: the
                         # bytes the compositing kernels can have moved (equal to `frac` unless lists are occluded)
                         "walked": {"instances_walked": D_walked, "algorithmic_bytes_per_launch": ab_walked[dom],
                                    "achieved": achieved_walked, "frac": achieved_walked / HBM_PEAK_GBS}},
            # the dominant kernels are VALU-bound: the same launch priced against the MEASURED
            # vector-issue ceiling of the chip instead of the HBM roofline (informational, not the contract's roofline)
            "valu_issue": (lambda n, clk: None if not n or not clk or per_kernel[dom] <= 0 else {
                "kernel": dom, "valu_wave_insts_per_launch": n, "achieved_ginst_s": n / (per_kernel[dom] * 1e-3) / 1e9,
                "clock_mhz": clk,
                "cycles_per_instr_per_simd": N_SIMD * clk * 1e6 * per_kernel[dom] * 1e-3 / n,
                "bound_cycles_per_instr": VALU_BOUND_CYCLES,
                "frac": VALU_BOUND_CYCLES / (N_SIMD * clk * 1e6 * per_kernel[dom] * 1e-3 / n),
                "mix_cycles_per_instr": VALU_MIX_CYCLES.get(dom),
                "frac_of_what_the_mix_allows": (VALU_MIX_CYCLES[dom] / (N_SIMD * clk * 1e6 * per_kernel[dom] * 1e-3 / n))
                if dom in VALU_MIX_CYCLES else None,
                "source": "instructions: profiles/pmc_traffic.json (SQ_INSTS_VALU, calibrated 1.0001x on known counts); clock: "
                          "in-kernel s_memtime / s_memrealtime (profiles/r04_notes/clock_probe_k6_k7.json); bound: 2 cycles per "
                          "wave64 instruction per SIMD (MI355X_MICROARCH.md); mix: per-kind rates of "
                          "scripts/microbench/valu_rate.hip x the loop's instruction mix (DESIGN.md section 5)"})(
                    pmc_valu_insts(dom) if headline else None, measured_clock_mhz(dom)),
            "hbm_peak_gb": {"allocated": round(torch.cuda.max_memory_allocated(dev) / 2**30, 2),
                            "reserved": round(torch.cuda.max_memory_reserved(dev) / 2**30, 2)},
            "reference_iter_time_ms": ref_iter_ms,    # median of the reference's fwd+loss+bwd bracket (no Adam)
            "forward_only_fps": fwd_fps,              # render() under no_grad (render.py / view.py), this frame
            "kernel_ms": {k: round(v, 4) for k, v in per_kernel.items()},
            "kernel_ms_warmup": {k: round(v, 4) for k, v in warm.items()},
            "kernel_ms_per_step": {k: round(ms / args.steps, 4) for k, (ms, n) in prof.items() if n},
            "iteration": {"algorithmic_bytes": iter_b, "hbm_frac": iter_b / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          "hbm_frac_walked": iteration_bytes(N, D, P, tiles, D_walked) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
        # top level, so that "did RCCL see N ranks, and which step ran" needs no digging in a scaling record
        out["comm_backend"] = dist.get_backend() if dist.is_initialized() else None
        out["comm_ranks_counted"] = ranks_counted          # all-reduce of ones over that backend (None: single process)
        out["view_parallel_path"] = dp_path                # "pipelined" / "blocking" / fallback note (None: single GPU)
        if vp is not None and (world > 1 or force_dp):
            out["view_parallel"] = {"path": dp_path, "comm_exposed_ms_per_step": comm_exposed_ms,
                                    "collective_waits_per_step": comm_waits,
                                    "per_rank_ms_per_step": [round(x, 4) for x in per_rank_ms],
                                    "replicas_identical_after_warmup": replicas_ok_warm,
                                    "replicas_identical_after_run": replicas_ok,
                                    "exchange": "40 B/Gaussian all-reduced (geometry) + 12 B/Gaussian/peer all-gathered "
                                                "(factored SH colour gradients)" if pipe.factored_sh_grad else
                                                "232 B/Gaussian all-reduced"}
        if world == 1 and not args.no_cpu_baseline:
            log(f"GPU part done ({out['value']:.2f} it/s, D={D}); timing the CPU oracle on {host_cores()} cores")
            out["cpu_baseline"] = cpu_baseline(params, cam, args.cpu_tiles, args.cpu_gaussians, dbg, W, H, full=args.cpu_full)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
