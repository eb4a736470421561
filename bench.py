#!/usr/bin/env python3
"""Headline benchmark: train iters/sec @ 1M Gaussians, 1920x1080 (BASELINE.json `metric`).

One step = one full training iteration of the hot path over one synthetic view:
  lr update -> render() forward (HIP) -> 0.8*L1 + 0.2*(1-SSIM) + 0.05*normal loss -> backward (HIP)
  -> [N>1: RCCL all-reduce of the per-Gaussian gradients] -> Adam step.
Iteration index is fixed in the 7k..30k regime of the reference schedule (train.py:132-133:
normal loss active, SH degree 3, no densification inside the timed region).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the dominant
kernel (timed live with HIP events on the launch stream) and `cpu_baseline` (the pure-PyTorch
oracle on a bounded sample of the same frame, on this box's host cores).
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


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


def iteration_bytes(N, D, P, tiles):
    npass = math.ceil((32 + math.ceil(math.log2(max(tiles, 2)))) / 8)
    b = algorithmic_bytes(N, D, P)
    return sum(b.values()) + D * 12 * (1 + 2 * npass) + N * 58 * 28


def pmc_traffic_bytes(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC summary (collected in separate rocprofv3 --pmc
    passes; counters cannot be read inside this process).  gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE
    counts half of the bytes of wide coalesced reads.  None when the summary does not cover the kernel."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            k = json.load(f)["kernels"].get(kernel)
        return int((2.0 * k["fetch_kb"] + k["write_kb"]) * 1024) if k else None
    except (OSError, KeyError, TypeError, ValueError):
        return None


VALU_CEILING_GINST_S = 922.0   # measured: scripts/microbench/valu_rate.hip, independent wave64 v_fma_f32, whole chip


def pmc_valu_insts(kernel):
    """SQ_INSTS_VALU wave-instructions per launch of `kernel` from the committed PMC summary (None if not covered)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return float(json.load(f)["kernels"][kernel]["valu_insts"])
    except (OSError, KeyError, TypeError, ValueError):
        return None


def cpu_baseline(params, cam, n_tiles_sample, n_gauss_sample, dbg, W, H):
    """Oracle (pure PyTorch, fp32) on the host cores: preprocess on a sample of the Gaussians,
    forward+backward compositing on a sample of the frame's tiles (the frame's own tile lists),
    both scaled to the whole frame."""
    import numpy as np
    from oracle import surfel_ref as O
    from gaussmart_amd.synthetic import activate
    torch.set_num_threads(host_cores())
    cores = torch.get_num_threads()
    a = {k: v.cpu() for k, v in activate(params).items()}
    N = a["means3D"].shape[0]
    S = O.Settings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3), 1.0,
                   cam.world_view_transform.cpu(), cam.full_proj_transform.cpu(), 3, cam.camera_center.cpu())
    # (1) per-Gaussian stage, forward + backward through autograd
    ns = min(n_gauss_sample, N)
    sub = {k: v[:ns].clone().requires_grad_(True) for k, v in a.items()}
    t0 = time.perf_counter()
    geom = O.preprocess(sub["means3D"], sub["scales"], sub["rotations"], sub["opacities"], sub["shs"], None, None, S)
    (geom.Tm.sum() + geom.xy.sum() + geom.normal.sum() + geom.rgb.sum()).backward()
    t_pre = (time.perf_counter() - t0) * (N / ns)
    # (2) compositing, forward + recompute-backward, on evenly spread tiles of the real frame
    spl = dbg["splat"].cpu()
    gT, gxy = spl[:, 0:9].reshape(-1, 3, 3).contiguous(), spl[:, 9:11].contiguous()
    gn, go, gc = spl[:, 11:14].contiguous(), spl[:, 14].contiguous(), spl[:, 15:18].contiguous()
    plist = dbg["point_list"].cpu().to(torch.int64)
    ranges = dbg["ranges"].cpu().numpy().astype(np.int64)
    total_tiles = ranges.shape[0]
    tiles = [int(i) for i in np.linspace(0, total_tiles - 1, n_tiles_sample).round()]
    dc, da = torch.ones(3, H, W), torch.ones(7, H, W)
    t0 = time.perf_counter()
    out = O.render_tiles(gT, gxy, gn, go, gc, plist, ranges, S, tiles=tiles)
    O.render_tiles_backward(gT, gxy, gn, go, gc, plist, ranges, out.luse, S, dc, da, tiles=tiles)
    t_tiles = time.perf_counter() - t0
    inst_sample = int(sum(ranges[t, 1] - ranges[t, 0] for t in tiles))
    inst_total = int((ranges[:, 1] - ranges[:, 0]).sum())
    t_render = t_tiles * (inst_total / max(inst_sample, 1))
    return {"value": 1.0 / (t_pre + t_render), "unit": "iters/s", "cores": cores, "kind": "port",
            "sample": f"oracle/surfel_ref.py fp32: preprocess fwd+bwd on {ns} of {N} Gaussians ({t_pre:.1f}s scaled) + "
                      f"composite fwd+bwd on {len(tiles)} of {total_tiles} tiles holding {inst_sample} of {inst_total} "
                      f"instances ({t_tiles:.1f}s measured, scaled by instances); binning, loss and Adam not included"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)    # >= 200 timed iterations after 20 warm-up
    ap.add_argument("--warmup", type=int, default=20)    # (0.6 s of GPU time at 1M / 1080p)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-tiles", type=int, default=1600)   # ~20 % of the frame: 15-20 s of host work
    args = ap.parse_args()

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
    log(f"rank {rank}/{world} on {torch.cuda.get_device_name(dev)}; building scene")
    N, W, H = args.gaussians, args.width, args.height
    params, _ = make_scene(N, W, H, seed=0, device="cpu")
    cam = jittered_cameras(world, W, H, seed=0, device=dev)[rank]   # one view per rank
    bg = torch.zeros(3, device=dev)
    pipe, opt = PipelineParams(), OptimizationParams()
    if os.environ.get("GSR_BENCH_PLAIN_ACTIVATIONS"):      # A/B aid: torch activations + reference-signature operator
        pipe.fused_activations = False
    if os.environ.get("GSR_BENCH_UNFACTORED"):             # A/B aid: explicit SH gradient tensors (58 floats per Gaussian)
        pipe.factored_sh_grad = False

    target = GaussianModel(3, device=dev)
    target.create_from_params(perturb(params))
    with torch.no_grad():
        gt = render(cam, target, pipe, bg)["render"].clamp(0, 1).contiguous()
    del target
    model = GaussianModel(3, device=dev)
    model.create_from_params(params)
    model.training_setup(opt)
    force_dp = bool(os.environ.get("GSR_BENCH_FORCE_DP")) and dist.is_initialized()   # rehearsal on one GPU
    vp = ViewParallel(model, force=force_dp) if (world > 1 or force_dp) else None
    if vp is None and os.environ.get("GSR_BENCH_LOCAL_OVERLAP"):     # A/B aid: SH update on a side stream at N = 1
        vp = ViewParallel(model, overlap_local=True)

    base_iter = 10_000
    def step(i):
        training_step(model, cam, gt, opt, pipe, bg, base_iter + i, view_parallel=vp)

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
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    _lib.profile_enable(False)
    prof = _lib.profile_read()
    log(f"timed {args.steps} steps in {elapsed:.3f}s")
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the reference's own `iter_time` bracket (train.py:91,145: forward + loss + backward, no optimiser step),
    # measured AFTER the timed region with device events on a few extra steps (rank 0, informational)
    ref_iter_ms = None
    if rank == 0:
        n_ref = min(20, max(args.steps, 1))
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ref)]
        for a, b in ev:
            a.record()
            training_step(model, cam, gt, opt, pipe, bg, base_iter + args.warmup + args.steps, step_optimizer=False)
            b.record()
            model.optimizer.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        ref_iter_ms = sorted(a.elapsed_time(b) for a, b in ev)[n_ref // 2]
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
        per_kernel = {k: (ms / n if n else 0.0) for k, (ms, n) in prof.items() if k in big and n}
        dom = max(per_kernel, key=per_kernel.get)
        ab = algorithmic_bytes(N, D, P)
        headline = (N, W, H) == (1_000_000, 1920, 1080)      # the committed PMC counters were collected on this workload
        achieved = ab[dom] / (per_kernel[dom] * 1e-3) / 1e9 if per_kernel[dom] > 0 else 0.0
        ms_per_step = elapsed / args.steps * 1e3
        iter_b = iteration_bytes(N, D, P, tiles)
        out = {
            "metric": "train iters/sec @1M Gaussians 1080p", "value": world * args.steps / elapsed, "unit": "iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"synthetic {N}-Gaussian {W}x{H} scene (SURVEY 8(d) recipe, seed 0), SH degree 3, "
                                   f"L1+SSIM+normal loss, Adam; one view per GPU per step",
                       "gaussians": N, "width": W, "height": H, "instances_D": D,
                       "parallelism": f"view-parallel dp{world}" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic_bytes(dom) if headline else None,
                         "traffic_source": "profiles/pmc_traffic.json (separate rocprofv3 --pmc passes, same scene; "
                                           "2 x FETCH_SIZE + WRITE_SIZE)",
                         "algorithmic_bytes_per_launch": ab[dom], "avg_launch_ms": per_kernel[dom]},
            # the dominant kernels are VALU-bound: the same launch priced against the MEASURED
            # vector-issue ceiling of the chip instead of the HBM roofline (informational, not the contract's roofline)
            "valu_issue": (lambda n: None if not n or per_kernel[dom] <= 0 else {
                "kernel": dom, "valu_wave_insts_per_launch": n, "achieved_ginst_s": n / (per_kernel[dom] * 1e-3) / 1e9,
                "ceiling_ginst_s": VALU_CEILING_GINST_S, "frac": n / (per_kernel[dom] * 1e-3) / 1e9 / VALU_CEILING_GINST_S,
                "source": "profiles/pmc_traffic.json (SQ_INSTS_VALU); ceiling: scripts/microbench/valu_rate.hip"})(
                    pmc_valu_insts(dom) if headline else None),
            "hbm_peak_gb": {"allocated": round(torch.cuda.max_memory_allocated(dev) / 2**30, 2),
                            "reserved": round(torch.cuda.max_memory_reserved(dev) / 2**30, 2)},
            "reference_iter_time_ms": ref_iter_ms,    # median of the reference's fwd+loss+bwd bracket (no Adam)
            "kernel_ms": {k: round(v, 4) for k, v in per_kernel.items()},
            "kernel_ms_warmup": {k: round(v, 4) for k, v in warm.items()},
            "kernel_ms_per_step": {k: round(ms / args.steps, 4) for k, (ms, n) in prof.items() if n},
            "iteration": {"algorithmic_bytes": iter_b, "hbm_frac": iter_b / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
        if world == 1 and not args.no_cpu_baseline:
            log(f"GPU part done ({out['value']:.2f} it/s, D={D}); timing the CPU oracle on {host_cores()} cores")
            out["cpu_baseline"] = cpu_baseline(params, cam, args.cpu_tiles, N, dbg, W, H)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
