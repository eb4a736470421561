"""Command-line training entry: the build's counterpart of the reference's train.py / render.py
front ends for the hot path (scene in -> optimised surfels + PSNR out).  SAM / DINO preprocessing,
LPIPS, TensorBoard and the viewer socket of the reference are out of scope.

    python -m gaussmart_amd.train_cli -s <colmap-or-blender-scene> -m <output dir> [--iterations 30000] [--eval]
    torchrun --nproc-per-node 8 -m gaussmart_amd.train_cli -s ... -m ...        # view-parallel
"""
import argparse
import json
import os
import time

import torch

from .gaussian_model import GaussianModel
from .gaussian_renderer import render
from .losses import psnr
from .params import OptimizationParams, PipelineParams
from .scene_io import Scene
from .trainer import train, TrainState
from .view_parallel import ViewParallel


def evaluate(gaussians, cameras, pipe, background):
    vals = []
    with torch.no_grad():
        for cam in cameras:
            img = render(cam, gaussians, pipe, background, surface_maps=False)["render"].clamp(0, 1)
            vals.append(psnr(img[None], cam.original_image.to(img.device)[None]).mean())
    return float(torch.stack(vals).mean()) if vals else float("nan")


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--source_path", "-s", required=True)
    ap.add_argument("--model_path", "-m", required=True)
    ap.add_argument("--images", "-i", default=None)
    ap.add_argument("--resolution", "-r", type=int, default=-1)
    ap.add_argument("--white_background", "-w", action="store_true")
    ap.add_argument("--eval", action="store_true")
    ap.add_argument("--sh_degree", type=int, default=3)
    ap.add_argument("--iterations", type=int, default=30_000)
    ap.add_argument("--save_iterations", type=int, nargs="*", default=[7000, 30000])
    ap.add_argument("--depth_ratio", type=float, default=0.0)
    ap.add_argument("--lambda_normal", type=float, default=0.05)
    ap.add_argument("--lambda_dist", type=float, default=0.0)
    ap.add_argument("--start_checkpoint", default=None)
    ap.add_argument("--log_every", type=int, default=500)
    args = ap.parse_args(argv)

    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    torch.manual_seed(0)

    opt = OptimizationParams(iterations=args.iterations, lambda_normal=args.lambda_normal, lambda_dist=args.lambda_dist)
    pipe = PipelineParams(depth_ratio=args.depth_ratio)
    gaussians = GaussianModel(args.sh_degree, device=dev)
    scene = Scene(args.source_path, gaussians, model_path=args.model_path, images=args.images, eval=args.eval,
                  white_background=args.white_background, resolution=args.resolution, data_device=dev)
    gaussians.training_setup(opt)
    first_iter = 0
    if args.start_checkpoint:
        # capture() holds ints, floats, tensors / Parameters and the optimiser's state_dict: the weights-only loader
        # (no arbitrary pickle code from the file) is sufficient
        model_params, first_iter = torch.load(args.start_checkpoint, map_location=dev, weights_only=True)
        gaussians.restore(model_params, opt)
    background = torch.tensor([1.0, 1.0, 1.0] if args.white_background else [0.0, 0.0, 0.0], device=dev)
    # one rank: no exchange; past the densification phase train() takes the same pipelined step as N > 1 (SH update on a
    # side stream beside the next forward's binning), before it the serial step behind the densification bookkeeping
    vp = ViewParallel(gaussians) if world > 1 else ViewParallel(gaussians, overlap_local=True)

    os.makedirs(args.model_path, exist_ok=True)
    t0 = time.time()
    done = first_iter
    state = TrainState(seed=0)      # one view sampler for the whole run: saving does not perturb the trajectory
    for stop in sorted(set([i for i in args.save_iterations if first_iter < i <= args.iterations] + [args.iterations])):
        train(gaussians, scene.getTrainCameras(), opt, pipe, background, cameras_extent=scene.cameras_extent,
              first_iter=done, iterations=stop, view_parallel=vp, white_background=args.white_background,
              log_every=args.log_every if rank == 0 else 0, final_iteration=args.iterations, state=state)
        done = stop
        if vp is not None:
            vp.finish()
        if rank == 0:
            scene.save(stop)
            torch.save((gaussians.capture(), stop), os.path.join(args.model_path, f"chkpnt{stop}.pth"))
    if rank == 0:
        res = {"iterations": done, "seconds": time.time() - t0, "points": int(gaussians.get_xyz.shape[0]),
               "psnr_train": evaluate(gaussians, scene.getTrainCameras()[:8], pipe, background),
               "psnr_test": evaluate(gaussians, scene.getTestCameras(), pipe, background)}
        with open(os.path.join(args.model_path, "results.json"), "w") as f:
            json.dump(res, f, indent=1)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
