"""One training iteration and the surrounding loop.

Counterpart of training() in the reference (train.py:45-243): random view, render, loss
0.8*L1 + 0.2*(1-SSIM) + lambda_normal*normal + lambda_dist*dist, backward, densification
bookkeeping, Adam step.  The reference's per-iteration `.item()` calls, CSV append, TensorBoard,
DINO scalar (no gradient, needs a network fetch), LPIPS and the viewer socket are left out: none
of them reaches the rasterizer, and the host syncs they cause are a throughput hazard.
"""
import random
import time

import torch

from .gaussian_renderer import render
from . import rasterizer as _rasterizer
from .fused_adam import FusedAdam
from .fused_loss import photometric_loss as fused_photometric_loss
from .fused_objective import training_objective
from .losses import l1_loss, ssim
from .view_parallel import ViewParallel


def photometric(image, gt_image, lambda_dssim):
    """(1-l)*L1 + l*(1-SSIM), train.py:113-114.  Device tensors go through the fused HIP kernels
    (gaussmart_amd/fused_loss.py); host tensors (the CPU plumbing tests) through the stock torch
    formulation the reference itself uses (utils/loss_utils.py)."""
    if image.is_cuda:
        loss, Ll1, _ = fused_photometric_loss(image, gt_image, lambda_dssim)
        return loss, Ll1
    Ll1 = l1_loss(image, gt_image)
    return (1.0 - lambda_dssim) * Ll1 + lambda_dssim * (1.0 - ssim(image, gt_image)), Ll1


_UNIT = {}


def _unit_gradient(t):
    key = (t.device, t.dtype)
    one = _UNIT.get(key)
    if one is None:
        one = _UNIT[key] = torch.ones((), device=t.device, dtype=t.dtype)
    return one


def training_losses(render_pkg, gt_image, opt, iteration, viewpoint_cam=None, pipe=None, defer_value=False):
    """train.py:113-143.  On a HIP device the whole objective (L1 + SSIM + surface regularizers) is
    one fused autograd node (gaussmart_amd/fused_objective.py); on the host (CPU plumbing tests)
    the stock torch formulation of the reference is used on the maps render() derived -- and on a device too when
    `pipe.reference_objective` is set (then with utils/loss_utils.py's torch L1 + SSIM: the reference-shaped loop).
    (The diagnostics follow what the forward was asked to build: "dist_mean" reads 0 while lambda_dist = 0 -- the reference
    logs lambda_dist * mean there, i.e. 0 as well -- and "normal_mean" reads 0 while no regularizer is active.)"""
    image = render_pkg["render"]
    lambda_normal = opt.lambda_normal if iteration > 7000 else 0.0
    lambda_dist = opt.lambda_dist if iteration > 3000 else 0.0
    if "rend_normal" not in render_pkg:
        total, parts = training_objective(image, render_pkg["allmap"], gt_image, viewpoint_cam, opt.lambda_dssim,
                                          lambda_normal, lambda_dist, pipe.depth_ratio, defer_value=defer_value)
        l1, ssim_v = parts[0], parts[1]
        return total, {"l1": l1, "ssim": ssim_v, "normal_mean": parts[2], "dist_mean": parts[3], "loss": total.detach()}
    if image.is_cuda and getattr(pipe, "reference_objective", False):      # utils/loss_utils.py's torch L1 + SSIM, on the device
        Ll1 = l1_loss(image, gt_image)
        loss = (1.0 - opt.lambda_dssim) * Ll1 + opt.lambda_dssim * (1.0 - ssim(image, gt_image))
    else:
        loss, Ll1 = photometric(image, gt_image, opt.lambda_dssim)
    normal_error = (1 - (render_pkg["rend_normal"] * render_pkg["surf_normal"]).sum(dim=0))[None]
    normal_loss = lambda_normal * normal_error.mean()
    dist_loss = lambda_dist * render_pkg["rend_dist"].mean()
    total = loss + dist_loss + normal_loss
    return total, {"l1": Ll1.detach(), "loss": loss.detach(), "normal": normal_loss.detach(), "dist": dist_loss.detach()}


def _use_factored_sh_grad(gaussians, pipe, render_fn, on_device):
    """The backward may leave dL/drgb [N,3] instead of the SH gradient tensors only when this module also performs the
    optimiser step with the kernel that understands it (FusedAdam.step_sh_factored) on the raw-parameter path."""
    return (on_device and render_fn is render and getattr(pipe, "factored_sh_grad", False)
            and not getattr(pipe, "reference_objective", False)        # (that formulation reads the maps render() derives)
            and getattr(gaussians, "raster_state", None) is not None
            and getattr(pipe, "fused_activations", False) and isinstance(getattr(gaussians, "optimizer", None), FusedAdam))


def optimizer_step(gaussians, view_parallel: ViewParallel = None):
    """Optimiser step of one iteration, including the gradient exchange of a view-parallel step (unless
    training_step(step_optimizer=False) already did it: then call this without `view_parallel`)."""
    optimizer = gaussians.optimizer
    if view_parallel is not None:
        rec = optimizer.take_pending_sh() if isinstance(optimizer, FusedAdam) else None
        view_parallel.reduce_and_step(optimizer, rec[2] if rec is not None else None)   # pipelined when RCCL allows it
    else:
        optimizer.step()        # FusedAdam applies a parked factored SH gradient itself
    optimizer.zero_grad(set_to_none=True)


def training_step(gaussians, viewpoint_cam, gt_image, opt, pipe, background, iteration,
                  view_parallel: ViewParallel = None, render_fn=render, step_optimizer=True, next_cam=None):
    """forward + loss + backward (+ gradient exchange) (+ Adam).  Returns (render_pkg, losses);
    nothing is synchronised with the host.  With `step_optimizer=False` the gradients are exchanged here (view-parallel
    runs) and the caller finishes the iteration with optimizer_step(gaussians) or gaussians.optimizer.step() -- train()
    does, after the densification bookkeeping.
    `next_cam`: the view the NEXT iteration will render, if the caller knows it: the factored SH optimiser step then also
    leaves that view's SH colours (it has the new coefficients on chip anyway) and the next forward skips its colour pass."""
    gaussians.update_learning_rate(iteration)
    on_device = gaussians.get_xyz.is_cuda
    factored = _use_factored_sh_grad(gaussians, pipe, render_fn, on_device)
    if isinstance(getattr(gaussians, "optimizer", None), FusedAdam):
        gaussians.optimizer.next_view = (next_cam.camera_center, gaussians.active_sh_degree) \
            if (factored and next_cam is not None and getattr(pipe, "color_cache", True)) else None
    if factored:
        # while no regularizer is active (train.py:132-133: lambda_normal from iteration 7,000, lambda_dist from 3,000 and 0 by
        # default; scripts/dtu_eval.py:45 trains with both at 0) nobody reads allmap: the forward does not build it
        no_reg = not ((opt.lambda_normal > 0.0 and iteration > 7000) or (opt.lambda_dist > 0.0 and iteration > 3000))
        lean = getattr(pipe, "color_only_when_unregularized", True) and _rasterizer._NO_SURFACE_FAST_PATH   # (=0: A/B aid)
        # ... and with the reference's defaults (lambda_dist = 0, depth_ratio = 0: arguments/__init__.py:72,87) nobody ever reads
        # the distortion and median-depth channels: the forward does not accumulate them
        no_dm = not (opt.lambda_dist > 0.0 and iteration > 3000) and float(getattr(pipe, "depth_ratio", 0.0)) == 0.0
        render_pkg = render_fn(viewpoint_cam, gaussians, pipe, background, surface_maps=False, factored_sh_grad=True,
                               color_only=no_reg and lean, no_dist_median=no_dm and lean)
    else:
        render_pkg = render_fn(viewpoint_cam, gaussians, pipe, background,
                               surface_maps=(not on_device) or bool(getattr(pipe, "reference_objective", False)))
    # (the backward follows at once: the loss scalars are written by a workgroup of its first kernel, not by a launch of
    # their own -- nobody reads them before this function returns)
    total, parts = training_losses(render_pkg, gt_image, opt, iteration, viewpoint_cam, pipe, defer_value=True)
    total.backward(gradient=_unit_gradient(total))   # cached: saves the ones_like() fill of every step
    rec = gaussians.raster_state.take_color_grad() if factored else None
    if factored:
        gaussians.optimizer.park_sh_gradient(gaussians._features_dc, gaussians._features_rest, rec)
    if step_optimizer:
        optimizer_step(gaussians, view_parallel)
    elif view_parallel is not None:
        if rec is not None:
            view_parallel.exchange_factored(rec)
        else:
            view_parallel.allreduce_gradients()
    parts["total"] = total.detach()
    return render_pkg, parts


def densification_step(gaussians, render_pkg, opt, iteration, cameras_extent, white_background=False,
                       view_parallel: ViewParallel = None):
    """train.py:198-211: statistics every iteration, densify / prune every
    `densification_interval`, opacity reset every `opacity_reset_interval`."""
    if iteration >= opt.densify_until_iter:
        return
    if view_parallel is not None:
        view_parallel.finish()       # a pipelined step may still be updating the SH tensors on its side stream
    with torch.no_grad():
        gaussians.update_densification_stats(render_pkg["viewspace_points"], render_pkg["radii"])
        if iteration > opt.densify_from_iter and iteration % opt.densification_interval == 0:
            gen = None
            if view_parallel is not None:
                view_parallel.sync_densification_stats()
                gen = view_parallel.replicated_generator(iteration, gaussians.get_xyz.device)
            size_threshold = 20 if iteration > opt.opacity_reset_interval else None
            gaussians.densify_and_prune(opt.densify_grad_threshold, opt.opacity_cull, cameras_extent,
                                        size_threshold, generator=gen)
        if iteration % opt.opacity_reset_interval == 0 or (white_background and iteration == opt.densify_from_iter):
            gaussians.reset_opacity()


class TrainState:
    """What a training run carries from one iteration to the next besides the model: the view sampler (RNG, the
    current epoch's stack of unvisited views, the epoch count).  Pass the same object to consecutive train() calls
    (train_cli.py saves between them) and the trajectory is the one of a single uninterrupted call."""

    def __init__(self, seed=0):
        self.seed = seed
        self.rng = random.Random(seed)
        self.stack, self.epoch = None, 0
        self.next_cam = None        # the view of the next iteration, drawn one iteration ahead (see train())

    def draw(self, cameras, view_parallel=None):
        """Next view of the schedule: epochs are shuffled stacks popped at random (train.py:99-102)."""
        if not self.stack:
            self.stack = list(view_parallel.shard_views(cameras, self.seed + self.epoch)) if view_parallel else list(cameras)
            self.epoch += 1
        return self.stack.pop(self.rng.randint(0, len(self.stack) - 1))


def train(gaussians, cameras, opt, pipe, background, *, cameras_extent=1.0, first_iter=0, iterations=None,
          view_parallel: ViewParallel = None, white_background=False, seed=0, log_every=0, log_fn=print,
          final_iteration=None, state: TrainState = None, on_iteration=None):
    """cameras: objects with .original_image [3,H,W].  Epochs are shuffled stacks popped at random,
    as train.py:99-102; under view parallelism each rank pops from its shard of the same shuffle.

    Runs iterations first_iter+1 .. iterations.  As in the reference (train.py:214-216: `if iteration < opt.iterations`)
    only the LAST iteration of the whole schedule skips the optimiser step: `final_iteration` (default: opt.iterations
    when this call ends there, else none) names it, so a run split into several calls -- one per save point -- steps on
    every intermediate stop exactly like an uninterrupted one.  `state` keeps the view sampler across such calls;
    `on_iteration(iteration)` is called after each iteration's optimiser step (saving / evaluation hooks)."""
    iterations = opt.iterations if iterations is None else iterations
    if final_iteration is None:
        final_iteration = opt.iterations
    st = state if state is not None else TrainState(seed)
    device = gaussians.get_xyz.device
    t0 = time.time()
    last = None
    for iteration in range(first_iter + 1, iterations + 1):
        if iteration % 1000 == 0:
            gaussians.oneupSHdegree()
        # views are drawn ONE iteration ahead (same sequence, same RNG use): the optimiser step of this iteration can then
        # leave the SH colours of the next view (training_step(next_cam=...)), which skips that iteration's colour pass
        cam = st.next_cam if st.next_cam is not None else st.draw(cameras, view_parallel)
        st.next_cam = st.draw(cameras, view_parallel)
        gt = cam.original_image.to(device)
        nxt = st.next_cam if iteration < final_iteration else None
        if view_parallel is not None and iteration >= opt.densify_until_iter and iteration < final_iteration:
            # past the densification phase (train.py:198: `if iteration < opt.densify_until_iter`) nothing sits between the
            # backward and the optimiser step, so the iteration takes the pipelined step of ViewParallel.reduce_and_step --
            # the step bench.py times: the SH update (and, for N > 1, the colour-gradient all-gather) on the side stream
            # beside the next forward's binning
            render_pkg, last = training_step(gaussians, cam, gt, opt, pipe, background, iteration,
                                             view_parallel=view_parallel, step_optimizer=True, next_cam=nxt)
        else:
            # the optimizer step comes after the densification bookkeeping, as in the reference
            render_pkg, last = training_step(gaussians, cam, gt, opt, pipe, background, iteration,
                                             view_parallel=view_parallel, step_optimizer=False, next_cam=nxt)
            densification_step(gaussians, render_pkg, opt, iteration, cameras_extent, white_background, view_parallel)
            if iteration < final_iteration:
                optimizer_step(gaussians)      # the exchange of a view-parallel run already happened in training_step
            else:
                gaussians.optimizer.zero_grad(set_to_none=True)   # the schedule's last iteration: no step (train.py:214)
        if on_iteration is not None:
            on_iteration(iteration)
        if log_every and iteration % log_every == 0:
            if view_parallel is not None:
                view_parallel.finish()
            log_fn(f"[it {iteration}] loss {float(last['loss']):.5f} points {gaussians.get_xyz.shape[0]} "
                   f"{(iteration - first_iter) / (time.time() - t0):.2f} it/s")
    if view_parallel is not None:
        view_parallel.finish()       # a pipelined last step may still be updating the SH tensors on its side stream
    return last
