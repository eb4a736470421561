"""The complete training objective of one iteration as ONE autograd node on the HIP library:

    total = (1-l)*L1(image, gt) + l*(1 - SSIM(image, gt))                   train.py:113-114
          + lambda_normal * mean(1 - rend_normal . surf_normal)             train.py:135-139
          + lambda_dist   * mean(rend_dist)                                 train.py:140

Forward: gsr_loss_forward (+ gsr_regularizer_forward) + gsr_objective_finish = 3 launches;
backward: gsr_loss_backward (+ gsr_regularizer_backward) = 2 launches.  The same kernels as
fused_loss.py / fused_regularizer.py, minus the ~25 zero-dimensional torch kernels that combining
their scalars in Python costs (each ~4.7 us of GPU timeline at this scale).
"""
import ctypes as C

import torch

from . import _lib
from .fused_regularizer import camera_kinv


class _Objective(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, allmap, gt, kinv, lambda_dssim, lambda_normal, lambda_dist, depth_ratio, defer_value=False,
                job=None):
        L = _lib.lib()
        if image.device.type != "cuda":
            raise _lib.GsrError("training_objective needs tensors on a HIP device (torch 'cuda'); there is no CPU path")
        img = image.detach().float().contiguous()
        tgt = gt.detach().float().contiguous()
        Cn, H, W = img.shape
        dev = img.device
        use_reg = allmap is not None and (lambda_normal > 0.0 or lambda_dist > 0.0)
        am = allmap.detach().float().contiguous() if use_reg else None
        with torch.cuda.device(dev):
            n_part = L.gsr_loss_num_partials(H, W)
            maps = torch.empty((3, Cn, H, W), dtype=torch.float32, device=dev)
            partials = torch.empty((2, n_part), dtype=torch.float32, device=dev)
            # (with defer_value the five scalars are only written during the backward: until then they read NaN -- the loss
            # forward's launch fills them in -- so a value taken without a backward, or before it, is visibly invalid)
            out = torch.empty(5, dtype=torch.float32, device=dev)
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            # the scan the rasterizer's backward starts with rides along with these launches, if the forward that produced
            # `image` offered it (training_objective took the job from image.grad_fn)
            if not ctx.needs_input_grad[0]:
                job = None
            ctx.row_scan_job = job
            if job is not None or defer_value:
                _lib.check(L.gsr_loss_forward_job(C.c_void_p(img.data_ptr()), C.c_void_p(tgt.data_ptr()), Cn, H, W,
                                                  C.c_void_p(maps.data_ptr()), C.c_void_p(partials[0].data_ptr()),
                                                  C.byref(job) if job is not None else None,
                                                  C.c_void_p(out.data_ptr()) if defer_value else None, stream))
            else:
                _lib.check(L.gsr_loss_forward(C.c_void_p(img.data_ptr()), C.c_void_p(tgt.data_ptr()), Cn, H, W,
                                              C.c_void_p(maps.data_ptr()), C.c_void_p(partials[0].data_ptr()), stream))
            # defer_value (training_step: the backward follows at once and nobody reads the value in between): the five
            # scalars are computed by a workgroup riding along with a kernel of the backward (gsr_loss_backward_finish), and
            # the regularizer's forward sums -- which only feed those scalars -- come from its backward kernel, which
            # evaluates every pixel's surface normal anyway (gsr_regularizer_backward_partials): no forward launch of it
            defer = bool(defer_value and ctx.needs_input_grad[0] and (not use_reg or ctx.needs_input_grad[1]))
            reg_ptr = None
            if use_reg and not defer:
                _lib.check(L.gsr_regularizer_forward(C.c_void_p(am.data_ptr()), H, W, kinv, float(depth_ratio),
                                                     C.c_void_p(partials[1].data_ptr()), stream))
            if use_reg:
                reg_ptr = C.c_void_p(partials[1].data_ptr())
            ctx.deferred = None
            if defer:
                ctx.deferred = (partials, out, use_reg)
            else:
                _lib.check(L.gsr_objective_finish(C.c_void_p(partials[0].data_ptr()), Cn, H, W, reg_ptr,
                                                  float(lambda_dssim), float(lambda_normal), float(lambda_dist),
                                                  C.c_void_p(out.data_ptr()), stream))
        ctx.use_reg = use_reg
        ctx.cfg = (kinv, float(lambda_dssim), float(lambda_normal), float(lambda_dist), float(depth_ratio))
        ctx.has_allmap = allmap is not None
        if use_reg:
            ctx.save_for_backward(img, tgt, maps, am)
        else:
            ctx.save_for_backward(img, tgt, maps)
        total, parts = out[0], out[1:]
        ctx.mark_non_differentiable(parts)
        ctx.set_materialize_grads(False)
        return total, parts

    @staticmethod
    def backward(ctx, g_total, _g_parts):
        if g_total is None:
            return (None,) * 10
        L = _lib.lib()
        saved = ctx.saved_tensors
        img, tgt, maps = saved[0], saved[1], saved[2]
        kinv, ld, ln, ldist, ratio = ctx.cfg
        Cn, H, W = img.shape
        dev = img.device
        dam = None
        with torch.cuda.device(dev):
            scale = g_total.detach().float().reshape(1).contiguous()
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            dimg = torch.empty_like(img)
            from .rasterizer import row_scan_job_alive
            job = getattr(ctx, "row_scan_job", None)
            ctx.row_scan_job = None
            # the second half may only be enqueued while the rasterizer's backward has not run (afterwards the job's
            # buffers are back in the pool), and on the stream the first half was ordered on
            if not row_scan_job_alive(job) or job._stream != torch.cuda.current_stream(dev).cuda_stream:
                job = None
            deferred, ctx.deferred = ctx.deferred, None
            if ctx.use_reg:       # first: with a deferred value its launch also leaves the regularizer's forward sums
                am = saved[3]
                dam = torch.empty_like(am)
                if deferred is not None:
                    _lib.check(L.gsr_regularizer_backward_partials(
                        C.c_void_p(am.data_ptr()), H, W, kinv, ratio, ln, ldist, C.c_void_p(scale.data_ptr()),
                        C.c_void_p(dam.data_ptr()), C.c_void_p(deferred[0][1].data_ptr()), stream))
                else:
                    _lib.check(L.gsr_regularizer_backward(C.c_void_p(am.data_ptr()), H, W, kinv, ratio, ln, ldist,
                                                          C.c_void_p(scale.data_ptr()), C.c_void_p(dam.data_ptr()), stream))
            if deferred is not None or job is not None:
                partials, out, with_reg = deferred if deferred is not None else (None, None, False)
                _lib.check(L.gsr_loss_backward_finish(
                    C.c_void_p(img.data_ptr()), C.c_void_p(tgt.data_ptr()), C.c_void_p(maps.data_ptr()), Cn, H, W, ld,
                    C.c_void_p(scale.data_ptr()), C.c_void_p(dimg.data_ptr()),
                    C.c_void_p(partials[0].data_ptr()) if partials is not None else None,
                    C.c_void_p(partials[1].data_ptr()) if with_reg else None, ln, ldist,
                    C.c_void_p(out.data_ptr()) if out is not None else None,
                    C.byref(job) if job is not None else None, stream))
            else:
                _lib.check(L.gsr_loss_backward(C.c_void_p(img.data_ptr()), C.c_void_p(tgt.data_ptr()),
                                               C.c_void_p(maps.data_ptr()), Cn, H, W, ld, C.c_void_p(scale.data_ptr()),
                                               C.c_void_p(dimg.data_ptr()), stream))
        return dimg, dam, None, None, None, None, None, None, None, None


def training_objective(image, allmap, gt, viewpoint_camera, lambda_dssim=0.2, lambda_normal=0.0, lambda_dist=0.0,
                       depth_ratio=0.0, defer_value=False):
    """-> (total, parts) with parts = [l1, ssim, mean normal error, mean distortion] (device tensor).
    `defer_value`: for callers that call backward() right away and read the values only afterwards -- the five scalars
    are then written during the backward (one launch less in the iteration); before it (or without it) they read NaN."""
    use_reg = allmap is not None and (lambda_normal > 0.0 or lambda_dist > 0.0)
    kinv = camera_kinv(viewpoint_camera) if use_reg else None
    job = None
    if image.is_cuda and image.requires_grad and torch.is_grad_enabled():
        from .rasterizer import take_row_scan_job
        job = take_row_scan_job(image)
    return _Objective.apply(image, allmap if use_reg else None, gt, kinv, lambda_dssim, lambda_normal, lambda_dist,
                            depth_ratio, bool(defer_value), job)
