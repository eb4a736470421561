"""Fused photometric loss on the HIP library: one kernel forward, one backward,
instead of five MIOpen depthwise convolutions each way.

    photometric_loss(image, gt, lambda_dssim) -> (loss, l1, ssim)      # all 0-dim device tensors
      loss = (1 - lambda) * l1 + lambda * (1 - ssim)                   # train.py:113-114

Only `image` receives a gradient (the ground truth never does in the reference either).
"""
import ctypes as C

import torch

from . import _lib


class _PhotometricLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, gt, lambda_dssim):
        L = _lib.lib()
        if image.device.type != "cuda":
            raise _lib.GsrError("photometric_loss needs tensors on a HIP device (torch 'cuda'); there is no CPU path")
        img = image.detach().float().contiguous()
        tgt = gt.detach().float().contiguous()
        if img.dim() != 3 or img.shape != tgt.shape:
            raise ValueError("image and gt must both be [C,H,W]")
        Cn, H, W = img.shape
        dev = img.device
        with torch.cuda.device(dev):
            maps = torch.empty((3, Cn, H, W), dtype=torch.float32, device=dev)
            partials = torch.empty((L.gsr_loss_num_partials(H, W) // 2, 2), dtype=torch.float32, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(L.gsr_loss_forward(C.c_void_p(img.data_ptr()), C.c_void_p(tgt.data_ptr()), Cn, H, W,
                                          C.c_void_p(maps.data_ptr()), C.c_void_p(partials.data_ptr()),
                                          C.c_void_p(stream)))
        sums = partials.sum(0) / float(Cn * H * W)
        ssim, l1 = sums[0], sums[1]
        loss = (1.0 - lambda_dssim) * l1 + lambda_dssim * (1.0 - ssim)
        ctx.save_for_backward(img, tgt, maps)
        ctx.lambda_dssim = float(lambda_dssim)
        ctx.mark_non_differentiable(l1, ssim)
        ctx.set_materialize_grads(False)
        return loss, l1, ssim

    @staticmethod
    def backward(ctx, g_loss, _g_l1, _g_ssim):
        if g_loss is None:
            return None, None, None
        L = _lib.lib()
        img, tgt, maps = ctx.saved_tensors
        Cn, H, W = img.shape
        dev = img.device
        with torch.cuda.device(dev):
            scale = g_loss.detach().float().reshape(1).contiguous()
            dimg = torch.empty_like(img)
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(L.gsr_loss_backward(C.c_void_p(img.data_ptr()), C.c_void_p(tgt.data_ptr()),
                                           C.c_void_p(maps.data_ptr()), Cn, H, W, ctx.lambda_dssim,
                                           C.c_void_p(scale.data_ptr()), C.c_void_p(dimg.data_ptr()),
                                           C.c_void_p(stream)))
        return dimg, None, None


def photometric_loss(image, gt, lambda_dssim=0.2):
    return _PhotometricLoss.apply(image, gt, lambda_dssim)
