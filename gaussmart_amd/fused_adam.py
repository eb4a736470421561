"""torch.optim.Adam whose step() is ONE HIP kernel launch over all parameter groups
(SURVEY 8(f) N2).  State layout (`exp_avg`, `exp_avg_sq`, `step`) and param_groups are exactly
torch's, so the reference's optimiser surgery during densification
(scene/gaussian_model.py:398-470: replace / prune / cat of the moment tensors) works unchanged."""
import ctypes as C
import math

import torch

from . import _lib


class FusedAdam(torch.optim.Adam):
    MAX_TENSORS = 8

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, foreach=False, fused=False)

    @torch.no_grad()
    def step(self, closure=None, only=None, stream=None):
        """`only`: iterable of parameters to update (the others keep their state and step count untouched);
        `stream`: torch.cuda.Stream to launch on instead of the current one.  Both serve the pipelined
        data-parallel step, which updates the geometry tensors and the SH tensors at different times."""
        only_ids = None if only is None else {id(p) for p in only}
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        batches = {}
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if p.grad is None or (only_ids is not None and id(p) not in only_ids):
                    continue
                if p.device.type != "cuda" or p.dtype != torch.float32 or not p.is_contiguous():
                    raise _lib.GsrError("FusedAdam needs contiguous float32 parameters on a HIP device")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                t = float(st["step"])
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                key = (p.device, beta1, beta2, group["eps"])
                batches.setdefault(key, []).append(
                    (p, g, st["exp_avg"], st["exp_avg_sq"], group["lr"] / (1.0 - beta1 ** t), 1.0 / math.sqrt(1.0 - beta2 ** t)))
        for (dev, beta1, beta2, eps), items in batches.items():
            with torch.cuda.device(dev):
                stream_h = (stream if stream is not None else torch.cuda.current_stream(dev)).cuda_stream
                for i in range(0, len(items), self.MAX_TENSORS):
                    chunk = items[i:i + self.MAX_TENSORS]
                    n = len(chunk)
                    arr = lambda k: (C.c_void_p * n)(*[c[k].data_ptr() for c in chunk])
                    _lib.check(L.gsr_adam_step(
                        n, arr(0), arr(1), arr(2), arr(3), (C.c_int64 * n)(*[c[0].numel() for c in chunk]),
                        (C.c_float * n)(*[c[4] for c in chunk]), (C.c_float * n)(*[c[5] for c in chunk]),
                        beta1, beta2, eps, C.c_void_p(stream_h)))
        return loss

    @torch.no_grad()
    def step_slice(self, p, start, stop, stream=None, count_step=True):
        """Adam update of the flat element range [start, stop) of ONE parameter (pipelined data-parallel step: a large
        tensor is updated chunk by chunk as the chunks of its gradient arrive).  `count_step=False` for every chunk
        after the first one of an iteration, so the step count advances once."""
        L = _lib.lib()
        group = next(g for g in self.param_groups if any(q is p for q in g["params"]))
        if p.grad is None or stop <= start:
            return
        if p.device.type != "cuda" or p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
            raise _lib.GsrError("FusedAdam needs contiguous float32 parameters and gradients on a HIP device")
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        if count_step:
            st["step"] += 1
        t = float(st["step"])
        beta1, beta2 = group["betas"]
        off = 4 * int(start)
        ptr = lambda x: (C.c_void_p * 1)(x.data_ptr() + off)
        with torch.cuda.device(p.device):
            stream_h = (stream if stream is not None else torch.cuda.current_stream(p.device)).cuda_stream
            _lib.check(L.gsr_adam_step(
                1, ptr(p), ptr(p.grad), ptr(st["exp_avg"]), ptr(st["exp_avg_sq"]), (C.c_int64 * 1)(int(stop - start)),
                (C.c_float * 1)(group["lr"] / (1.0 - beta1 ** t)), (C.c_float * 1)(1.0 / math.sqrt(1.0 - beta2 ** t)),
                beta1, beta2, group["eps"], C.c_void_p(stream_h)))
