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
        self.pending_sh = None      # (features_dc, features_rest, rasterizer.ColorGradRecord) of the last factored backward
        # Colour cache (gsr_adam_sh_factored_next): while the factored SH step has a Gaussian's new coefficients on chip it can
        # also evaluate the SH colour of the NEXT view -- the next forward then skips the colour pass and its 192-byte read
        # per Gaussian.  `next_view` = (camera centre tensor [3] on the device, active SH degree) of that view, set by the
        # trainer; `color_cache` = (key, tensor f32[13 N]) of the last such step, looked up by render().
        self.next_view = None
        self.color_cache = None
        self._cache_buf = None
        self._xyz_old = None
        # every kernel of this optimiser writes parameters through raw pointers, which torch's version counters never see:
        # the cache key therefore carries this counter, bumped by every call that updates parameters (a cache is keyed
        # with the value it has AFTER the call that built it)
        self._writes = 0

    # ---- factored SH gradient: the two feature parameters have NO .grad after such a backward; their gradient is the
    # record parked here, which the next full step() turns into an update (so `optimizer.step()` keeps meaning "apply
    # the gradients of the last backward" for every caller) and zero_grad() drops.
    def park_sh_gradient(self, f_dc, f_rest, rec):
        self.pending_sh = None if rec is None else (f_dc, f_rest, rec)

    def take_pending_sh(self):
        """The parked record if it still matches the parameters, else None; clears the slot.  After a densification
        the feature tensors were replaced (scene/gaussian_model.py:398-470): like every replaced parameter, whose
        .grad is None, they then sit this step out."""
        pend, self.pending_sh = self.pending_sh, None
        if pend is None:
            return None
        f_dc, f_rest, rec = pend
        live = {id(q) for g in self.param_groups for q in g["params"]}
        if id(f_dc) not in live or id(f_rest) not in live or rec.n != f_dc.shape[0]:
            return None
        return pend

    def zero_grad(self, set_to_none: bool = True):
        self.pending_sh = None
        return super().zero_grad(set_to_none=set_to_none)

    def color_cache_key(self, campos, sh_degree, xyz, f_dc, f_rest):
        """What a colour cache is valid for: the view (its camera-centre tensor), the active degree and the exact parameter
        tensors in their current state.  torch's version counters catch every in-place modification made through torch;
        the optimiser's own kernels write through raw pointers, so its write counter is part of the key: any later
        step() / step_slice() / step_sh_factored() -- with or without a next view -- retires the cache."""
        return (campos.data_ptr(), int(sh_degree), int(xyz.shape[0]), xyz.data_ptr(), f_dc.data_ptr(), f_rest.data_ptr(),
                xyz._version, f_dc._version, f_rest._version, campos._version, self._writes)

    def invalidate_color_cache(self):
        """For code that rewrites parameters behind torch's back (`.data` copies, broadcasts into `.data`, raw pointers)."""
        self._writes += 1
        self.color_cache = None

    def lookup_color_cache(self, campos, sh_degree, xyz, f_dc, f_rest):
        """The cache tensor if the last factored step prepared the colour of exactly this view for exactly these
        parameters, else None (the forward then runs its ordinary SH colour pass)."""
        cc = self.color_cache
        if cc is None or cc[0] != self.color_cache_key(campos, sh_degree, xyz, f_dc, f_rest):
            return None
        return cc[1]

    @torch.no_grad()
    def step(self, closure=None, only=None, stream=None, keep_old=None):
        """`only`: iterable of parameters to update (the others keep their state and step count untouched);
        `stream`: torch.cuda.Stream to launch on instead of the current one.  Both serve the pipelined
        data-parallel step, which updates the geometry tensors and the SH tensors at different times.
        `keep_old`: (parameter, tensor of its shape) -- the update kernel also stores the parameter's values BEFORE the
        update there (gsr_adam_step_keep), which replaces a copy launch beside the step."""
        only_ids = None if only is None else {id(p) for p in only}
        self._writes += 1
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        sh_after = None
        if only is None:
            pend = self.take_pending_sh()
            if pend is not None:
                f_dc, f_rest, rec = pend
                records = rec.gathered if rec.gathered is not None else rec.record
                if self.next_view is not None and rec.xyz.grad is not None:
                    # the SH step will also evaluate the NEXT view's colour, which needs the positions AFTER their own
                    # update: the geometry step below keeps the positions the backward saw (keep_old), then the SH tensors
                    if keep_old is None:
                        keep_old = (rec.xyz, self._snapshot_buffer(rec.xyz))
                    sh_after = (f_dc, f_rest, rec, records, keep_old[1])
                else:       # first: it reads the positions the backward saw, which the launch below updates
                    self.step_sh_factored(f_dc, f_rest, rec.xyz, records, rec.n_views, rec.record.numel(), rec.sh_degree,
                                          rec.grad_scale, stream=stream)
        kept = False
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
                    olds = None
                    if keep_old is not None and any(c[0] is keep_old[0] for c in chunk):
                        kept = True
                        olds = (C.c_void_p * n)(*[keep_old[1].data_ptr() if c[0] is keep_old[0] else None for c in chunk])
                    _lib.check(L.gsr_adam_step_keep(
                        n, arr(0), arr(1), arr(2), arr(3), (C.c_int64 * n)(*[c[0].numel() for c in chunk]),
                        (C.c_float * n)(*[c[4] for c in chunk]), (C.c_float * n)(*[c[5] for c in chunk]),
                        beta1, beta2, eps, olds, C.c_void_p(stream_h)))
        if keep_old is not None and not kept:      # the parameter took no update this step: its old value is its value
            if stream is not None:
                with torch.cuda.stream(stream):
                    keep_old[1].copy_(keep_old[0])
            else:
                keep_old[1].copy_(keep_old[0])
        if sh_after is not None:
            f_dc, f_rest, rec, records, xyz_old = sh_after
            self.step_sh_factored(f_dc, f_rest, xyz_old, records, rec.n_views, rec.record.numel(), rec.sh_degree,
                                  rec.grad_scale, stream=stream, next_view=self.next_view, xyz_next=rec.xyz)
        return loss

    @torch.no_grad()
    def step_with_sh(self, f_dc, f_rest, rec, records, n_views, grad_scale=1.0):
        """Full optimiser step of an iteration whose SH gradient is the factored record `rec` (already taken out of the
        pending slot): geometry tensors by their .grad, SH tensors from `records`.  Without a next view the SH step runs
        first (it reads the positions the backward saw, which the geometry step moves); with one (self.next_view) the
        order is snapshot -> geometry -> SH + colour of the next view from the NEW positions."""
        stride, deg = rec.record.numel(), rec.sh_degree
        if self.next_view is not None and rec.xyz.grad is not None:
            xyz_old = self._snapshot_buffer(rec.xyz)
            self.step(keep_old=(rec.xyz, xyz_old))
            self.step_sh_factored(f_dc, f_rest, xyz_old, records, n_views, stride, deg, grad_scale,
                                  next_view=self.next_view, xyz_next=rec.xyz)
        else:
            self.step_sh_factored(f_dc, f_rest, rec.xyz, records, n_views, stride, deg, grad_scale)
            self.step()

    def _snapshot_buffer(self, xyz):
        if self._xyz_old is None or self._xyz_old.shape != xyz.shape or self._xyz_old.device != xyz.device:
            self._xyz_old = torch.empty_like(xyz)
        return self._xyz_old

    @torch.no_grad()
    def step_slice(self, p, start, stop, stream=None, count_step=True):
        """Adam update of the flat element range [start, stop) of ONE parameter (pipelined data-parallel step: a large
        tensor is updated chunk by chunk as the chunks of its gradient arrive).  `count_step=False` for every chunk
        after the first one of an iteration, so the step count advances once."""
        L = _lib.lib()
        group = next(g for g in self.param_groups if any(q is p for q in g["params"]))
        if p.grad is None or stop <= start:
            return
        self._writes += 1
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

    def _state_of(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def step_sh_factored(self, f_dc, f_rest, xyz, records, n_views, view_stride, sh_degree, grad_scale=1.0,
                         first=0, count=None, stream=None, count_step=True, next_view=None, xyz_next=None):
        """Adam update of the two SH parameters from FACTORED gradients (include/gsr.h: gsr_adam_sh_factored):
        `records` holds n_views blocks of `view_stride` floats, each the [N,3] clamp-masked colour gradient of one view
        followed by that view's camera position (rasterizer.ColorGradRecord.record, or the all-gather of it over the
        ranks of a view-parallel step).  `xyz` must be the positions the backward saw: call this BEFORE updating xyz,
        or pass a snapshot.  [first, first + count) restricts the update to a range of Gaussians; `count_step=False`
        for every range after the first one of an iteration."""
        L = _lib.lib()
        N = f_dc.shape[0]
        count = N - first if count is None else count
        if count <= 0:
            return
        self._writes += 1
        for t in (f_dc, f_rest, xyz, records):
            if t.device.type != "cuda" or t.dtype != torch.float32 or not t.is_contiguous():
                raise _lib.GsrError("step_sh_factored needs contiguous float32 tensors on a HIP device")
        if f_rest.shape[0] != N or xyz.shape[0] != N or view_stride < 3 * N + 3 or records.numel() < n_views * view_stride:
            raise _lib.GsrError("step_sh_factored: shapes do not match")
        ptrs, sizes = [], []
        for p in (f_dc, f_rest):
            group = next(g for g in self.param_groups if any(q is p for q in g["params"]))
            st = self._state_of(p)
            if count_step:
                st["step"] += 1
            t = float(st["step"])
            beta1, beta2 = group["betas"]
            ptrs.append((p, st["exp_avg"], st["exp_avg_sq"], group["lr"] / (1.0 - beta1 ** t), 1.0 / math.sqrt(1.0 - beta2 ** t)))
            sizes.append((beta1, beta2, group["eps"]))
        if sizes[0] != sizes[1]:
            raise _lib.GsrError("step_sh_factored: f_dc and f_rest must share betas and eps")
        (beta1, beta2, eps) = sizes[0]
        M = 1 + f_rest.shape[1]
        want_next = next_view is not None and xyz_next is not None and first == 0 and count == N and M >= 1
        if want_next:
            campos_next, deg_next = next_view
            if (not campos_next.is_cuda or campos_next.dtype != torch.float32 or campos_next.numel() != 3
                    or not campos_next.is_contiguous() or xyz_next.shape[0] != N or not xyz_next.is_contiguous()):
                want_next = False
        with torch.cuda.device(f_dc.device):
            stream_h = (stream if stream is not None else torch.cuda.current_stream(f_dc.device)).cuda_stream
            a, b = ptrs
            if want_next:
                if self._cache_buf is None or self._cache_buf.numel() != 13 * N or self._cache_buf.device != f_dc.device:
                    self._cache_buf = torch.empty(13 * N, dtype=torch.float32, device=f_dc.device)
                _lib.check(L.gsr_adam_sh_factored_next(
                    int(first), int(count), M, int(sh_degree), C.c_void_p(xyz.data_ptr()), int(n_views),
                    C.c_void_p(records.data_ptr()), int(view_stride), C.c_void_p(records.data_ptr() + 12 * N), int(view_stride),
                    float(grad_scale),
                    C.c_void_p(a[0].data_ptr()), C.c_void_p(a[1].data_ptr()), C.c_void_p(a[2].data_ptr()), a[3], a[4],
                    C.c_void_p(b[0].data_ptr()), C.c_void_p(b[1].data_ptr()), C.c_void_p(b[2].data_ptr()), b[3], b[4],
                    beta1, beta2, eps, C.c_void_p(xyz_next.data_ptr()), C.c_void_p(campos_next.data_ptr()), int(deg_next), int(N),
                    C.c_void_p(self._cache_buf.data_ptr()), C.c_void_p(stream_h)))
                self.color_cache = (self.color_cache_key(campos_next, deg_next, xyz_next, f_dc, f_rest), self._cache_buf)
                return
            _lib.check(L.gsr_adam_sh_factored(
                int(first), int(count), M, int(sh_degree), C.c_void_p(xyz.data_ptr()), int(n_views),
                C.c_void_p(records.data_ptr()), int(view_stride), C.c_void_p(records.data_ptr() + 12 * N), int(view_stride),
                float(grad_scale),
                C.c_void_p(a[0].data_ptr()), C.c_void_p(a[1].data_ptr()), C.c_void_p(a[2].data_ptr()), a[3], a[4],
                C.c_void_p(b[0].data_ptr()), C.c_void_p(b[1].data_ptr()), C.c_void_p(b[2].data_ptr()), b[3], b[4],
                beta1, beta2, eps, C.c_void_p(stream_h)))
