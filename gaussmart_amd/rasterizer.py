"""`diff_surfel_rasterization` operator surface on top of libgsr_hip.so.

Mirrors what the reference imports and calls (gaussian_renderer/__init__.py:14, 37-53, 97-106):
`GaussianRasterizationSettings(image_height, image_width, tanfovx, tanfovy, bg, scale_modifier,
viewmatrix, projmatrix, sh_degree, campos, prefiltered, debug)` and
`GaussianRasterizer(raster_settings)(means3D, means2D, opacities, shs=, colors_precomp=, scales=,
rotations=, cov3D_precomp=) -> (color [3,H,W], radii int32 [N], allmap [7,H,W])`.
All compute happens in hand-written HIP kernels; this file only marshals pointers.
"""
import ctypes as C
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _lib


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


# quirk flags handed to the kernels (see include/gsr.h); tests flip this to compare against the
# gradcheck-clean oracle
DEFAULT_FLAGS = _lib.GSR_FLAGS_UPSTREAM


def _f32c(t, name, device):
    if t is None:
        return None
    if t.device != device:
        raise ValueError(f"{name} is on {t.device}, expected {device}")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


_PLACEHOLDER = {}
_ZERO_IMAGES = {}


def _zero_image(channels, H, W, device):
    """Read-only zero gradient image for an output nobody differentiated (e.g. allmap before the regularizers
    switch on): cached per shape, so the backward does not fill 58 MB every step."""
    key = (int(channels), int(H), int(W), torch.device(device))
    z = _ZERO_IMAGES.get(key)
    if z is None:
        z = _ZERO_IMAGES[key] = torch.zeros((channels, H, W), dtype=torch.float32, device=device)
    return z


def _ptr(t):
    """Device pointer of a tensor; None -> NULL.  Empty tensors have a NULL data_ptr, but NULL means
    "argument absent" in the C ABI, so they get the address of a small per-device placeholder."""
    if t is None:
        return C.c_void_p(0)
    if t.numel() == 0:
        ph = _PLACEHOLDER.get(t.device)
        if ph is None:
            ph = _PLACEHOLDER[t.device] = torch.zeros(64, dtype=torch.float32, device=t.device)
        return C.c_void_p(ph.data_ptr())
    return C.c_void_p(t.data_ptr())


def _round_up(n, q):
    return (int(n) + q - 1) // q * q


class _BufferPool:
    """Grow-only pool of the byte buffers the library asks for (include/gsr.h: gsr_alloc_fn).

    Their sizes follow the instance count D, which drifts by a fraction of a percent every
    iteration.  Handing such requests to the caching allocator directly makes it hipMalloc a
    slightly larger block every few iterations (each one a device synchronisation, and the smaller
    block stays cached forever): the loop stalls and reserved memory creeps up.  The pool keeps the
    buffers it ever created, per (device, stream, kind), with 25 % headroom, and leases them out:
      * scratch kinds are only live inside one library call -> released when the call returns;
      * geom / binning / image are needed by the backward -> released when the autograd node that
        holds the lease dies (after backward, or when the outputs are dropped).
    All leases of one pool key are used on one stream, so reuse is ordered by the stream."""

    def __init__(self):
        self.free = {}      # key -> list of tensors

    def take(self, key, nbytes, device):
        lst = self.free.setdefault(key, [])
        best = None
        for i, t in enumerate(lst):
            if t.numel() >= nbytes and (best is None or t.numel() < lst[best].numel()):
                best = i
        if best is not None:
            return lst.pop(best)
        if lst:                                   # too small: let the largest one go, grow once
            lst.sort(key=lambda t: t.numel())
            lst.pop()
        STATS["pool_buffers_created"] += 1
        STATS["pool_bytes_created"] += _round_up(nbytes + nbytes // 4, 1 << 22)
        if _POOL_DEBUG:
            print(f"[gsr pool] new buffer kind {key[2]}: need {nbytes / 2**20:.1f} MiB (had {[round(t.numel() / 2**20, 1) for t in lst]})", flush=True)
        return torch.empty(_round_up(nbytes + nbytes // 4, 1 << 22), dtype=torch.uint8, device=device)

    def give(self, key, t):
        self.free.setdefault(key, []).append(t)

    def clear(self):
        self.free.clear()


_POOL = _BufferPool()
STATS = {"color_pass_on_second_stream": 0,     # how often a forward put its SH colour pass on the updater's stream
         "row_scans_carried": 0,               # how often a backward found its row scan done by the objective's kernels
         "pool_buffers_created": 0,            # grow-only workspace: buffers the pool had to create (first use or re-grow)
         "pool_bytes_created": 0}
_POOL_DEBUG = bool(__import__("os").environ.get("GSR_POOL_DEBUG"))


class _Lease:
    """Buffers of one forward call; returns them to the pool when garbage collected."""

    def __init__(self):
        self.items = []     # (key, tensor)
        self.released = False

    def release(self):
        for key, t in self.items:
            _POOL.give(key, t)
        self.items = []
        self.released = True

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class _Allocator:
    """Adapter between gsr_alloc_fn and the pool for ONE library call."""

    def __init__(self, device, before_color=None, color_stream=None):
        self.device = device
        self.buffers = {}
        self.error = None
        self.before_color = before_color        # called once, when the library announces the SH colour pass
        self.color_stream = color_stream        # torch stream on which the SH parameters will be ready (or None)
        self.stream = torch.cuda.current_stream(device).cuda_stream
        self.kept = _Lease()        # geom / binning / image: travel with the autograd node
        self.scratch = _Lease()     # released by done()
        self.cb = _lib.ALLOC_FN(self._alloc)

    def _alloc(self, _ctx, which, nbytes):
        try:
            which, nbytes = int(which), max(int(nbytes), 1)
            if which == _lib.GSR_BUF_COLOR_STREAM:   # question, not an allocation: a second stream for the colour pass?
                if self.color_stream is None:
                    return 0
                self.before_color = None             # that stream is already ordered behind the parameter update
                STATS["color_pass_on_second_stream"] += 1
                return self.color_stream.cuda_stream
            if which == _lib.GSR_BUF_SYNC_SH:        # notification, not an allocation
                if self.before_color is not None:
                    hook, self.before_color = self.before_color, None
                    hook()
                return 1
            key = (self.device, self.stream, which)
            t = _POOL.take(key, nbytes, self.device)
            lease = self.scratch if which in (_lib.GSR_BUF_SCRATCH, _lib.GSR_BUF_SCRATCH2) else self.kept
            lease.items.append((key, t))
            self.buffers[which] = t
            return t.data_ptr()
        except Exception as e:  # surfaces as GSR_E_ALLOC
            self.error = e
            return 0

    def done(self):
        """The library call has returned: its scratch may be reused by the next call on this stream."""
        self.scratch.release()


# Set to True to backpropagate through one forward more than once (retain_graph=True): the buffers the
# forward saved then stay leased until the autograd node is garbage collected instead of going back
# to the pool right after the first backward.
KEEP_BUFFERS_AFTER_BACKWARD = False


def _check_lease(ctx):
    if ctx.lease.released:
        raise _lib.GsrError("the buffers saved by this forward were recycled after its first backward; set "
                            "gaussmart_amd.rasterizer.KEEP_BUFFERS_AFTER_BACKWARD = True to backpropagate twice")


def _finish_lease(ctx):
    # the autograd node often outlives backward() by several iterations (reference cycles through
    # retain_grad hooks are only broken by the cyclic GC), so the lease is returned here
    if not KEEP_BUFFERS_AFTER_BACKWARD:
        ctx.lease.release()


class ColorGradRecord:
    """Factored SH gradient of ONE backward (GSR_FLAG_FACTORED_SH_GRAD): instead of dL/d(features_dc) and
    dL/d(features_rest) -- 48 floats per Gaussian -- the backward leaves the clamp-masked colour gradient [N,3]
    followed by the camera position in `record` (f32[3N + 4]); gsr_adam_sh_factored rebuilds basis_k(dir) x g inside
    the optimiser step.  `flat` = [xyz | opacity | scaling | rotation | record] is one allocation, `head` its first
    four segments (what a view-parallel step all-reduces), `xyz` the positions the backward saw."""
    __slots__ = ("flat", "head", "record", "n", "sh_coeffs", "sh_degree", "xyz", "exchanged", "gathered", "n_views",
                 "grad_scale")

    def __init__(self, flat, head, record, n, sh_coeffs, sh_degree, xyz):
        self.flat, self.head, self.record, self.n = flat, head, record, int(n)
        self.sh_coeffs, self.sh_degree, self.xyz = int(sh_coeffs), int(sh_degree), xyz
        # filled by ViewParallel.exchange_factored: records of all ranks, their number, 1 / world when averaging
        self.exchanged, self.gathered, self.n_views, self.grad_scale = False, None, 1, 1.0


class RasterState:
    """The hand-over slots between the raw-parameter operator and whoever owns the optimiser step of ONE model.  One
    object per model (GaussianModel.raster_state), passed to rasterize_gaussians_raw(state=...): nothing is parked at module
    level, so two models on one device -- or two threads with a model each -- cannot see each other's slots
    (re-entrant, no global mutable state).

    `pending`: pipelined data-parallel step (view_parallel.py) -- the SH parameters may still be receiving their Adam
    update on a side stream when the next forward starts.  (event marking the end of that update, the stream it runs
    on): the next raw forward given this state lets its geometry / binning phase run, puts its SH colour pass on that
    stream too (GSR_FLAG_DEFER_COLOR + GSR_BUF_COLOR_STREAM), behind the update, or -- without a second stream -- makes
    its own stream wait right before the colour pass.  Every other consumer of the parameters calls wait_pending() first.
    `color_grad`: the ColorGradRecord the last factored backward of a forward given this state left."""
    __slots__ = ("pending", "color_grad")

    def __init__(self):
        self.pending = None
        self.color_grad = None

    def set_pending_param_event(self, event, stream=None):
        self.pending = (event, stream)

    def pop_pending(self):
        pend, self.pending = self.pending, None
        return pend

    def wait_pending(self, device):
        """Make the current stream of `device` wait for an outstanding side-stream parameter update (no-op if none)."""
        pend = self.pop_pending()
        if pend is not None:
            torch.cuda.current_stream(torch.device(device)).wait_event(pend[0])

    def take_color_grad(self):
        rec, self.color_grad = self.color_grad, None
        return rec


# Row-scan side job (include/gsr.h: GsrRowScanJob): the raw forward describes the scan its backward starts with and keeps
# the description ON ITS AUTOGRAD NODE; the fused objective, whose kernels run between the forward and the backward, finds
# it through the grad_fn of the image it was handed -- so a job only ever rides with an objective of the image THAT forward
# produced -- and carries the scan in extra workgroups of its own launches.  The job holds the forward's buffer lease until
# the rasterizer's backward has consumed it; that backward marks it dead, so a kernel that would write into BINNING after
# the buffers went back to the pool (an objective whose backward runs later, or twice) is never launched.
# GSR_ROW_SCAN_RIDE=0 switches the hand-over off.
_ROW_SCAN_RIDE = __import__("os").environ.get("GSR_ROW_SCAN_RIDE", "1") != "0"
# a backward that receives no gradient for allmap says so (GSR_FLAG_NO_SURFACE_GRAD); GSR_NO_SURFACE_FAST_PATH=0: A/B aid
_NO_SURFACE_FAST_PATH = __import__("os").environ.get("GSR_NO_SURFACE_FAST_PATH", "1") != "0"


def take_row_scan_job(image):
    """The row-scan job of the forward that produced `image` (one of its outputs, not a tensor derived from them), for a
    caller about to launch kernels on the CURRENT stream -- None if that forward offered none, if somebody already took
    it, if its backward already ran, or if the forward ran on another stream (the hand-over relies on stream order and
    nothing else)."""
    job = getattr(getattr(image, "grad_fn", None), "row_scan_job", None)
    if job is None or job._taken or job._dead:
        return None
    if job._stream != torch.cuda.current_stream(image.device).cuda_stream:
        return None
    job._taken = True
    return job


def row_scan_job_alive(job):
    """False once the rasterizer's backward has consumed the job (its buffers may be back in the pool)."""
    return job is not None and not job._dead


def release_workspace():
    """Drop every cached library buffer (e.g. before handing the GPU to something else)."""
    _POOL.clear()
    _ZERO_IMAGES.clear()


def _make_view(rs: GaussianRasterizationSettings, sh_coeffs: int, flags: int, device, channels: int = 3, keep=None):
    """GsrView + the tensors its pointers refer to.  `keep` = the tuple a previous call returned for the same
    settings (the backward reuses the forward's contiguous copies instead of making them again)."""
    if keep is not None:
        bg, vm, pm, cp = keep
    else:
        bg = _f32c(rs.bg, "bg", device)
        vm = _f32c(rs.viewmatrix, "viewmatrix", device)
        pm = _f32c(rs.projmatrix, "projmatrix", device)
        cp = _f32c(rs.campos, "campos", device)
    if bg.numel() != channels or vm.numel() != 16 or pm.numel() != 16 or cp.numel() != 3:
        raise ValueError(f"bg must have {channels} elements (one per output channel), campos 3, "
                         "viewmatrix/projmatrix 16")
    v = _lib.GsrView(int(rs.image_width), int(rs.image_height), float(rs.tanfovx), float(rs.tanfovy),
                     float(rs.scale_modifier), int(rs.sh_degree), int(sh_coeffs), int(channels), int(flags),
                     bg.data_ptr(), vm.data_ptr(), pm.data_ptr(), cp.data_ptr())
    return v, (bg, vm, pm, cp)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                cov3Ds_precomp, raster_settings, flags):
        L = _lib.lib()
        device = means3D.device
        if device.type != "cuda":
            raise _lib.GsrError("GaussianRasterizer needs tensors on a HIP device (torch 'cuda'); "
                                "there is no CPU path")
        rs = raster_settings
        N = means3D.shape[0]
        H, W = int(rs.image_height), int(rs.image_width)
        means3D = _f32c(means3D, "means3D", device)
        sh = _f32c(sh, "shs", device)
        colors_precomp = _f32c(colors_precomp, "colors_precomp", device)
        opacities = _f32c(opacities, "opacities", device)
        scales = _f32c(scales, "scales", device)
        rotations = _f32c(rotations, "rotations", device)
        cov3Ds_precomp = _f32c(cov3Ds_precomp, "cov3D_precomp", device)
        channels = 3
        if colors_precomp is not None:
            # [N,3] RGB, or a wide per-pixel payload [N,C] with C = 4, 8, ... 64 (semantic / feature splatting)
            if colors_precomp.dim() != 2 or colors_precomp.shape[0] != N:
                raise _lib.GsrError("colors_precomp must be [N,C]")
            channels = int(colors_precomp.shape[1])
        sh_coeffs = sh.shape[1] if sh is not None else 0

        with torch.cuda.device(device):
            view, keep = _make_view(rs, sh_coeffs, flags, device, channels)
            g = _lib.GsrGaussians(N, _ptr(means3D), _ptr(sh), _ptr(colors_precomp), _ptr(opacities),
                                  _ptr(scales), _ptr(rotations), _ptr(cov3Ds_precomp), None)
            color = torch.empty((channels, H, W), dtype=torch.float32, device=device)
            allmap = torch.empty((7, H, W), dtype=torch.float32, device=device)
            radii = torch.empty((N,), dtype=torch.int32, device=device)
            out = _lib.GsrForwardOut(color.data_ptr(), allmap.data_ptr(), radii.data_ptr(), 0, None, None, None)
            alloc = _Allocator(device)
            stream = torch.cuda.current_stream(device).cuda_stream
            rc = L.gsr_forward(C.byref(view), C.byref(g), C.byref(out), alloc.cb, None, C.c_void_p(stream))
            alloc.done()
            if rc != 0 and alloc.error is not None:
                raise alloc.error
            _lib.check(rc)

        ctx.lease = alloc.kept          # geom / binning / image go back to the pool with this node
        ctx.channels = channels
        ctx.raster_settings = rs
        ctx.flags = flags
        ctx.num_rendered = int(out.num_rendered)
        ctx.view_keep = keep
        ctx.set_materialize_grads(False)     # no zero tensors for the unused radii / image gradients
        ctx.none_mask = (sh is None, colors_precomp is None, scales is None, cov3Ds_precomp is None)
        geom = alloc.buffers[_lib.GSR_BUF_GEOM]
        binning = alloc.buffers[_lib.GSR_BUF_BINNING]
        image = alloc.buffers[_lib.GSR_BUF_IMAGE]
        empty = torch.empty(0, device=device)
        ctx.save_for_backward(means3D, sh if sh is not None else empty,
                              colors_precomp if colors_precomp is not None else empty, opacities,
                              scales if scales is not None else empty,
                              rotations if rotations is not None else empty,
                              cov3Ds_precomp if cov3Ds_precomp is not None else empty,
                              radii, geom, binning, image)
        ctx.mark_non_differentiable(radii)
        return color, radii, allmap

    @staticmethod
    def backward(ctx, grad_color, _grad_radii, grad_allmap):
        L = _lib.lib()
        _check_lease(ctx)
        (means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, radii, geom,
         binning, image) = ctx.saved_tensors
        no_sh, no_col, no_sr, no_cov = ctx.none_mask
        sh = None if no_sh else sh
        colors_precomp = None if no_col else colors_precomp
        scales = None if no_sr else scales
        rotations = None if no_sr else rotations
        cov3Ds_precomp = None if no_cov else cov3Ds_precomp
        rs = ctx.raster_settings
        device = means3D.device
        N = means3D.shape[0]
        H, W = int(rs.image_height), int(rs.image_width)
        grad_color = _f32c(grad_color, "grad_color", device) if grad_color is not None else \
            _zero_image(ctx.channels, H, W, device)
        # nothing was differentiated through allmap (e.g. lambda_normal = lambda_dist = 0): the backward is told so
        bwd_flags = ctx.flags | (_lib.GSR_FLAG_NO_SURFACE_GRAD if (grad_allmap is None and _NO_SURFACE_FAST_PATH) else 0)
        grad_allmap = _f32c(grad_allmap, "grad_allmap", device) if grad_allmap is not None else \
            _zero_image(7, H, W, device)

        with torch.cuda.device(device):
            view, keep = _make_view(rs, sh.shape[1] if sh is not None else 0, bwd_flags, device, ctx.channels, ctx.view_keep)
            g = _lib.GsrGaussians(N, _ptr(means3D), _ptr(sh), _ptr(colors_precomp), _ptr(opacities),
                                  _ptr(scales), _ptr(rotations), _ptr(cov3Ds_precomp), None)
            d_means3D = torch.empty_like(means3D)
            d_means2D = torch.empty((N, 3), dtype=torch.float32, device=device)
            d_opac = torch.empty_like(opacities)
            d_sh = torch.empty_like(sh) if sh is not None else None
            d_col = torch.empty_like(colors_precomp) if colors_precomp is not None else None
            d_scales = torch.empty_like(scales) if scales is not None else None
            d_rot = torch.empty_like(rotations) if rotations is not None else None
            d_cov = torch.empty_like(cov3Ds_precomp) if cov3Ds_precomp is not None else None
            grads = _lib.GsrGrads(_ptr(d_means3D), _ptr(d_means2D), _ptr(d_opac), _ptr(d_sh), _ptr(d_col),
                                  _ptr(d_scales), _ptr(d_rot), _ptr(d_cov), None)
            alloc = _Allocator(device)
            stream = torch.cuda.current_stream(device).cuda_stream
            rc = L.gsr_backward(C.byref(view), C.byref(g), ctx.num_rendered, _ptr(radii), _ptr(geom),
                                _ptr(binning), _ptr(image), _ptr(grad_color), _ptr(grad_allmap),
                                C.byref(grads), alloc.cb, None, C.c_void_p(stream))
            alloc.done()
            if rc != 0 and alloc.error is not None:
                raise alloc.error
            _lib.check(rc)
        del keep
        _finish_lease(ctx)
        return (d_means3D, d_means2D, d_sh, d_col, d_opac, d_scales, d_rot, d_cov, None, None)


class _RasterizeGaussiansRaw(torch.autograd.Function):
    """Same kernels, fed with the model's RAW parameters (GSR_FLAG_RAW_PARAMS + split SH storage):
    sigmoid / exp / normalize of scene/gaussian_model.py:37-43 and the dc|rest concatenation of
    :116-119 happen inside preprocess_fwd, their derivatives inside preprocess_bwd.  Saves ~20
    element-wise launches and two 192 MB copies per iteration; results equal the activated path."""

    @staticmethod
    def forward(ctx, xyz, means2D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, raster_settings, flags,
                color_cache=None, state=None, color_only=False, no_dist_median=False):
        L = _lib.lib()
        device = xyz.device
        if device.type != "cuda":
            raise _lib.GsrError("GaussianRasterizer needs tensors on a HIP device (torch 'cuda'); there is no CPU path")
        rs = raster_settings
        N = xyz.shape[0]
        H, W = int(rs.image_height), int(rs.image_width)
        xyz, f_dc, f_rest = _f32c(xyz, "xyz", device), _f32c(f_dc, "features_dc", device), _f32c(f_rest, "features_rest", device)
        opacity_raw, scaling_raw = _f32c(opacity_raw, "opacity", device), _f32c(scaling_raw, "scaling", device)
        rotation_raw = _f32c(rotation_raw, "rotation", device)
        if f_dc.dim() != 3 or f_dc.shape[1] != 1 or f_rest.dim() != 3 or f_rest.shape[0] != N:
            raise ValueError("features_dc must be [N,1,3] and features_rest [N,K-1,3]")
        M = 1 + f_rest.shape[1]
        flags = int(flags) | _lib.GSR_FLAG_RAW_PARAMS
        if color_only:                    # the caller does not consume allmap (no regularizer active): not accumulated, not written
            flags |= _lib.GSR_FLAG_COLOR_ONLY
        elif no_dist_median:              # ... or not its distortion / median-depth channels (lambda_dist = 0, depth_ratio = 0)
            flags |= _lib.GSR_FLAG_NO_DIST_MEDIAN
        if color_cache is not None:       # the SH colour of this view was left by the optimiser step (FusedAdam.color_cache)
            if color_cache.numel() != 13 * N or color_cache.dtype != torch.float32 or color_cache.device != device:
                raise ValueError("color_cache must be float32 [13 N] on the parameters' device")
            flags |= _lib.GSR_FLAG_COLOR_CACHED
        pending = state.pop_pending() if state is not None else None
        hook = color_stream = None
        if pending is not None:
            flags |= _lib.GSR_FLAG_DEFER_COLOR
            pending_event, color_stream = pending
            if color_stream is not None and color_stream == torch.cuda.current_stream(device):
                color_stream = None
            hook = lambda: torch.cuda.current_stream(device).wait_event(pending_event)
        rest = f_rest if f_rest.shape[1] > 0 else None
        with torch.cuda.device(device):
            view, keep = _make_view(rs, M, flags, device)
            g = _lib.GsrGaussians(N, _ptr(xyz), _ptr(f_dc), _ptr(color_cache), _ptr(opacity_raw), _ptr(scaling_raw),
                                  _ptr(rotation_raw), None, _ptr(rest) if rest is not None else None)
            color = torch.empty((3, H, W), dtype=torch.float32, device=device)
            allmap = torch.empty((7, H, W), dtype=torch.float32, device=device)
            radii = torch.empty((N,), dtype=torch.int32, device=device)
            out = _lib.GsrForwardOut(color.data_ptr(), allmap.data_ptr(), radii.data_ptr(), 0, None, None, None)
            alloc = _Allocator(device, before_color=hook, color_stream=color_stream)
            stream = torch.cuda.current_stream(device).cuda_stream
            rc = L.gsr_forward(C.byref(view), C.byref(g), C.byref(out), alloc.cb, None, C.c_void_p(stream))
            alloc.done()
            if alloc.before_color is not None:       # the library returned before announcing the colour pass
                alloc.before_color()
            if rc != 0 and alloc.error is not None:
                raise alloc.error
            _lib.check(rc)
        ctx.lease = alloc.kept
        ctx.raster_settings, ctx.flags, ctx.num_rendered, ctx.M = rs, flags & ~_lib.GSR_FLAG_DEFER_COLOR, int(out.num_rendered), M
        ctx.view_keep = keep
        ctx.color_cache = color_cache        # (the backward reads d(rgb)/d(dir) from it)
        ctx.state = state                    # (the backward leaves the factored SH gradient there)
        ctx.row_scan_job = None
        if _ROW_SCAN_RIDE and int(out.num_rendered) > 0:
            job = _lib.GsrRowScanJob()
            _lib.check(L.gsr_row_scan_job(C.c_void_p(alloc.buffers[_lib.GSR_BUF_BINNING].data_ptr()), int(out.num_rendered),
                                          W, H, C.byref(job)))
            job._lease = alloc.kept            # keeps BINNING out of the pool while anybody may still write into it
            job._stream = stream               # the hand-over is ordered by this stream and nothing else
            job._taken = job._dead = False
            ctx.row_scan_job = job             # found by the objective through image.grad_fn (take_row_scan_job)
        ctx.set_materialize_grads(False)     # no zero tensors for the unused radii / image gradients
        ctx.save_for_backward(xyz, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, radii,
                              alloc.buffers[_lib.GSR_BUF_GEOM], alloc.buffers[_lib.GSR_BUF_BINNING],
                              alloc.buffers[_lib.GSR_BUF_IMAGE])
        del keep
        if color_only:
            ctx.mark_non_differentiable(radii, allmap)       # (allmap was not written: rasterize_gaussians_raw hands back None)
        else:
            ctx.mark_non_differentiable(radii)
        return color, radii, allmap

    @staticmethod
    def backward(ctx, grad_color, _grad_radii, grad_allmap):
        L = _lib.lib()
        _check_lease(ctx)
        xyz, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, radii, geom, binning, image = ctx.saved_tensors
        if ctx.flags & _lib.GSR_FLAG_COLOR_ONLY:
            grad_allmap = None
        rs = ctx.raster_settings
        device = xyz.device
        N = xyz.shape[0]
        H, W = int(rs.image_height), int(rs.image_width)
        grad_color = _f32c(grad_color, "grad_color", device) if grad_color is not None else _zero_image(3, H, W, device)
        # nothing was differentiated through allmap (the first 7,000 iterations, or lambda_normal = lambda_dist = 0)
        bwd_flags = ctx.flags | (_lib.GSR_FLAG_NO_SURFACE_GRAD if (grad_allmap is None and (
            _NO_SURFACE_FAST_PATH or ctx.flags & _lib.GSR_FLAG_COLOR_ONLY)) else 0)
        grad_allmap = _f32c(grad_allmap, "grad_allmap", device) if grad_allmap is not None else _zero_image(7, H, W, device)
        rest = f_rest if f_rest.shape[1] > 0 else None
        with torch.cuda.device(device):
            view, keep = _make_view(rs, ctx.M, bwd_flags, device, 3, ctx.view_keep)
            g = _lib.GsrGaussians(N, _ptr(xyz), _ptr(f_dc), _ptr(ctx.color_cache), _ptr(opacity_raw), _ptr(scaling_raw),
                                  _ptr(rotation_raw), None, _ptr(rest) if rest is not None else None)
            # ONE buffer for the six parameter gradients, [xyz | f_dc | opacity | scaling | rotation | f_rest]: the
            # data-parallel step all-reduces it as a whole (or as "geometry + dc" / "rest" halves) without copies
            d_2d = torch.empty((N, 3), dtype=torch.float32, device=device)
            factored = bool(ctx.flags & _lib.GSR_FLAG_FACTORED_SH_GRAD)
            srcs = (xyz, opacity_raw, scaling_raw, rotation_raw) if factored else \
                (xyz, f_dc, opacity_raw, scaling_raw, rotation_raw, f_rest)
            offs, total = [], 0
            for t in srcs:                       # every segment starts 16-byte aligned (K8 stores float4)
                offs.append(total)
                total += _round_up(t.numel(), 4)
            if factored:
                # [xyz | opacity | scaling | rotation | colour gradient [N,3] + camera position]: 13 floats per Gaussian
                n_head = total
                flat = torch.empty(n_head + 3 * N + 4, dtype=torch.float32, device=device)
                d_xyz, d_op, d_sc, d_rot = [flat[o:o + t.numel()].view_as(t) for o, t in zip(offs, srcs)]
                d_dc = d_rest = None
                record = flat[n_head:]
                grads = _lib.GsrGrads(_ptr(d_xyz), _ptr(d_2d), _ptr(d_op), None, _ptr(record), _ptr(d_sc), _ptr(d_rot),
                                      None, None)
            else:
                flat = torch.empty(total, dtype=torch.float32, device=device)
                d_xyz, d_dc, d_op, d_sc, d_rot, d_rest = [flat[o:o + t.numel()].view_as(t) for o, t in zip(offs, srcs)]
                grads = _lib.GsrGrads(_ptr(d_xyz), _ptr(d_2d), _ptr(d_op), _ptr(d_dc), None, _ptr(d_sc), _ptr(d_rot), None,
                                      _ptr(d_rest) if rest is not None else None)
            alloc = _Allocator(device)
            stream = torch.cuda.current_stream(device).cuda_stream
            job = getattr(ctx, "row_scan_job", None)
            if job is not None:
                # whatever happens below, nobody may enqueue a half of this job any more: after this backward the buffers
                # go back to the pool (an objective holding the job checks row_scan_job_alive before it launches)
                usable = not job._dead and job._stream == stream      # else: not the stream the hand-over was ordered by
                job._dead = True
                job._lease = None
                if not usable:
                    job = None
            if job is not None:
                rc = L.gsr_backward_with_job(C.byref(view), C.byref(g), ctx.num_rendered, _ptr(radii), _ptr(geom),
                                             _ptr(binning), _ptr(image), _ptr(grad_color), _ptr(grad_allmap), C.byref(grads),
                                             C.byref(job), alloc.cb, None, C.c_void_p(stream))
                STATS["row_scans_carried"] += int(job.stage == 2)
            else:
                rc = L.gsr_backward(C.byref(view), C.byref(g), ctx.num_rendered, _ptr(radii), _ptr(geom), _ptr(binning),
                                    _ptr(image), _ptr(grad_color), _ptr(grad_allmap), C.byref(grads), alloc.cb, None,
                                    C.c_void_p(stream))
            alloc.done()
            if rc != 0 and alloc.error is not None:
                raise alloc.error
            _lib.check(rc)
        del keep
        _finish_lease(ctx)
        if factored:
            ctx.state.color_grad = ColorGradRecord(flat, flat[:n_head], record, N, ctx.M, rs.sh_degree, xyz)
        return d_xyz, d_2d, d_dc, d_rest, d_op, d_sc, d_rot, None, None, None, None, None, None


def _wants_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def _forward_only(device, rs, flags, sh_coeffs, gaussians_args, N):
    """Inference path (render.py / view.py / GaussianExtractor under torch.no_grad(): utils/mesh_utils.py:100-123,
    view.py:15-31): GSR_FLAG_FORWARD_ONLY -- no touch words, no per-pixel state, no autograd node, and every library
    buffer goes back to the pool as soon as the call has been enqueued (stream-ordered reuse)."""
    L = _lib.lib()
    if device.type != "cuda":
        raise _lib.GsrError("GaussianRasterizer needs tensors on a HIP device (torch 'cuda'); there is no CPU path")
    H, W = int(rs.image_height), int(rs.image_width)
    with torch.cuda.device(device):
        view, keep = _make_view(rs, sh_coeffs, int(flags) | _lib.GSR_FLAG_FORWARD_ONLY, device)
        g = _lib.GsrGaussians(N, *[_ptr(a) for a in gaussians_args])
        color = torch.empty((3, H, W), dtype=torch.float32, device=device)
        allmap = torch.empty((7, H, W), dtype=torch.float32, device=device)
        radii = torch.empty((N,), dtype=torch.int32, device=device)
        out = _lib.GsrForwardOut(color.data_ptr(), allmap.data_ptr(), radii.data_ptr(), 0, None, None, None)
        alloc = _Allocator(device)
        stream = torch.cuda.current_stream(device).cuda_stream
        rc = L.gsr_forward(C.byref(view), C.byref(g), C.byref(out), alloc.cb, None, C.c_void_p(stream))
        alloc.done()
        alloc.kept.release()
        if rc != 0 and alloc.error is not None:
            raise alloc.error
        _lib.check(rc)
    del keep
    return color, radii, allmap


def rasterize_gaussians_raw(xyz, means2D, features_dc, features_rest, opacity_raw, scaling_raw, rotation_raw,
                            raster_settings, flags=None, factored_sh_grad=False, color_cache=None, state=None, color_only=False,
                            no_dist_median=False):
    """(color, radii, allmap) from the model's raw parameter tensors; activations fused in-kernel.
    `state`: the model's RasterState (hand-over slots with the optimiser step); None for a caller that has neither a
    pipelined update in flight nor a factored gradient to receive.
    `factored_sh_grad`: the backward leaves NO gradient on features_dc / features_rest; it leaves a ColorGradRecord in
    `state.color_grad` (RasterState.take_color_grad) for FusedAdam.step_sh_factored instead -- only for callers that own
    the optimiser step, and only with a `state` to leave it in.
    `color_only`: the caller consumes the colour image alone (a training step with no regularizer active): allmap comes back
    as None -- the forward neither accumulates nor writes it (GSR_FLAG_COLOR_ONLY), the backward runs without its terms.
    `no_dist_median`: the caller consumes neither the distortion nor the median-depth channel of allmap (lambda_dist = 0 and
    depth_ratio = 0, the reference's defaults): both come back as zeros (GSR_FLAG_NO_DIST_MEDIAN)."""
    flags = DEFAULT_FLAGS if flags is None else flags
    if factored_sh_grad and state is None:
        raise ValueError("factored_sh_grad needs state=RasterState(): the backward leaves its colour-gradient record there")
    if not _wants_grad(xyz, means2D, features_dc, features_rest, opacity_raw, scaling_raw, rotation_raw):
        device = xyz.device
        if state is not None:
            state.wait_pending(device)
        args = [_f32c(t, n, device) for t, n in ((xyz, "xyz"), (features_dc, "features_dc"), (None, ""), (opacity_raw, "opacity"),
                                                 (scaling_raw, "scaling"), (rotation_raw, "rotation"), (None, ""))]
        rest = _f32c(features_rest, "features_rest", device)
        if args[1].dim() != 3 or args[1].shape[1] != 1 or rest.dim() != 3 or rest.shape[0] != xyz.shape[0]:
            raise ValueError("features_dc must be [N,1,3] and features_rest [N,K-1,3]")
        args.append(rest if rest.shape[1] > 0 else None)
        return _forward_only(device, raster_settings, int(flags) | _lib.GSR_FLAG_RAW_PARAMS, 1 + rest.shape[1], args,
                             xyz.shape[0])
    if factored_sh_grad:
        flags |= _lib.GSR_FLAG_FACTORED_SH_GRAD
    color, radii, allmap = _RasterizeGaussiansRaw.apply(xyz, means2D, features_dc, features_rest, opacity_raw, scaling_raw,
                                                        rotation_raw, raster_settings, flags,
                                                        color_cache if factored_sh_grad else None, state, bool(color_only),
                                                        bool(no_dist_median))
    return color, radii, (None if color_only else allmap)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                        cov3Ds_precomp, raster_settings, flags=None):
    wide = colors_precomp is not None and colors_precomp.dim() == 2 and colors_precomp.shape[1] != 3
    if not wide and not _wants_grad(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp):
        device = means3D.device
        args = [_f32c(t, n, device) for t, n in ((means3D, "means3D"), (sh, "shs"), (colors_precomp, "colors_precomp"),
                                                 (opacities, "opacities"), (scales, "scales"), (rotations, "rotations"),
                                                 (cov3Ds_precomp, "cov3D_precomp"))] + [None]
        if args[2] is not None and (args[2].dim() != 2 or args[2].shape[0] != means3D.shape[0]):
            raise _lib.GsrError("colors_precomp must be [N,C]")
        return _forward_only(device, raster_settings, DEFAULT_FLAGS if flags is None else flags,
                             sh.shape[1] if sh is not None else 0, args, means3D.shape[0])
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales,
                                     rotations, cov3Ds_precomp, raster_settings,
                                     DEFAULT_FLAGS if flags is None else flags)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings, flags=None):
        super().__init__()
        self.raster_settings = raster_settings
        self.flags = flags

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """Frustum test used by callers that pre-filter: view-space z beyond the near plane [U]."""
        with torch.no_grad():
            V = self.raster_settings.viewmatrix
            z = positions @ V[:3, 2] + V[3, 2]
            return z > 0.2

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None,
                rotations=None, cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, self.raster_settings, self.flags)


def rasterize_debug(means3D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                    cov3D_precomp=None, *, raster_settings, flags=None):
    """Forward only, returning the operator outputs AND typed views of every saved buffer field
    (include/gsr.h: gsr_buffer_field).  Used by the parity tests; no autograd."""
    L = _lib.lib()
    device = means3D.device
    rs = raster_settings
    flags = DEFAULT_FLAGS if flags is None else flags
    N = means3D.shape[0]
    H, W = int(rs.image_height), int(rs.image_width)
    args = [_f32c(t, n, device) for t, n in ((means3D, "means3D"), (shs, "shs"), (colors_precomp, "colors"),
                                             (opacities, "opacities"), (scales, "scales"),
                                             (rotations, "rotations"), (cov3D_precomp, "cov3D"))]
    with torch.cuda.device(device), torch.no_grad():
        view, keep = _make_view(rs, args[1].shape[1] if args[1] is not None else 0, flags, device)
        g = _lib.GsrGaussians(N, *[_ptr(a) for a in args], None)
        color = torch.empty((3, H, W), dtype=torch.float32, device=device)
        allmap = torch.empty((7, H, W), dtype=torch.float32, device=device)
        radii = torch.empty((N,), dtype=torch.int32, device=device)
        out = _lib.GsrForwardOut(color.data_ptr(), allmap.data_ptr(), radii.data_ptr(), 0, None, None, None)
        alloc = _Allocator(device)
        stream = torch.cuda.current_stream(device).cuda_stream
        rc = L.gsr_forward(C.byref(view), C.byref(g), C.byref(out), alloc.cb, None, C.c_void_p(stream))
        if rc != 0 and alloc.error is not None:
            raise alloc.error
        _lib.check(rc)
        torch.cuda.synchronize(device)
    D = int(out.num_rendered)
    # the views below alias pooled buffers: the lease keeps them out of the pool while `res` lives
    res = dict(color=color, allmap=allmap, radii=radii, num_rendered=D, _lease=alloc.kept)
    spec = {
        _lib.GSR_BUF_GEOM: [("splat", torch.float32, (N, 20)), ("clamped", torch.int32, (N,)),
                            ("tiles_touched", torch.int32, (N,)), ("depth_key", torch.int32, (N,)),
                            ("order", torch.int32, (N,)), ("offs", torch.int32, (N + 1,))],
        _lib.GSR_BUF_BINNING: [("point_list", torch.int32, (D,)), ("inst_row", torch.int32, (D,)),
                               ("ranges", torch.int32, (-1, 2)), ("covered", torch.int32, (-1, 4)),
                               ("touch", torch.int32, (D,)), ("row_count", torch.uint8, (D,))],
        _lib.GSR_BUF_IMAGE: [("final_T", torch.float32, (3, H, W)), ("n_contrib", torch.int32, (2, H, W))],
    }
    for which, fields in spec.items():
        buf = alloc.buffers[which]
        for name, dt, shape in fields:
            try:
                off, nbytes = _lib.buffer_field(which, name, N, D, W, H)
            except _lib.GsrError:
                if name not in ("covered", "touch", "row_count"):   # (an older library build, loaded through GSR_LIB_PATH for an A/B run)
                    raise
                continue
            res[name] = buf[off:off + nbytes].view(dt).reshape(shape) if nbytes else \
                torch.empty(0, dtype=dt, device=device).reshape([s if s >= 0 else 0 for s in shape])
    # first emission index of every Gaussian (depth rank r = order^-1): offs is indexed by rank
    res["inst_begin"] = torch.zeros(N, dtype=torch.int32, device=device)
    res["inst_begin"][res["order"].long()] = res["offs"][:N]
    del keep
    return res
