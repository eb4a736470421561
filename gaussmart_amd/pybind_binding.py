"""The reference operator bound through pybind11 instead of ctypes: `GaussianRasterizer` with the reference's call
signature (gaussian_renderer/__init__.py:97-106) on top of gaussmart_amd/csrc/pybind_shim.cpp -> include/gsr.h.

This is the binding shape north_star words ("thin C++/pybind11 C-ABI extension") and upstream's own `_C` module has: the
extension marshals pointers and sizes, every buffer -- outputs, the saved geometry / binning / image state, scratch -- is a
torch tensor allocated HERE, handed to the library through the allocator callback.  It drives exactly the same entry points
as the ctypes binding (gaussmart_amd/rasterizer.py), which stays the default because it also carries the optional
extensions (raw parameters, colour cache, pooled buffers); tests/test_pybind_binding.py checks both produce the same bits.
There is no CPU path: the extension links libgsr_hip.so and the operator refuses host tensors.
"""
import importlib
import os
import sys

import torch

from . import _lib
from .rasterizer import GaussianRasterizationSettings

_HERE = os.path.dirname(os.path.abspath(__file__))
_MOD = None


def module():
    """The compiled extension (gaussmart_amd/lib/_gsr_pybind*.so, built by `make -C gaussmart_amd/csrc`)."""
    global _MOD
    if _MOD is None:
        _lib.lib()                      # torch's HIP runtime and libgsr_hip.so first (same load order as the ctypes path)
        libdir = os.path.join(_HERE, "lib")
        if libdir not in sys.path:
            sys.path.insert(0, libdir)
        try:
            _MOD = importlib.import_module("_gsr_pybind")
        except ImportError as e:
            raise _lib.GsrError(f"the pybind11 binding is not built (make -C gaussmart_amd/csrc): {e}") from e
        if _MOD.abi_version() != _lib.ABI_VERSION:
            raise _lib.GsrError(f"_gsr_pybind reports ABI {_MOD.abi_version()}, expected {_lib.ABI_VERSION}")
    return _MOD


def _p(t):
    return None if t is None else int(t.data_ptr())


def _c(t, device):
    if t is None:
        return None
    if t.device != device:
        raise _lib.GsrError("all operator inputs must live on the same HIP device")
    return t.detach().to(torch.float32).contiguous()


class _Buffers:
    """Allocator callback of one library call: plain torch tensors, kept by kind."""

    def __init__(self, device):
        self.device, self.by_kind = device, {}

    def __call__(self, which, nbytes):
        if which >= 100:                 # GSR_BUF_SYNC_SH / GSR_BUF_COLOR_STREAM: only with GSR_FLAG_DEFER_COLOR, not used here
            return None
        t = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=self.device)
        self.by_kind[int(which)] = t
        return int(t.data_ptr())


def _view_dict(rs, sh_coeffs, channels, flags, device):
    keep = [_c(torch.as_tensor(x, dtype=torch.float32, device=device), device).reshape(-1)
            for x in (rs.bg, rs.viewmatrix, rs.projmatrix, rs.campos)]
    if keep[0].numel() != channels or keep[1].numel() != 16 or keep[2].numel() != 16 or keep[3].numel() != 3:
        raise ValueError(f"bg must have {channels} elements, campos 3, viewmatrix / projmatrix 16")
    d = dict(width=int(rs.image_width), height=int(rs.image_height), tanfovx=float(rs.tanfovx), tanfovy=float(rs.tanfovy),
             scale_modifier=float(rs.scale_modifier), sh_degree=int(rs.sh_degree), sh_coeffs=int(sh_coeffs),
             channels=int(channels), flags=int(flags), bg=_p(keep[0]), viewmatrix=_p(keep[1]), projmatrix=_p(keep[2]),
             campos=_p(keep[3]))
    return d, keep


class _Rasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs, flags):
        B = module()
        device = means3D.device
        if device.type != "cuda":
            raise _lib.GsrError("GaussianRasterizer needs tensors on a HIP device (torch 'cuda'); there is no CPU path")
        N, H, W = means3D.shape[0], int(rs.image_height), int(rs.image_width)
        means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp = \
            [_c(t, device) for t in (means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp)]
        channels = 3 if colors_precomp is None else int(colors_precomp.shape[1])
        with torch.cuda.device(device):
            view, keep = _view_dict(rs, sh.shape[1] if sh is not None else 0, channels, flags, device)
            g = dict(count=N, means3D=_p(means3D), shs=_p(sh), colors_precomp=_p(colors_precomp), opacities=_p(opacities),
                     scales=_p(scales), rotations=_p(rotations), transmat_precomp=_p(cov3Ds_precomp))
            color = torch.empty((channels, H, W), dtype=torch.float32, device=device)
            allmap = torch.empty((7, H, W), dtype=torch.float32, device=device)
            radii = torch.empty((N,), dtype=torch.int32, device=device)
            bufs = _Buffers(device)
            num_rendered, _, _, _ = B.forward(view, g, _p(color), _p(allmap), _p(radii), bufs,
                                              torch.cuda.current_stream(device).cuda_stream)
        ctx.rs, ctx.flags, ctx.channels, ctx.num_rendered = rs, flags, channels, int(num_rendered)
        ctx.inputs = (means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp)
        ctx.state = (radii, bufs.by_kind[_lib.GSR_BUF_GEOM], bufs.by_kind[_lib.GSR_BUF_BINNING], bufs.by_kind[_lib.GSR_BUF_IMAGE])
        ctx.keep = keep
        ctx.mark_non_differentiable(radii)
        return color, radii, allmap

    @staticmethod
    def backward(ctx, grad_color, _grad_radii, grad_allmap):
        B = module()
        means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp = ctx.inputs
        radii, geom, binning, image = ctx.state
        rs, device = ctx.rs, means3D.device
        N, H, W = means3D.shape[0], int(rs.image_height), int(rs.image_width)
        grad_color = _c(grad_color, device) if grad_color is not None else torch.zeros((ctx.channels, H, W), device=device)
        grad_allmap = _c(grad_allmap, device) if grad_allmap is not None else torch.zeros((7, H, W), device=device)
        like = lambda t: None if t is None else torch.empty_like(t)
        d = dict(dL_dmeans3D=like(means3D), dL_dmeans2D=torch.empty((N, 3), dtype=torch.float32, device=device),
                 dL_dopacity=like(opacities), dL_dshs=like(sh), dL_dcolors=like(colors_precomp), dL_dscales=like(scales),
                 dL_drotations=like(rotations), dL_dtransmat=like(cov3Ds_precomp))
        with torch.cuda.device(device):
            view, keep = _view_dict(rs, sh.shape[1] if sh is not None else 0, ctx.channels, ctx.flags, device)
            g = dict(count=N, means3D=_p(means3D), shs=_p(sh), colors_precomp=_p(colors_precomp), opacities=_p(opacities),
                     scales=_p(scales), rotations=_p(rotations), transmat_precomp=_p(cov3Ds_precomp))
            B.backward(view, g, ctx.num_rendered, _p(radii), _p(geom), _p(binning), _p(image), _p(grad_color), _p(grad_allmap),
                       {k: _p(v) for k, v in d.items()}, _Buffers(device), torch.cuda.current_stream(device).cuda_stream)
        del keep
        return (d["dL_dmeans3D"], d["dL_dmeans2D"], d["dL_dshs"], d["dL_dcolors"], d["dL_dopacity"], d["dL_dscales"],
                d["dL_drotations"], d["dL_dtransmat"], None, None)


class GaussianRasterizer(torch.nn.Module):
    """Same constructor and call signature as the reference's operator; argument checks mirror upstream's."""

    def __init__(self, raster_settings: GaussianRasterizationSettings, flags=_lib.GSR_FLAGS_UPSTREAM):
        super().__init__()
        self.raster_settings, self.flags = raster_settings, int(flags)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None) == (colors_precomp is None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        return _Rasterize.apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                self.raster_settings, self.flags)
