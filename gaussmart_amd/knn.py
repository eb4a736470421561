"""`simple_knn._C.distCUDA2` on top of libgsr_hip.so (call site scene/gaussian_model.py:22,261)."""
import ctypes as C

import torch

from . import _lib


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """points f32 [N,3] on a HIP device -> f32 [N]: mean squared distance to the 3 nearest other
    points."""
    L = _lib.lib()
    if points.device.type != "cuda":
        raise _lib.GsrError("distCUDA2 needs a tensor on a HIP device (torch 'cuda'); there is no CPU path")
    if points.dim() != 2 or points.shape[1] != 3:
        raise ValueError("points must be [N,3]")
    pts = points.detach().float().contiguous()
    n = pts.shape[0]
    out = torch.empty((n,), dtype=torch.float32, device=pts.device)
    if n == 0:
        return out
    with torch.cuda.device(pts.device):
        nbytes = L.gsr_knn3_workspace_bytes(n)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=pts.device)
        stream = torch.cuda.current_stream(pts.device).cuda_stream
        _lib.check(L.gsr_knn3(C.c_void_p(pts.data_ptr()), n, C.c_void_p(out.data_ptr()),
                              C.c_void_p(ws.data_ptr()), nbytes, C.c_void_p(stream)))
    return out


def sort_pairs_u32(keys: torch.Tensor, vals, begin_bit=0, end_bit=32):
    """Stable LSD radix sort of (key, value) pairs (int32 tensors viewed as u32); vals=None sorts
    indices.  Exposed for the bit-exactness tests of the binning sort."""
    L = _lib.lib()
    n = keys.numel()
    k = keys.contiguous()
    v = vals.contiguous() if vals is not None else None
    ko = torch.empty_like(k)
    vo = torch.empty_like(k)
    if n == 0:
        return ko, vo
    with torch.cuda.device(k.device):
        nbytes = L.gsr_sort_workspace_bytes(n)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=k.device)
        stream = torch.cuda.current_stream(k.device).cuda_stream
        _lib.check(L.gsr_sort_pairs_u32(C.c_void_p(k.data_ptr()), C.c_void_p(v.data_ptr()) if v is not None else None,
                                        C.c_void_p(ko.data_ptr()), C.c_void_p(vo.data_ptr()), n, begin_bit,
                                        end_bit, C.c_void_p(ws.data_ptr()), nbytes, C.c_void_p(stream)))
    return ko, vo
