"""Row compaction of the Gaussian SoA on the HIP library.

`compact_rows(tensors, keep)` == `[t[keep] for t in tensors]` for tensors that share their first dimension, with ONE
host synchronisation (the kept count) and ONE copy launch for all of them instead of a nonzero() + index kernel per
tensor -- what pruning does to the six parameters, their twelve Adam moments and the densification statistics
(scene/gaussian_model.py:398-470).  CPU tensors fall back to torch indexing (host-side plumbing tests)."""
import ctypes as C

import torch

from . import _lib

MAX_TENSORS = 24


def compact_rows(tensors, keep):
    tensors = list(tensors)
    if not tensors:
        return []
    n = tensors[0].shape[0]
    if keep.dtype != torch.bool or keep.dim() != 1 or keep.shape[0] != n:
        raise ValueError("keep must be a bool vector with one entry per row")
    usable = keep.is_cuda and all(t.is_cuda and t.shape[0] == n and t.is_contiguous() and t.element_size() * t[0].numel() % 4 == 0
                                  and t[0].numel() > 0 for t in tensors) and n > 0
    if not usable:
        return [t[keep] for t in tensors]
    L = _lib.lib()
    dev = keep.device
    keep = keep.contiguous()
    out = []
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        ws = torch.empty(L.gsr_compact_workspace_bytes(n), dtype=torch.uint8, device=dev)
        offsets = C.c_void_p()
        _lib.check(L.gsr_compact_plan(C.c_void_p(keep.data_ptr()), n, C.c_void_p(ws.data_ptr()), ws.numel(),
                                      C.byref(offsets), stream))
        off_t = ws[offsets.value - ws.data_ptr():].view(torch.int32)
        kept = int(off_t[n].item())                     # the one host synchronisation
        out = [torch.empty((kept,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev) for t in tensors]
        if kept > 0:
            for i in range(0, len(tensors), MAX_TENSORS):
                src, dst = tensors[i:i + MAX_TENSORS], out[i:i + MAX_TENSORS]
                k = len(src)
                _lib.check(L.gsr_compact_apply(
                    k, (C.c_void_p * k)(*[t.data_ptr() for t in src]), (C.c_void_p * k)(*[t.data_ptr() for t in dst]),
                    (C.c_int32 * k)(*[t.element_size() * t[0].numel() for t in src]), n, C.c_void_p(keep.data_ptr()),
                    offsets, stream))
    return out
