"""The C-ABI library loads on a machine without a GPU and exports every symbol include/gsr.h
declares; host-only entry points behave; the Python operators refuse CPU tensors loudly."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT
from gaussmart_amd import _lib


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "gsr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gsr_[a-z0-9_]+)\s*\(", text)) - {"gsr_alloc_fn"})


def test_library_exports_every_declared_symbol():
    names = _declared_functions()
    assert len(names) >= 12
    L = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/gsr.h but not exported by libgsr_hip.so"


def test_abi_version_and_host_only_calls():
    L = _lib.lib()
    assert L.gsr_abi_version() == _lib.ABI_VERSION
    off, nbytes = _lib.buffer_field(_lib.GSR_BUF_GEOM, "splat", 1000, 5000, 256, 256)
    assert (off, nbytes) == (0, 1000 * 20 * 4)
    off2, nb2 = _lib.buffer_field(_lib.GSR_BUF_BINNING, "ranges", 1000, 5000, 250, 130)
    assert nb2 == 16 * 9 * 8 and off2 % 256 == 0
    with pytest.raises(_lib.GsrError, match="unknown buffer field"):
        _lib.buffer_field(0, "nope", 1, 1, 16, 16)
    assert L.gsr_knn3_workspace_bytes(1000) > 1000 * 4 * 5
    assert L.gsr_sort_workspace_bytes(1000) > 0


def test_struct_layout_matches_header():
    # field order/size sanity: pointers are 8 bytes, struct sizes as the C compiler lays them out
    assert ctypes.sizeof(_lib.GsrView) == 4 * 9 + 4 + 8 * 4      # 9 scalars, pad, 4 pointers
    assert ctypes.sizeof(_lib.GsrGaussians) == 8 + 8 * 8
    assert ctypes.sizeof(_lib.GsrForwardOut) == 8 * 3 + 8 + 8 * 3
    assert ctypes.sizeof(_lib.GsrGrads) == 8 * 9


def test_operators_have_no_cpu_path():
    from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from simple_knn._C import distCUDA2
    from gaussmart_amd.synthetic import make_scene, activate
    from conftest import hip_settings
    p, cam = make_scene(10, 32, 32)
    a = activate(p)
    rs = hip_settings(cam, device="cpu")
    assert isinstance(rs, GaussianRasterizationSettings)
    rast = GaussianRasterizer(rs)
    with pytest.raises(_lib.GsrError, match="no CPU path"):
        rast(a["means3D"], torch.zeros(10, 3), a["opacities"], shs=a["shs"], scales=a["scales"], rotations=a["rotations"])
    with pytest.raises(_lib.GsrError, match="no CPU path"):
        distCUDA2(a["means3D"])
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        rast(a["means3D"], torch.zeros(10, 3), a["opacities"], scales=a["scales"], rotations=a["rotations"])


def test_product_never_imports_oracle():
    bad = []
    for base in ("gaussmart_amd", "diff_surfel_rasterization", "simple_knn"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp")):
                    if re.search(r"^\s*(from|import)\s+oracle\b", open(os.path.join(dp, f)).read(), flags=re.M):
                        bad.append(os.path.join(dp, f))
    assert not bad, f"product code imports the oracle: {bad}"
