"""The pybind11 flavour of the binding (gaussmart_amd/csrc/pybind_shim.cpp, gaussmart_amd/pybind_binding.py): builds, loads
without a GPU, reports the library's ABI version, refuses host tensors; on a GPU it drives the same C-ABI entry points as
the ctypes binding and must produce the same bits (reference call site: gaussian_renderer/__init__.py:97-106)."""
import pytest
import torch

from conftest import hip_settings, facing_scene
from gaussmart_amd import _lib
from gaussmart_amd.synthetic import activate


def test_extension_loads_and_reports_the_abi():
    from gaussmart_amd import pybind_binding as PB
    B = PB.module()
    assert B.abi_version() == _lib.ABI_VERSION == _lib.lib().gsr_abi_version()
    assert callable(B.forward) and callable(B.backward) and callable(B.knn3)
    assert B.knn3_workspace_bytes(1000) == _lib.lib().gsr_knn3_workspace_bytes(1000)


def test_operator_has_no_cpu_path_and_checks_arguments():
    from gaussmart_amd.pybind_binding import GaussianRasterizer
    p, cam = facing_scene(10, 32, 32)
    a = activate(p)
    rast = GaussianRasterizer(hip_settings(cam, device="cpu"))
    with pytest.raises(_lib.GsrError, match="no CPU path"):
        rast(a["means3D"], torch.zeros(10, 3), a["opacities"], shs=a["shs"], scales=a["scales"], rotations=a["rotations"])
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        rast(a["means3D"], torch.zeros(10, 3), a["opacities"], scales=a["scales"], rotations=a["rotations"])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["sh", "colors", "transmat"])
def test_pybind_and_ctypes_bindings_give_identical_bits(gpu_device, mode):
    from gaussmart_amd.pybind_binding import GaussianRasterizer as PyRast
    from gaussmart_amd.rasterizer import GaussianRasterizer as CtRast
    from oracle import surfel_ref as O
    from conftest import oracle_settings
    dev = gpu_device
    p, cam = facing_scene(1500, 200, 136, seed=4)
    a = activate(p)
    kw = dict(scales=a["scales"], rotations=a["rotations"], shs=a["shs"])
    if mode != "sh":
        kw.pop("shs"); kw["colors_precomp"] = torch.rand(1500, 3, generator=torch.Generator().manual_seed(1))
    if mode == "transmat":
        S = oracle_settings(cam, 3, torch.float32)
        geom = O.preprocess(a["means3D"], a["scales"], a["rotations"], a["opacities"], a["shs"], None, None, S)
        T = torch.eye(3).reshape(1, 9).repeat(1500, 1)
        T[geom.vis_idx] = geom.Tm.reshape(-1, 9)
        kw.pop("scales"); kw.pop("rotations"); kw["cov3D_precomp"] = T
    g = torch.Generator().manual_seed(2)
    wc, wa = torch.randn(3, 136, 200, generator=g).to(dev), torch.randn(7, 136, 200, generator=g).to(dev)
    outs = []
    for cls in (CtRast, PyRast):
        ins = {k: v.clone().to(dev).requires_grad_(True) for k, v in kw.items()}
        m3 = a["means3D"].clone().to(dev).requires_grad_(True)
        op = a["opacities"].clone().to(dev).requires_grad_(True)
        m2 = torch.zeros(1500, 3, device=dev, requires_grad=True)
        c, r, am = cls(hip_settings(cam, 3, (0.1, 0.3, 0.2), dev))(m3, m2, op, **ins)
        ((c * wc).sum() + (am * wa).sum()).backward()
        torch.cuda.synchronize()
        outs.append([c.detach(), r, am.detach(), m3.grad, m2.grad, op.grad] + [ins[k].grad for k in sorted(ins)])
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    assert int((outs[0][1] > 0).sum()) > 500
