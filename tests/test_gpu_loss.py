"""Fused photometric loss (HIP) vs the oracle restatement of the reference's l1_loss / ssim, and
vs the golden values captured from the reference itself (tests/golden/loss.npz)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import loss_ref

pytestmark = pytest.mark.gpu


def test_matches_reference_golden(gpu_device):
    from gaussmart_amd.fused_loss import photometric_loss
    g = np.load(os.path.join(GOLDEN, "loss.npz"))
    a, b = torch.from_numpy(g["a"]).to(gpu_device), torch.from_numpy(g["b"]).to(gpu_device)
    loss, l1, ssim = photometric_loss(a, b, 0.2)
    np.testing.assert_allclose(l1.item(), g["l1"], rtol=2e-6)
    np.testing.assert_allclose(ssim.item(), g["ssim"], rtol=1e-5)
    np.testing.assert_allclose(loss.item(), 0.8 * g["l1"] + 0.2 * (1 - g["ssim"]), rtol=1e-5)


@pytest.mark.parametrize("shape,lam", [((3, 37, 53), 0.2), ((3, 256, 256), 0.2), ((1, 16, 16), 0.5), ((3, 1080, 1920), 0.2),
                                       ((3, 5, 7), 1.0), ((3, 130, 250), 0.0)])
def test_forward_backward_vs_oracle(gpu_device, shape, lam):
    from gaussmart_amd.fused_loss import photometric_loss
    g = torch.Generator().manual_seed(shape[1])
    x = torch.rand(shape, generator=g)
    y = (x + 0.1 * torch.randn(shape, generator=g)).clamp(0, 1)
    xo = x.clone().double().requires_grad_(True)
    lo, l1o, so = loss_ref.photometric_loss(xo, y.double(), lam)
    (3.0 * lo).backward()
    xh = x.clone().to(gpu_device).requires_grad_(True)
    lh, l1h, sh = photometric_loss(xh, y.to(gpu_device), lam)
    (3.0 * lh).backward()
    np.testing.assert_allclose(lh.item(), lo.item(), rtol=2e-5)
    np.testing.assert_allclose(sh.item(), so.item(), rtol=2e-5)
    gh, go = xh.grad.cpu().double(), xo.grad
    assert float((gh - go).abs().max()) <= 2e-4 * float(go.abs().max()) + 1e-12


def test_loss_is_reproducible(gpu_device):
    from gaussmart_amd.fused_loss import photometric_loss
    x = torch.rand(3, 300, 500, device=gpu_device, requires_grad=True)
    y = torch.rand(3, 300, 500, device=gpu_device)
    a = photometric_loss(x, y, 0.2)[0]; a.backward(); ga = x.grad.clone(); x.grad = None
    b = photometric_loss(x, y, 0.2)[0]; b.backward()
    assert torch.equal(a, b) and torch.equal(ga, x.grad)
