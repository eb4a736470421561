"""gaussmart_amd.loss_utils: the reference's `l1_loss` / `ssim` signatures (utils/loss_utils.py:16-17, 42-57) on the fused HIP
kernel -- values and gradients against the torch formulation of the same functions (gaussmart_amd.losses, which is pinned to
the reference by tests/golden/loss.npz), used the way train.py:113-114 uses them."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(3, 64, 80), (3, 270, 480), (1, 33, 47)])
def test_values_and_gradients_match_the_torch_formulation(gpu_device, shape):
    from gaussmart_amd import loss_utils as H, losses as T
    dev = gpu_device
    g = torch.Generator().manual_seed(3)
    gt = torch.rand(shape, generator=g).to(dev)
    base = (gt.cpu() + 0.2 * torch.randn(shape, generator=g)).clamp(0, 1).to(dev)
    out = {}
    for name, M in (("hip", H), ("torch", T)):
        img = base.clone().requires_grad_(True)
        l1, s = M.l1_loss(img, gt), M.ssim(img, gt)
        loss = 0.8 * l1 + 0.2 * (1.0 - s)                     # train.py:113-114
        loss.backward()
        out[name] = (float(l1), float(s), float(loss), img.grad.clone())
    for i in range(3):
        assert abs(out["hip"][i] - out["torch"][i]) < 2e-6, (i, out["hip"][i], out["torch"][i])
    gh, gt_ = out["hip"][3], out["torch"][3]
    assert float((gh - gt_).abs().max()) < 1e-5 * float(gt_.abs().max()) + 1e-9
    # each function on its own, and the [1,C,H,W] form the reference's ssim also takes
    img = base.clone().requires_grad_(True)
    H.ssim(img[None], gt[None]).backward()
    img2 = base.clone().requires_grad_(True)
    T.ssim(img2[None], gt[None]).backward()
    assert float((img.grad - img2.grad).abs().max()) < 1e-5 * float(img2.grad.abs().max()) + 1e-9


def test_unsupported_arguments_are_refused(gpu_device):
    from gaussmart_amd import _lib, loss_utils as H
    dev = gpu_device
    a, b = torch.rand(3, 32, 32, device=dev, requires_grad=True), torch.rand(3, 32, 32, device=dev)
    with pytest.raises(ValueError):
        H.ssim(a, b, window_size=7)
    with pytest.raises(ValueError):
        H.ssim(a, b, size_average=False)
    with pytest.raises(ValueError):
        H.l1_loss(a, b.clone().requires_grad_(True))
    with pytest.raises(ValueError):
        H.l1_loss(a, b[:, :16])
    with pytest.raises(_lib.GsrError):
        H.l1_loss(a.detach().cpu(), b.cpu())           # no CPU path
