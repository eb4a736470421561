"""HIP-backed stand-ins for the two functions train.py takes from the reference's utils/loss_utils.py:

    from utils.loss_utils import l1_loss, ssim            # train.py:15
    Ll1 = l1_loss(image, gt_image)                         # train.py:113
    loss = (1.0 - opt.lambda_dssim) * Ll1 + opt.lambda_dssim * (1.0 - ssim(image, gt_image))

Same names, same argument meaning (utils/loss_utils.py:16-17 and :42-57), same values; each call is ONE launch of the fused
photometric kernel each way (csrc/loss.hip) instead of the five grouped convolutions `ssim` costs through MIOpen -- at 1080p
that torch formulation alone is 61 % of an iteration of the reference's loop around the drop-in operator
(profiles/r04_v4_dropin_kernel_stats.csv).  The swap is one line in train.py:

    from gaussmart_amd.loss_utils import l1_loss, ssim

What the reference's versions accept and these do not is refused loudly, never approximated: tensors must live on a HIP
device (there is no CPU path -- gaussmart_amd.losses holds the torch formulation), the second argument receives no
gradient (the ground truth never does in train.py), `ssim` is the 11-tap window averaged over the whole image.
"""
import torch

from .fused_loss import photometric_loss


def _chw(t, name):
    if t.dim() == 4 and t.shape[0] == 1:        # the reference's ssim() also takes [1,C,H,W]
        t = t[0]
    if t.dim() != 3:
        raise ValueError(f"{name}: expected [C,H,W] (or [1,C,H,W]), got {tuple(t.shape)}")
    return t


def _check(a, b, what):
    if b.requires_grad and torch.is_grad_enabled():
        raise ValueError(f"{what}: the second argument (the ground truth) must not require a gradient")
    a, b = _chw(a, what), _chw(b, what)
    if a.shape != b.shape:
        raise ValueError(f"{what}: shapes differ: {tuple(a.shape)} vs {tuple(b.shape)}")
    return a, b


def l1_loss(network_output, gt):
    """mean |network_output - gt|  (utils/loss_utils.py:16-17)."""
    a, b = _check(network_output, gt, "l1_loss")
    return photometric_loss(a, b, 0.0)[0]


def ssim(img1, img2, window_size=11, size_average=True):
    """Mean SSIM with the reference's 11-tap Gaussian window, sigma 1.5, zero padding (utils/loss_utils.py:42-57)."""
    if window_size != 11 or not size_average:
        raise ValueError("ssim: the HIP kernel implements window_size=11, size_average=True (what train.py calls); "
                         "gaussmart_amd.losses.ssim is the general torch formulation")
    a, b = _check(img1, img2, "ssim")
    return 1.0 - photometric_loss(a, b, 1.0)[0]
