"""Image losses of the training step: L1, SSIM (11x11 Gaussian window, sigma 1.5), PSNR.

Counterparts of utils/loss_utils.py:16-57 and utils/image_utils.py:19-21 of the reference;
checked against golden vectors of those functions in tests/test_golden.py.
"""
import math

import torch
import torch.nn.functional as F

_WINDOWS = {}


def _window(size: int, channels: int, like: torch.Tensor) -> torch.Tensor:
    key = (size, channels, like.device, like.dtype)
    w = _WINDOWS.get(key)
    if w is None:
        g = torch.tensor([math.exp(-(x - size // 2) ** 2 / (2 * 1.5 ** 2)) for x in range(size)])
        g = (g / g.sum()).unsqueeze(1)
        w2d = (g @ g.t()).float()[None, None]
        w = w2d.expand(channels, 1, size, size).contiguous().to(device=like.device, dtype=like.dtype)
        _WINDOWS[key] = w
    return w


def l1_loss(network_output, gt):
    return (network_output - gt).abs().mean()


def l2_loss(network_output, gt):
    return ((network_output - gt) ** 2).mean()


def ssim(img1, img2, window_size=11, size_average=True):
    ch = img1.size(-3)
    w = _window(window_size, ch, img1)
    pad = window_size // 2
    mu1 = F.conv2d(img1, w, padding=pad, groups=ch)
    mu2 = F.conv2d(img2, w, padding=pad, groups=ch)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = F.conv2d(img1 * img1, w, padding=pad, groups=ch) - mu1_sq
    s2 = F.conv2d(img2 * img2, w, padding=pad, groups=ch) - mu2_sq
    s12 = F.conv2d(img1 * img2, w, padding=pad, groups=ch) - mu12
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu12 + c1) * (2 * s12 + c2)) / ((mu1_sq + mu2_sq + c1) * (s1 + s2 + c2))
    return m.mean() if size_average else m.mean(1).mean(1).mean(1)


def psnr(img1, img2):
    mse = ((img1 - img2) ** 2).reshape(img1.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))
