"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's photometric loss
(utils/loss_utils.py:16-57: l1_loss, gaussian window, ssim; combined as train.py:113-114).
PINNED by tests/golden/loss.npz (outputs of the reference's own l1_loss / ssim captured in the
build container, tests/golden/make_golden.py)."""
from math import exp

import torch
import torch.nn.functional as F


def gaussian_window(window_size=11, sigma=1.5, channels=3, dtype=torch.float32):
    g = torch.Tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    g = (g / g.sum()).unsqueeze(1)
    w = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)
    return w.expand(channels, 1, window_size, window_size).contiguous().to(dtype)


def ssim_map(img1, img2, window_size=11):
    ch = img1.size(-3)
    w = gaussian_window(window_size, 1.5, ch, img1.dtype)
    p = window_size // 2
    mu1, mu2 = F.conv2d(img1, w, padding=p, groups=ch), F.conv2d(img2, w, padding=p, groups=ch)
    s1 = F.conv2d(img1 * img1, w, padding=p, groups=ch) - mu1.pow(2)
    s2 = F.conv2d(img2 * img2, w, padding=p, groups=ch) - mu2.pow(2)
    s12 = F.conv2d(img1 * img2, w, padding=p, groups=ch) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1.pow(2) + mu2.pow(2) + C1) * (s1 + s2 + C2))


def photometric_loss(image, gt, lambda_dssim=0.2):
    l1 = (image - gt).abs().mean()
    ssim = ssim_map(image, gt).mean()
    return (1.0 - lambda_dssim) * l1 + lambda_dssim * (1.0 - ssim), l1, ssim
