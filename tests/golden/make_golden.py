#!/usr/bin/env python3
"""Generate golden input/output vectors from the reference's CPU-importable helpers.

Run ONCE in the build container (needs /root/reference); the resulting .npz files are
committed and are the only thing that travels.  Nothing here is imported by the product.

Reference functions exercised (file:line in /root/reference):
  utils/sh_utils.py:57-117     eval_sh, RGB2SH, SH2RGB
  utils/graphics_utils.py:39-72 getWorld2View2, getProjectionMatrix, fov2focal, focal2fov
  utils/loss_utils.py:16-57    l1_loss, ssim
  utils/image_utils.py:19-21   psnr
  utils/general_utils.py:18-62 inverse_sigmoid, get_expon_lr_func
"""
import os
import sys

import numpy as np
import torch

REF = os.environ.get("GSR_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    from utils.sh_utils import eval_sh, RGB2SH, SH2RGB
    from utils.graphics_utils import getWorld2View2, getProjectionMatrix, fov2focal, focal2fov
    from utils.loss_utils import l1_loss, ssim
    from utils.image_utils import psnr
    from utils.general_utils import inverse_sigmoid, get_expon_lr_func

    g = torch.Generator().manual_seed(1234)

    # ---- SH ----------------------------------------------------------------------
    n = 257
    sh = torch.randn(n, 3, 16, generator=g, dtype=torch.float64)
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g, dtype=torch.float64), dim=-1)
    sh_out = {f"eval_deg{d}": eval_sh(d, sh, dirs).numpy() for d in range(4)}
    rgb = torch.rand(n, 3, generator=g, dtype=torch.float64)
    np.savez(os.path.join(OUT, "sh.npz"), sh=sh.numpy(), dirs=dirs.numpy(), rgb=rgb.numpy(),
             rgb2sh=RGB2SH(rgb).numpy(), sh2rgb=SH2RGB(rgb).numpy(), **sh_out)

    # ---- camera matrices ----------------------------------------------------------
    rng = np.random.default_rng(99)
    cams = {}
    for i in range(4):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        w, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        T = rng.normal(size=3) * 2
        trans = rng.normal(size=3) * 0.3 if i % 2 else np.zeros(3)
        scale = 1.0 if i < 2 else 1.7
        fovx, fovy = 0.6 + 0.2 * i, 0.5 + 0.15 * i
        cams[f"R{i}"] = R; cams[f"T{i}"] = T; cams[f"trans{i}"] = trans
        cams[f"scale{i}"] = np.float64(scale); cams[f"fov{i}"] = np.array([fovx, fovy])
        cams[f"w2v{i}"] = getWorld2View2(R, T, trans, scale)
        cams[f"proj{i}"] = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).numpy()
        cams[f"focal{i}"] = np.array([fov2focal(fovx, 640), focal2fov(fov2focal(fovy, 480), 480)])
    np.savez(os.path.join(OUT, "camera.npz"), **cams)

    # ---- image losses --------------------------------------------------------------
    a = torch.rand(3, 37, 53, generator=g)
    b = (a + 0.1 * torch.randn(3, 37, 53, generator=g)).clamp(0, 1)
    np.savez(os.path.join(OUT, "loss.npz"), a=a.numpy(), b=b.numpy(),
             l1=l1_loss(a, b).numpy(), ssim=ssim(a, b).numpy(),
             ssim_per=ssim(a[None], b[None], size_average=False).numpy(),
             psnr=psnr(a[None], b[None]).numpy())

    # ---- scalar helpers ----------------------------------------------------------
    x = torch.rand(64, generator=g) * 0.98 + 0.01
    f = get_expon_lr_func(lr_init=0.00016, lr_final=0.0000016, lr_delay_mult=0.01, max_steps=30000)
    steps = np.array([0, 1, 10, 100, 1000, 7000, 15000, 29999, 30000, 40000])
    f2 = get_expon_lr_func(lr_init=1e-2, lr_final=1e-4, lr_delay_steps=500, lr_delay_mult=0.1, max_steps=2000)
    np.savez(os.path.join(OUT, "scalar.npz"), x=x.numpy(), inv_sigmoid=inverse_sigmoid(x).numpy(),
             steps=steps, lr=np.array([f(int(s)) for s in steps]), lr2=np.array([f2(int(s)) for s in steps]))
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
