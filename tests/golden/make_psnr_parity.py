"""PSNR-parity stand-in (BASELINE config 1): trains the 2k-Gaussian 256x256 scene for 300 iterations on the HOST with the
oracle standing in for the rasterizer operator -- evaluated in fp64 on the fp32 parameters -- the reference's torch loss
(utils/loss_utils.py formulation, gaussmart_amd/losses.py) and torch.optim.Adam, and stores the PSNR curve and the final
parameters in tests/golden/psnr_parity_c1.npz.  tests/test_gpu_psnr_parity.py trains the same schedule with the HIP
operator, the fused objective and the fused Adam and compares.

What this pins: HIP-trained == oracle-trained.  It does NOT pin either against upstream (DTU scan24 within 0.05 dB of the
reference needs the dataset and the upstream rasterizer, neither of which exists here: SURVEY 8(c)) -- "parity unpinned".

    python tests/golden/make_psnr_parity.py        (about ten minutes on one core; no GPU, no reference import)
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from gaussmart_amd import gaussian_renderer                                   # noqa: E402
from gaussmart_amd.gaussian_model import GaussianModel                       # noqa: E402
from gaussmart_amd.losses import psnr                                        # noqa: E402
from gaussmart_amd.params import OptimizationParams, PipelineParams          # noqa: E402
from gaussmart_amd.synthetic import make_scene, perturb, jittered_cameras    # noqa: E402
from gaussmart_amd.trainer import train, TrainState                          # noqa: E402
from oracle import surfel_ref as O                                           # noqa: E402

N, W, H, VIEWS, SEED = 2000, 256, 256, 8, 0
FIRST, LAST, EVAL_EVERY = 7000, 7300, 50      # iterations 7001..7300 of the schedule: lambda_normal is on (train.py:131)


def config():
    """Scene, cameras and hyper-parameters shared by the generator and the GPU test."""
    params, _ = make_scene(N, W, H, seed=SEED)
    start = perturb(params, pos=0.02, log_scale=0.2, opa=0.5, color=0.3)
    opt = OptimizationParams()
    opt.densify_until_iter = 0          # no densification: the comparison is about the operator, the loss and the step
    return params, start, opt, PipelineParams()


class Oracle64(torch.nn.Module):
    """Duck-type of diff_surfel_rasterization.GaussianRasterizer: the oracle evaluated in fp64 on fp32 tensors."""

    def __init__(self, raster_settings, flags=O.QUIRKS_UPSTREAM):
        super().__init__()
        self.inner = O.OracleRasterizer(raster_settings._replace(
            bg=raster_settings.bg.double(), viewmatrix=raster_settings.viewmatrix.double(),
            projmatrix=raster_settings.projmatrix.double(), campos=raster_settings.campos.double()), flags)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        d = lambda t: None if t is None else t.double()
        c, r, am = self.inner(d(means3D), d(means2D), d(opacities), d(shs), d(colors_precomp), d(scales), d(rotations),
                              d(cov3D_precomp))
        return c.float(), r, am.float()


def main():
    torch.set_num_threads(1)
    gaussian_renderer.GaussianRasterizer = Oracle64
    params, start, opt, pipe = config()
    bg = torch.zeros(3)
    cams = jittered_cameras(VIEWS, W, H, seed=SEED, device="cpu", amount=0.3)
    target = GaussianModel(3, device="cpu")
    target.create_from_params(params)
    with torch.no_grad():
        for c in cams:
            c.original_image = gaussian_renderer.render(c, target, pipe, bg)["render"].clamp(0, 1).contiguous()
    m = GaussianModel(3, device="cpu")
    m.create_from_params(start)
    m.use_fused_adam = False
    m.training_setup(opt)

    def mean_psnr():
        with torch.no_grad():
            return float(torch.stack([psnr(gaussian_renderer.render(c, m, pipe, bg)["render"].clamp(0, 1)[None],
                                           c.original_image[None]).mean() for c in cams]).mean())

    curve = [(FIRST, mean_psnr())]
    t0 = time.time()

    def on_iteration(it):
        if it % EVAL_EVERY == 0:
            curve.append((it, mean_psnr()))
            print(f"[it {it}] PSNR {curve[-1][1]:.4f} dB  ({time.time() - t0:.0f} s)", flush=True)

    train(m, cams, opt, pipe, bg, cameras_extent=5.0, first_iter=FIRST, iterations=LAST, seed=SEED, state=TrainState(SEED),
          on_iteration=on_iteration)
    out = dict(curve=np.array(curve, dtype=np.float64),
               xyz=m._xyz.detach().numpy(), opacity=m._opacity.detach().numpy(), scaling=m._scaling.detach().numpy(),
               rotation=m._rotation.detach().numpy(), features_dc=m._features_dc.detach().numpy(),
               # (the whole model: the GPU test renders it next to the HIP-trained one -- 330 KB of the 450 KB fixture)
               features_rest=m._features_rest.detach().numpy(),
               gt_mean=np.array([float(c.original_image.mean()) for c in cams]))
    np.savez_compressed(os.path.join(HERE, "psnr_parity_c1.npz"), **out)
    print("written", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
