"""Deterministic synthetic scenes for tests, smoke and bench (SURVEY.md section 8(d) recipe).

Camera at the origin looking down +z; Gaussians fill the frustum between z=2 and z=10 with a
log-normal size distribution whose mean projected 3-sigma radius is `radius_px` pixels.
All randomness comes from one torch.Generator so every rank / run sees the same scene.
"""
import math

import numpy as np
import torch

from .camera import look_at_camera, fov2focal
from .sh import RGB2SH


def make_scene(n_gaussians: int, width: int, height: int, *, seed: int = 0, radius_px: float = 6.0,
               fovx_deg: float = 60.0, sh_degree: int = 3, dtype=torch.float32, device="cpu",
               znear_scene: float = 2.0, zfar_scene: float = 10.0):
    """Returns (params, camera).  `params` holds the raw (pre-activation) tensors in
    GaussianModel's storage layout: xyz [N,3], features_dc [N,1,3], features_rest [N,K-1,3],
    scaling (log) [N,2], rotation (un-normalised wxyz) [N,4], opacity (logit) [N,1]."""
    g = torch.Generator().manual_seed(seed)
    fovx = math.radians(fovx_deg)
    focal = fov2focal(fovx, width)
    tanx = math.tan(fovx / 2)
    tany = tanx * height / width

    z = torch.rand(n_gaussians, generator=g, dtype=torch.float64) * (zfar_scene - znear_scene) + znear_scene
    ndc = (torch.rand(n_gaussians, 2, generator=g, dtype=torch.float64) * 2 - 1) * 1.1
    xyz = torch.stack([ndc[:, 0] * tanx * z, ndc[:, 1] * tany * z, z], -1)

    inv_z_mean = math.log(zfar_scene / znear_scene) / (zfar_scene - znear_scene)
    sigma_ln = 0.5
    mean_scale = (radius_px / 3.0) / (focal * inv_z_mean)
    mu = math.log(mean_scale) - 0.5 * sigma_ln ** 2
    scaling = mu + sigma_ln * torch.randn(n_gaussians, 2, generator=g, dtype=torch.float64)
    rotation = torch.randn(n_gaussians, 4, generator=g, dtype=torch.float64)
    opacity = 1.5 * torch.randn(n_gaussians, 1, generator=g, dtype=torch.float64)
    k = (sh_degree + 1) ** 2
    f_dc = RGB2SH(torch.rand(n_gaussians, 1, 3, generator=g, dtype=torch.float64))
    f_rest = 0.05 * torch.randn(n_gaussians, k - 1, 3, generator=g, dtype=torch.float64)

    params = dict(xyz=xyz, features_dc=f_dc, features_rest=f_rest, scaling=scaling,
                  rotation=rotation, opacity=opacity)
    params = {k_: v.to(dtype).to(device).contiguous() for k_, v in params.items()}
    cam = look_at_camera(eye=(0, 0, 0), target=(0, 0, 1), up=(0, -1, 0), fovx=fovx, width=width,
                         height=height, device=device)
    return params, cam


def perturb(params, *, seed: int = 1, pos=0.01, log_scale=0.1, rot=0.05, opa=0.3, color=0.1):
    """A nearby scene: rendering it gives a target image with non-trivial gradients."""
    g = torch.Generator().manual_seed(seed)
    amt = dict(xyz=pos, features_dc=color, features_rest=color * 0.2, scaling=log_scale,
               rotation=rot, opacity=opa)
    out = {}
    for k, v in params.items():
        noise = torch.randn(v.shape, generator=g, dtype=torch.float32).to(v.dtype).to(v.device)
        out[k] = (v + amt[k] * noise).contiguous()
    return out


def activate(params):
    """Raw storage -> what render() hands to the rasterizer (scene/gaussian_model.py:103-123)."""
    return dict(
        means3D=params["xyz"],
        scales=torch.exp(params["scaling"]),
        rotations=torch.nn.functional.normalize(params["rotation"]),
        opacities=torch.sigmoid(params["opacity"]),
        shs=torch.cat((params["features_dc"], params["features_rest"]), dim=1),
    )


def jittered_cameras(n: int, width: int, height: int, *, seed: int = 0, fovx_deg: float = 60.0,
                     device="cpu", amount: float = 0.15):
    """n views around the canonical one (for view-parallel training / multi-view tests)."""
    rng = np.random.default_rng(seed)
    cams = []
    for i in range(n):
        eye = rng.normal(size=3) * amount * (0 if i == 0 else 1)
        tgt = np.array([0, 0, 6.0]) + rng.normal(size=3) * amount * (0 if i == 0 else 1)
        cams.append(look_at_camera(eye, tgt, (0, -1, 0), math.radians(fovx_deg), width, height,
                                   device=device, uid=i))
    return cams
