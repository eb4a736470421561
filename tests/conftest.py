import math
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def oracle_settings(cam, deg=3, dtype=torch.float32, bg=(0.0, 0.0, 0.0), scale_modifier=1.0):
    from oracle import surfel_ref as O
    return O.Settings(cam.image_height, cam.image_width, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2),
                      torch.tensor(bg, dtype=dtype), scale_modifier, cam.world_view_transform.cpu().to(dtype),
                      cam.full_proj_transform.cpu().to(dtype), deg, cam.camera_center.cpu().to(dtype))


def hip_settings(cam, deg=3, bg=(0.0, 0.0, 0.0), device="cuda:0", scale_modifier=1.0):
    from gaussmart_amd.rasterizer import GaussianRasterizationSettings
    return GaussianRasterizationSettings(
        cam.image_height, cam.image_width, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2),
        torch.tensor(bg, dtype=torch.float32, device=device), scale_modifier, cam.world_view_transform.to(device),
        cam.full_proj_transform.to(device), deg, cam.camera_center.to(device), False, False)


def facing_scene(n, w, h, seed=0, tilt=0.35, **kw):
    """Synthetic scene whose surfels face the camera within a moderate tilt: keeps the ray-splat
    intersection well conditioned so fp32-vs-fp64 comparisons measure the kernels, not the
    conditioning of edge-on splats."""
    from gaussmart_amd.synthetic import make_scene
    params, cam = make_scene(n, w, h, seed=seed, **kw)
    g = torch.Generator().manual_seed(seed + 77)
    q = torch.zeros(n, 4)
    q[:, 0] = 1.0
    q = q + tilt * torch.randn(n, 4, generator=g)
    params["rotation"] = q.to(params["rotation"].dtype)
    return params, cam


@pytest.fixture(scope="session")
def gpu_device():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")
