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
    config.addinivalue_line("markers", "oracle_cases(*keys): oracle-farm cases this test asks for (tests/oracle_farm.py)")


def pytest_collection_finish(session):
    """Start the CPU oracle of every selected GPU parity case in worker processes (tests/oracle_farm.py): they run beside
    the GPU tests instead of inside them.  Only on a box with a device -- elsewhere those tests skip."""
    try:
        from oracle_farm import FARM
    except Exception:
        return
    keys = []
    for item in session.items:
        if item.get_closest_marker("gpu") is None:
            continue
        cs = getattr(item, "callspec", None)
        if cs is not None:
            keys += [v for v in cs.params.values() if isinstance(v, str) and v in FARM.specs]
        m = item.get_closest_marker("oracle_cases")
        if m is not None:
            keys += [k for k in m.args if k in FARM.specs]
        fn = getattr(item, "function", None)
        keys += [k for k in getattr(fn, "oracle_keys", ()) if k in FARM.specs]
    keys = list(dict.fromkeys(keys))
    if keys and torch.cuda.device_count() > 0:
        n = FARM.start(keys)
        print(f"\n[oracle farm] {len(keys)} oracle cases on {n} worker processes", flush=True)


def pytest_sessionfinish(session, exitstatus):
    try:
        from oracle_farm import FARM, REPORT, format_report
    except Exception:
        return
    FARM.shutdown()
    if REPORT:
        text = format_report()
        out = os.environ.get("GSR_BARS_REPORT")
        if out:
            os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
            with open(out, "w") as f:
                f.write(text + "\n")
        print("\n[gradient bars] measured figures per case and tensor (HIP vs fp64 oracle | fp32 oracle vs fp64 | bar usage)")
        print(text)


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
