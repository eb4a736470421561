"""Developer probe: the model a sparse (SfM-like) start grows into -- few, large, overlapping surfels -- is where the forward
costs as much as the backward (profiles/r04_schedule_settings/headline_sparse.*).  Trains it for `--iterations`, then reports,
for one training view: instances D, tile-list length, entries walked per pixel, contributors per pixel, and K6 / K7 time."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import full_schedule_train as FS
from gaussmart_amd import _lib
from gaussmart_amd.gaussian_renderer import render
from gaussmart_amd.params import PipelineParams
from gaussmart_amd.rasterizer import rasterize_debug, GaussianRasterizationSettings
from gaussmart_amd.synthetic import jittered_cameras
import math

ap = argparse.ArgumentParser()
ap.add_argument("--iterations", type=int, default=16000)
ap.add_argument("--start-fraction", type=float, default=0.1)
a = ap.parse_args()
models = []
s = FS.run("headline", a.iterations, 8, 4000, 1, 5.0, schedule_iterations=30000, quiet=False, model_out=models, start_fraction=a.start_fraction)
m = models[0]
dev = m.get_xyz.device
cam = jittered_cameras(10, 1920, 1080, seed=1, device=dev, amount=0.3)[3]
pipe, bg = PipelineParams(), torch.zeros(3, device=dev)
rs = GaussianRasterizationSettings(1080, 1920, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), bg, 1.0, cam.world_view_transform,
                                   cam.full_proj_transform, 3, cam.camera_center, False, False)
with torch.no_grad():
    dbg = rasterize_debug(m.get_xyz, m.get_opacity, m.get_features, None, m.get_scaling, m.get_rotation, None, raster_settings=rs)
D = int(dbg["num_rendered"])
ranges = dbg["ranges"].cpu().long()
lens = (ranges[:, 1] - ranges[:, 0]).float()
radii = dbg["radii"].float()
print(f"N {m.get_xyz.shape[0]}  D {D}  instances per Gaussian {D / max(int((radii > 0).sum()), 1):.1f}  tile list mean {lens.mean():.0f} max {int(lens.max())}"
      f"  radius px: median {float(radii[radii > 0].median()):.0f} p90 {float(radii[radii > 0].quantile(0.9)):.0f} max {int(radii.max())}")
scal = m.get_scaling.detach()
print(f"scales (scene units): median {float(scal.median()):.4f} p90 {float(scal.flatten().quantile(0.9)):.4f}; opacity median {float(m.get_opacity.median()):.3f}")
_lib.profile_reset(); _lib.profile_enable(True)
for _ in range(5):
    pkg = render(cam, m, pipe, bg, surface_maps=False)
    (pkg["render"].sum() + pkg["allmap"].sum()).backward()
torch.cuda.synchronize(); _lib.profile_enable(False)
print({k: round(ms / n, 3) for k, (ms, n) in _lib.profile_read().items() if n})
pkg = render(cam, m, pipe, bg, surface_maps=False)
