"""Developer aid: per tile, the longest staged prefix (`covered`) against the list length."""
import os, sys, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussmart_amd.rasterizer import rasterize_debug, GaussianRasterizationSettings
from gaussmart_amd.synthetic import make_scene, activate
dev = torch.device("cuda:0")
N, W, H = int(os.environ.get("N", 200000)), 1920, 1080
params, cam = make_scene(N, W, H, seed=0)
a = {k: v.to(dev) for k, v in activate(params).items()}
rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev), 1.0,
                                   cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), 3, cam.camera_center.to(dev), False, False)
dbg = rasterize_debug(a["means3D"], a["opacities"], a["shs"], None, a["scales"], a["rotations"], None, raster_settings=rs)
cov = dbg["covered"].long()
rng = dbg["ranges"].long()
n = rng[:, 1] - rng[:, 0]
mx = cov.max(dim=1).values
print("D", dbg["num_rendered"], "sum lists", int(n.sum()), "sum max covered", int(mx.sum()), "tiles with covered > list", int((mx > n).sum()))
bad = torch.nonzero(mx > n)[:5, 0]
for t in bad.tolist():
    print("tile", t, "list", int(n[t]), "covered", cov[t].tolist())
