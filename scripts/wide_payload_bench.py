"""SURVEY 8(d) wide-payload stress: the 1M / 1920x1080 synthetic scene rendered with colors_precomp of
C in {3, 16, 64} channels; per-kernel time of forward + backward from the library profiler."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussmart_amd import _lib
from gaussmart_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
from gaussmart_amd.synthetic import activate, make_scene
import math

dev = torch.device("cuda:0")
N, W, H = int(os.environ.get("N", 1000000)), 1920, 1080
params, cam = make_scene(N, W, H)
a = {k: v.to(dev) for k, v in activate(params).items()}
g = torch.Generator().manual_seed(0)
rows = []
for Cn in (3, 16, 64):
    col = torch.rand(N, Cn, generator=g).to(dev).requires_grad_(True)
    ins = {k: a[k].clone().requires_grad_(True) for k in ("means3D", "opacities", "scales", "rotations")}
    rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(Cn, device=dev), 1.0,
                                       cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), 3,
                                       cam.camera_center.to(dev), False, False)
    rast = GaussianRasterizer(rs)
    wc, wa = torch.randn(Cn, H, W, device=dev), torch.randn(7, H, W, device=dev)
    _lib.profile_enable(True)
    for it in range(8):
        if it == 3:
            torch.cuda.synchronize(); _lib.profile_reset()
        m2d = torch.zeros(N, 3, device=dev, requires_grad=True)
        c, r, am = rast(means3D=ins["means3D"], means2D=m2d, colors_precomp=col, opacities=ins["opacities"],
                        scales=ins["scales"], rotations=ins["rotations"])
        ((c * wc).sum() + (am * wa).sum()).backward()
    torch.cuda.synchronize()
    prof = {k: round(ms / max(n, 1), 4) for k, (ms, n) in _lib.profile_read().items() if n}
    _lib.profile_enable(False)
    rows.append(dict(channels=Cn, ms_per_launch=prof))
    print(json.dumps(rows[-1]), flush=True)
    del col, ins, c, r, am, m2d
