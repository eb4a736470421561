"""Developer aid: N forward+backward passes of the rasterizer on the headline scene (for rocprofv3)."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussmart_amd.synthetic import make_scene, activate
from gaussmart_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
N, W, H, reps = int(os.environ.get("N", 1000000)), int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080)), int(os.environ.get("REPS", 3))
dev = torch.device("cuda:0")
p, cam = make_scene(N, W, H, seed=0)
a = {k: v.to(dev).requires_grad_(True) for k, v in activate(p).items()}
rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev), 1.0,
                                   cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), 3, cam.camera_center.to(dev), False, False)
g = torch.Generator(device=dev).manual_seed(0)
wc, wa = torch.randn(3, H, W, device=dev, generator=g), torch.randn(7, H, W, device=dev, generator=g) * 0.1
for _ in range(reps):
    c, r, am = GaussianRasterizer(rs)(means3D=a["means3D"], means2D=torch.zeros(N, 3, device=dev, requires_grad=True), shs=a["shs"],
                                      opacities=a["opacities"], scales=a["scales"], rotations=a["rotations"])
    ((c * wc).sum() + (am * wa).sum()).backward()
torch.cuda.synchronize()
print("done")
