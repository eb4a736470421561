"""Developer aid: where do channels 0..2 of a wide payload differ from the RGB kernel fed the same columns?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import hip_settings
from test_gpu_wide_payload import _wide_inputs
from gaussmart_amd.rasterizer import GaussianRasterizer
dev = torch.device("cuda:0")
b, cam, bg = _wide_inputs(3000, 320, 200, 8, seed=5)
def run(cols, bgv):
    rast = GaussianRasterizer(hip_settings(cam, 3, bgv, dev), flags=3)
    with torch.no_grad():
        c, r, am = rast(means3D=b["means3D"].to(dev), means2D=torch.zeros(3000, 3, device=dev), colors_precomp=cols.to(dev),
                        opacities=b["opacities"].to(dev), scales=b["scales"].to(dev), rotations=b["rotations"].to(dev))
    return c, am
c8, am8 = run(b["colors_precomp"], bg)
c3, am3 = run(b["colors_precomp"][:, :3].contiguous(), bg[:3])
d = (c8[:3] - c3).abs()
print("colour: mismatching values", int((c8[:3] != c3).sum()), "of", c3.numel(), "max abs diff", float(d.max()), "allmap equal", torch.equal(am8, am3))
idx = torch.nonzero(c8[:3] != c3)[:10]
for i in idx.tolist():
    print(i, float(c8[i[0], i[1], i[2]]), float(c3[i[0], i[1], i[2]]))
print("bg", bg)
