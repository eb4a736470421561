"""Developer aid: a few hundred iterations with densification / pruning at the headline size (robustness + timing)."""
import os, sys, time, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussmart_amd.gaussian_model import GaussianModel
from gaussmart_amd.gaussian_renderer import render
from gaussmart_amd.losses import psnr
from gaussmart_amd.params import OptimizationParams, PipelineParams
from gaussmart_amd.synthetic import make_scene, perturb, jittered_cameras
from gaussmart_amd.trainer import train
dev = torch.device("cuda:0")
n, w, h = int(os.environ.get("N", 1000000)), 1920, 1080
params, _ = make_scene(n, w, h, seed=1)
cams = jittered_cameras(8, w, h, seed=1, device=dev, amount=0.3)
pipe, opt, bg = PipelineParams(), OptimizationParams(), torch.zeros(3, device=dev)
target = GaussianModel(3, device=dev); target.create_from_params(params)
with torch.no_grad():
    for c in cams:
        c.original_image = render(c, target, pipe, bg, surface_maps=False)["render"].clamp(0, 1)
del target
m = GaussianModel(3, device=dev); m.create_from_params(perturb(params, pos=0.02, log_scale=0.2, opa=0.5, color=0.3)); m.training_setup(opt)
def mean_psnr():
    with torch.no_grad():
        return float(torch.stack([psnr(render(c, m, pipe, bg, surface_maps=False)["render"][None], c.original_image[None]).mean() for c in cams]).mean())
before = mean_psnr()
torch.cuda.synchronize(); t0 = time.time()
train(m, cams, opt, pipe, bg, cameras_extent=5.0, first_iter=600, iterations=900, log_every=100)
torch.cuda.synchronize(); dt = time.time() - t0
after = mean_psnr()
print(f"PSNR {before:.2f} -> {after:.2f} dB, points {n} -> {m.get_xyz.shape[0]}, {300 / dt:.1f} it/s incl. densification, "
      f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
assert math.isfinite(after) and after > before
