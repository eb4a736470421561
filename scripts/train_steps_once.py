"""Developer aid: a few full training steps on the headline scene (1M Gaussians, 1920x1080), the target of the
rocprofv3 --pmc passes of scripts/pmc_collect.sh (every kernel of a step, same code path as bench.py)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussmart_amd.synthetic import make_scene, perturb, jittered_cameras
from gaussmart_amd.gaussian_model import GaussianModel
from gaussmart_amd.gaussian_renderer import render
from gaussmart_amd.params import OptimizationParams, PipelineParams
from gaussmart_amd.trainer import training_step
dev = torch.device("cuda:0")
N, W, H = int(os.environ.get("N", 1000000)), int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080))
steps = int(os.environ.get("STEPS", 4))
params, _ = make_scene(N, W, H, seed=0)
cam = jittered_cameras(1, W, H, seed=0, device=dev)[0]
bg = torch.zeros(3, device=dev)
pipe, opt = PipelineParams(), OptimizationParams()
tgt = GaussianModel(3, device=dev); tgt.create_from_params(perturb(params))
with torch.no_grad():
    gt = render(cam, tgt, pipe, bg)["render"].clamp(0, 1).contiguous()
del tgt
m = GaussianModel(3, device=dev); m.create_from_params(params); m.training_setup(opt)
for it in range(steps):
    training_step(m, cam, gt, opt, pipe, bg, 10000 + it)
torch.cuda.synchronize()
print("done")
