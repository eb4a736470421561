"""Developer aid: wall time of the parts of a training step, fused_activations on/off."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussmart_amd.synthetic import make_scene, jittered_cameras
from gaussmart_amd.gaussian_model import GaussianModel
from gaussmart_amd.gaussian_renderer import render
from gaussmart_amd.params import OptimizationParams, PipelineParams
from gaussmart_amd.trainer import training_losses
dev = torch.device("cuda:0")
N, W, H = 1000000, 1920, 1080
params, _ = make_scene(N, W, H)
cam = jittered_cameras(1, W, H, device=dev)[0]
bg = torch.zeros(3, device=dev)
gt = torch.rand(3, H, W, device=dev)
for fused in (False, True):
    pipe, opt = PipelineParams(fused_activations=fused), OptimizationParams()
    m = GaussianModel(3, device=dev); m.create_from_params(params); m.training_setup(opt)
    def sync(): torch.cuda.synchronize(); return time.perf_counter()
    acc = [0, 0, 0, 0]
    for it in range(13):
        t0 = sync(); pkg = render(cam, m, pipe, bg, surface_maps=False)
        t1 = sync(); total, parts = training_losses(pkg, gt, opt, 10000 + it, cam, pipe)
        t2 = sync(); total.backward()
        t3 = sync(); m.optimizer.step(); m.optimizer.zero_grad(set_to_none=True)
        t4 = sync()
        if it >= 3:
            for k, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)): acc[k] += d
    print("fused" if fused else "plain", [round(a / 10 * 1e3, 3) for a in acc], "ms: render, loss, backward, adam")
