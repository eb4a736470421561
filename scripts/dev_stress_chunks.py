import os, sys, time, math, torch
sys.path.insert(0, "/root/repo")
from gaussmart_amd.gaussian_model import GaussianModel
from gaussmart_amd.gaussian_renderer import render
from gaussmart_amd.params import OptimizationParams, PipelineParams
from gaussmart_amd.synthetic import make_scene, perturb, jittered_cameras
from gaussmart_amd.trainer import train, TrainState
dev = torch.device("cuda:0")
n, w, h = 1000000, 1920, 1080
params, _ = make_scene(n, w, h, seed=1)
cams = jittered_cameras(8, w, h, seed=1, device=dev, amount=0.3)
pipe, opt, bg = PipelineParams(), OptimizationParams(), torch.zeros(3, device=dev)
if os.environ.get("COLOR_CACHE") == "0": pipe.color_cache = False
target = GaussianModel(3, device=dev); target.create_from_params(params)
with torch.no_grad():
    for c in cams:
        c.original_image = render(c, target, pipe, bg, surface_maps=False)["render"].clamp(0, 1)
del target
m = GaussianModel(3, device=dev); m.create_from_params(perturb(params, pos=0.02, log_scale=0.2, opa=0.5, color=0.3)); m.training_setup(opt)
st = TrainState(0)
it = 600
for chunk in range(30):
    torch.cuda.synchronize(); t0 = time.time()
    train(m, cams, opt, pipe, bg, cameras_extent=5.0, first_iter=it, iterations=it + 10, state=st, final_iteration=30000)
    torch.cuda.synchronize(); dt = time.time() - t0
    it += 10
    print(f"it {it}: {dt/10*1e3:.2f} ms/it  points {m.get_xyz.shape[0]} reserved {torch.cuda.memory_reserved()/2**30:.1f} GiB", flush=True)
