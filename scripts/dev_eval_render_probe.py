"""Developer probe: frames per second of render() under torch.no_grad() with all derived maps (the evaluation / mesh-extraction
path of the reference, render.py) at the headline shape, with the torch post-processing and with fused_surface_maps."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from gaussmart_amd.gaussian_model import GaussianModel
from gaussmart_amd.gaussian_renderer import render
from gaussmart_amd.params import PipelineParams
from gaussmart_amd.synthetic import make_scene, jittered_cameras
dev = torch.device("cuda:0")
params, _ = make_scene(1_000_000, 1920, 1080, seed=0)
cam = jittered_cameras(1, 1920, 1080, device=dev)[0]
m = GaussianModel(3, device=dev); m.create_from_params(params)
bg = torch.zeros(3, device=dev)
for fa, fm in ((False, False), (True, False), (True, True), (False, True), (True, False), (False, False), (True, True)):
    pipe = PipelineParams(); pipe.fused_activations, pipe.fused_surface_maps = fa, fm
    with torch.no_grad():
        for _ in range(10): render(cam, m, pipe, bg)
        torch.cuda.synchronize(); t = time.time()
        for _ in range(100): render(cam, m, pipe, bg)
        torch.cuda.synchronize()
    print(f"render() under no_grad, all maps: fused_activations {fa} fused_surface_maps {fm}: {100 / (time.time() - t):.1f} frames/s")
