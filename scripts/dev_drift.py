"""Developer aid: step time per 10-step chunk over a few hundred training steps of a preset, with allocator activity
(GSR_POOL_DEBUG=1 prints) interleaved -- is a slow-down drift of the workload (D, entries walked) or of the runtime?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gaussmart_amd.synthetic import make_scene, perturb, jittered_cameras
from gaussmart_amd.gaussian_model import GaussianModel
from gaussmart_amd.gaussian_renderer import render
from gaussmart_amd.params import OptimizationParams, PipelineParams
from gaussmart_amd.trainer import training_step
from gaussmart_amd import rasterizer as R
name = sys.argv[1] if len(sys.argv) > 1 else "headline"
radius = float(sys.argv[2]) if len(sys.argv) > 2 else None
ps = bench.PRESETS[name]
N, W, H, r = ps["gaussians"], ps["width"], ps["height"], radius or ps["radius_px"]
dev = torch.device("cuda:0")
params, _ = make_scene(N, W, H, seed=0, radius_px=r)
cam = jittered_cameras(1, W, H, seed=0, device=dev)[0]
bg = torch.zeros(3, device=dev)
pipe, opt = PipelineParams(), OptimizationParams()
tgt = GaussianModel(3, device=dev); tgt.create_from_params(perturb(params))
with torch.no_grad():
    gt = render(cam, tgt, pipe, bg)["render"].clamp(0, 1).contiguous()
del tgt
m = GaussianModel(3, device=dev); m.create_from_params(params); m.training_setup(opt)
from gaussmart_amd import _lib
for c in range(int(os.environ.get("CHUNKS", 40))):
    prof = c % 13 == 0
    if prof:
        _lib.profile_reset(); _lib.profile_enable(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(10):
        pkg, _ = training_step(m, cam, gt, opt, pipe, bg, 10000 + c * 10 + i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10 * 1e3
    vis = int((pkg["radii"] > 0).sum())
    print(f"chunk {c:3d}: {dt:7.3f} ms/step  visible {vis}  reserved {torch.cuda.memory_reserved(dev) / 2**30:.2f} GiB  "
          f"opacity mean {float(m.get_opacity.detach().mean()):.4f} scale mean {float(m.get_scaling.detach().mean()):.5f}", flush=True)
    if prof:
        _lib.profile_enable(False)
        print("      per-launch ms:", {k: round(ms / n, 3) for k, (ms, n) in _lib.profile_read().items() if n}, flush=True)
