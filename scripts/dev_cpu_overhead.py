"""Developer aid: host-side cost of one training step -- a scene so small that the GPU work is a few launch latencies, so the
wall time per step is what Python + ctypes + torch spend enqueueing it.  Prints ms/step and a cProfile top list."""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussmart_amd.synthetic import make_scene, perturb, jittered_cameras
from gaussmart_amd.gaussian_model import GaussianModel
from gaussmart_amd.gaussian_renderer import render
from gaussmart_amd.params import OptimizationParams, PipelineParams
from gaussmart_amd.trainer import training_step
from gaussmart_amd.view_parallel import ViewParallel
dev = torch.device("cuda:0")
N, W, H = 2000, 64, 64
params, _ = make_scene(N, W, H, seed=0)
cam = jittered_cameras(1, W, H, seed=0, device=dev)[0]
bg = torch.zeros(3, device=dev)
pipe, opt = PipelineParams(), OptimizationParams()
tgt = GaussianModel(3, device=dev); tgt.create_from_params(perturb(params))
with torch.no_grad():
    gt = render(cam, tgt, pipe, bg)["render"].clamp(0, 1).contiguous()
m = GaussianModel(3, device=dev); m.create_from_params(params); m.training_setup(opt)
vp = ViewParallel(m, overlap_local=True) if os.environ.get("VP", "1") == "1" else None
def step(i):
    training_step(m, cam, gt, opt, pipe, bg, 10000 + i, view_parallel=vp, next_cam=cam)
for i in range(30): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(300): step(i)
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 300 * 1e3:.3f} ms per step (host-bound)")
pr = cProfile.Profile(); pr.enable()
for i in range(200): step(i)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
