"""Developer aid: per-step wall time + allocator activity of the bench loop."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussmart_amd.synthetic import make_scene, jittered_cameras
from gaussmart_amd.gaussian_model import GaussianModel
from gaussmart_amd.params import OptimizationParams, PipelineParams
from gaussmart_amd.trainer import training_step
dev = torch.device("cuda:0")
N, W, H = 1000000, 1920, 1080
params, _ = make_scene(N, W, H)
cam = jittered_cameras(1, W, H, device=dev)[0]
bg = torch.zeros(3, device=dev); gt = torch.rand(3, H, W, device=dev)
pipe, opt = PipelineParams(), OptimizationParams()
m = GaussianModel(3, device=dev); m.create_from_params(params); m.training_setup(opt)
prev = torch.cuda.memory_stats()["num_device_alloc"]
for it in range(40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    training_step(m, cam, gt, opt, pipe, bg, 10000 + it)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = torch.cuda.memory_stats()
    print(f"it {it:2d} {dt*1e3:7.2f} ms  device_allocs {st['num_device_alloc']-prev} frees {st['num_device_free']} reserved {st['reserved_bytes.all.current']/2**30:.2f} GiB retries {st['num_alloc_retries']}")
    prev = st["num_device_alloc"]
