"""Developer aid: does the step time change over a run?  Headline scene, bench's step; one timing event every 10 steps (no host
synchronisation inside the run), printed as ms/step per group of 10 -- after 5 warm-up steps, like the driver's bench call."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussmart_amd.synthetic import make_scene, perturb, jittered_cameras
from gaussmart_amd.gaussian_model import GaussianModel
from gaussmart_amd.gaussian_renderer import render
from gaussmart_amd.params import OptimizationParams, PipelineParams
from gaussmart_amd.trainer import training_step
from gaussmart_amd.view_parallel import ViewParallel
dev = torch.device("cuda:0")
N, W, H = 1000000, 1920, 1080
params, _ = make_scene(N, W, H, seed=0)
cam = jittered_cameras(1, W, H, seed=0, device=dev)[0]
bg = torch.zeros(3, device=dev)
pipe, opt = PipelineParams(), OptimizationParams()
tgt = GaussianModel(3, device=dev); tgt.create_from_params(perturb(params))
with torch.no_grad():
    gt = render(cam, tgt, pipe, bg)["render"].clamp(0, 1).contiguous()
del tgt
m = GaussianModel(3, device=dev); m.create_from_params(params); m.training_setup(opt)
vp = ViewParallel(m, overlap_local=True)
hp = torch.cuda.Stream(device=dev, priority=-1); hp.wait_stream(torch.cuda.current_stream(dev)); torch.cuda.set_stream(hp)
def step(i):
    training_step(m, cam, gt, opt, pipe, bg, 10000 + i, view_parallel=vp, next_cam=cam)
if os.environ.get("PREHEAT"):      # unrelated GPU work before the warm-up: is the slow start a clock ramp or something of ours?
    a = torch.randn(4096, 4096, device=dev); t = time.perf_counter()
    while time.perf_counter() - t < float(os.environ["PREHEAT"]):
        for _ in range(10): a = (a @ a) * 1e-4
        torch.cuda.synchronize()
for i in range(int(os.environ.get("WARM", 5))): step(i)
torch.cuda.synchronize()
if os.environ.get("PERSTEP"):
    n = 30
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    allocs = []
    ev[0].record()
    for i in range(n):
        step(i); ev[i + 1].record()
        st = torch.cuda.memory_stats()
        allocs.append((st["num_device_alloc"], st["reserved_bytes.all.current"] >> 20, st["allocation.all.allocated"]))
    torch.cuda.synchronize()
    for i in range(n):
        print(f"step {i:2d} {ev[i].elapsed_time(ev[i+1]):.3f} ms  device_allocs {allocs[i][0]} reserved {allocs[i][1]} MiB  allocations {allocs[i][2]}")
    sys.exit(0)
if os.environ.get("PERKERNEL"):
    from gaussmart_amd import _lib
    for i in range(24):
        _lib.profile_reset(); _lib.profile_enable(True)
        step(i)
        torch.cuda.synchronize()
        _lib.profile_enable(False)
        pr = {k: ms for k, (ms, c) in _lib.profile_read().items() if c}
        print(f"step {i:2d} " + " ".join(f"{k}={v*1e3:.0f}" for k, v in pr.items()))
    sys.exit(0)
G = 10
evs = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
t0 = time.perf_counter()
evs[0].record()
for g in range(20):
    for i in range(G): step(g * G + i)
    evs[g + 1].record()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print("ms/step per group of 10:", " ".join(f"{evs[g].elapsed_time(evs[g+1]) / G:.3f}" for g in range(20)))
print(f"wall {wall / 200 * 1e3:.3f} ms/step over 200 steps; first 20 by events: {evs[0].elapsed_time(evs[2]) / 20:.3f}")
