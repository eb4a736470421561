"""Developer aid: which torch ops (not library kernels) run inside one training step, with Python stacks."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.profiler import profile, ProfilerActivity
from gaussmart_amd.synthetic import make_scene, jittered_cameras
from gaussmart_amd.gaussian_model import GaussianModel
from gaussmart_amd.params import OptimizationParams, PipelineParams
from gaussmart_amd.trainer import training_step
dev = torch.device("cuda:0")
N, W, H = 200000, 1920, 1080
params, _ = make_scene(N, W, H)
cam = jittered_cameras(1, W, H, device=dev)[0]
bg = torch.zeros(3, device=dev)
gt = torch.rand(3, H, W, device=dev)
pipe, opt = PipelineParams(), OptimizationParams()
m = GaussianModel(3, device=dev); m.create_from_params(params); m.training_setup(opt)
for it in range(3):
    training_step(m, cam, gt, opt, pipe, bg, 10000 + it)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for it in range(2):
        training_step(m, cam, gt, opt, pipe, bg, 10010 + it)
    torch.cuda.synchronize()
seen = {}
for e in prof.events():
    ks = getattr(e, "kernels", [])
    if e.name.startswith("aten::") and ks:
        st = [s.split("/")[-1] for s in (e.stack or []) if ".py" in s and "torch/" not in s][:4]
        st.append(str([tuple(x) for x in (e.input_shapes or [])][:2]))
        key = (e.name, tuple(st))
        seen[key] = seen.get(key, 0) + 1
for (name, st), n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(n, name, " <- ", " | ".join(st))
