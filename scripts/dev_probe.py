"""Developer aid: sensitivity probes of the compositing kernels.  The library re-reads GSR_K6_PROBE / GSR_K7_PROBE per
launch; each probe adds (or removes) one class of instructions in the per-splat loop, so the change of the event-timed
kernel duration says what that loop is bound by.  One process, one device, static workload (no optimiser step),
probes interleaved over several rounds.  Usage: python scripts/dev_probe.py GSR_K6_PROBE 0,1,2,3,4,5 [preset[:radius]]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gaussmart_amd import _lib
from gaussmart_amd.synthetic import make_scene, jittered_cameras
from gaussmart_amd.gaussian_model import GaussianModel
from gaussmart_amd.gaussian_renderer import render
from gaussmart_amd.params import OptimizationParams, PipelineParams

var = sys.argv[1]
probes = [int(x) for x in sys.argv[2].split(",")]
name = sys.argv[3] if len(sys.argv) > 3 else "headline"
radius = None
if ":" in name:
    name, radius = name.split(":"); radius = float(radius)
ps = bench.PRESETS[name]
N, W, H, r = ps["gaussians"], ps["width"], ps["height"], radius or ps["radius_px"]
dev = torch.device("cuda:0")
params, _ = make_scene(N, W, H, seed=0, radius_px=r)
cam = jittered_cameras(1, W, H, seed=0, device=dev)[0]
bg = torch.zeros(3, device=dev)
pipe, opt = PipelineParams(), OptimizationParams()
pipe.fused_activations = True
m = GaussianModel(3, device=dev); m.create_from_params(params); m.training_setup(opt)
g = torch.Generator(device=dev).manual_seed(1)
wc = torch.randn(3, H, W, device=dev, generator=g) * 1e-3
wa = torch.randn(7, H, W, device=dev, generator=g) * 1e-4
wa[5:] = 0   # median / distortion gradients off, as in the headline step (depth_ratio 0, lambda_dist 0)

def step():
    m.optimizer.zero_grad(set_to_none=True)
    pkg = render(cam, m, pipe, bg, surface_maps=False)
    loss = (pkg["render"] * wc).sum() + (pkg["allmap"] * wa).sum()
    loss.backward()

big = ("render_fwd", "render_bwd")
for _ in range(5): step()
res = {p: [] for p in probes}
for rnd in range(5):
    for p in probes:
        if p: os.environ[var] = str(p)
        else: os.environ.pop(var, None)   # 0 = the product kernel
        for _ in range(2): step()
        _lib.profile_reset(); _lib.profile_enable(big)
        for _ in range(12): step()
        torch.cuda.synchronize()
        _lib.profile_enable(False)
        pr = {k: ms / c for k, (ms, c) in _lib.profile_read().items() if k in big and c}
        res[p].append(pr)
os.environ.pop(var, None)
for p in probes:
    out = []
    for k in big:
        v = sorted(x[k] for x in res[p])
        out.append(f"{k} {v[len(v)//2]*1e3:.1f}us (min {v[0]*1e3:.1f})")
    print(f"{name} r={r} {var}={p}: " + " | ".join(out), flush=True)
