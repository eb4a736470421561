"""Developer aid: interleaved A/B of kernel variants selected by environment variables the library reads per launch,
in ONE process on ONE device (cdna_hip_programming.md rule 24).  Usage:
  python scripts/ab_kernels.py VAR [preset ...]     e.g.  python scripts/ab_kernels.py GSR_K6_OLD headline scan24
Prints per preset the per-launch times of the big kernels and the step time with VAR unset / set, over several rounds."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gaussmart_amd import _lib
from gaussmart_amd.synthetic import make_scene, perturb, jittered_cameras
from gaussmart_amd.gaussian_model import GaussianModel
from gaussmart_amd.gaussian_renderer import render
from gaussmart_amd.params import OptimizationParams, PipelineParams
from gaussmart_amd.trainer import training_step

var = sys.argv[1]
presets = sys.argv[2:] or ["headline"]
dev = torch.device("cuda:0")
big = ("render_fwd", "render_bwd", "preprocess_bwd", "preprocess_fwd")
for name in presets:
    radius = None
    if ":" in name:
        name, radius = name.split(":"); radius = float(radius)
    ps = bench.PRESETS[name]
    N, W, H, r = ps["gaussians"], ps["width"], ps["height"], radius or ps["radius_px"]
    params, _ = make_scene(N, W, H, seed=0, radius_px=r)
    cam = jittered_cameras(1, W, H, seed=0, device=dev)[0]
    bg = torch.zeros(3, device=dev)
    pipe, opt = PipelineParams(), OptimizationParams()
    tgt = GaussianModel(3, device=dev); tgt.create_from_params(perturb(params))
    with torch.no_grad():
        gt = render(cam, tgt, pipe, bg)["render"].clamp(0, 1).contiguous()
    del tgt
    m = GaussianModel(3, device=dev); m.create_from_params(params); m.training_setup(opt)
    for i in range(10):
        training_step(m, cam, gt, opt, pipe, bg, 10000 + i)
    res = {0: [], 1: []}
    for rnd in range(4):
        for setting in (0, 1):
            if setting: os.environ[var] = "1"
            else: os.environ.pop(var, None)
            for i in range(3):
                training_step(m, cam, gt, opt, pipe, bg, 10000 + i)
            _lib.profile_reset(); _lib.profile_enable(big)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 30
            for i in range(n):
                training_step(m, cam, gt, opt, pipe, bg, 10000 + i)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e3
            _lib.profile_enable(False)
            pr = {k: ms / c for k, (ms, c) in _lib.profile_read().items() if k in big and c}
            res[setting].append((dt, pr))
    os.environ.pop(var, None)
    for setting in (0, 1):
        steps = sorted(x[0] for x in res[setting])
        ks = {k: sorted(x[1][k] for x in res[setting])[len(res[setting]) // 2] for k in big}
        print(f"{name} r={r} {var}={'1' if setting else '-'}: step median {steps[len(steps)//2]:.3f} ms (min {steps[0]:.3f}) | " +
              " ".join(f"{k} {v*1e3:.0f}us" for k, v in ks.items()), flush=True)
    del m; torch.cuda.empty_cache()
