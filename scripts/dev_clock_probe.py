"""Developer aid (make PROBES=1): which clock does the chip hold while K6 / K7 run?  Probe 7 of both kernels makes every
wave bracket its life with s_memtime (shader cycles) and s_memrealtime (100 MHz); the ratio of the sums is the effective
shader clock.  With the per-launch VALU instruction counts of the committed PMC summary this turns the kernels' issue rate
into cycles per wave64 instruction per SIMD at the clock that was really held.
Usage: python scripts/dev_clock_probe.py [preset]  -> one JSON line"""
import ctypes, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gaussmart_amd import _lib
from gaussmart_amd.synthetic import make_scene, perturb, jittered_cameras
from gaussmart_amd.gaussian_model import GaussianModel
from gaussmart_amd.gaussian_renderer import render
from gaussmart_amd.params import OptimizationParams, PipelineParams
from gaussmart_amd.trainer import training_losses

name = sys.argv[1] if len(sys.argv) > 1 else "headline"
ps = bench.PRESETS[name]
N, W, H, r = ps["gaussians"], ps["width"], ps["height"], ps["radius_px"]
dev = torch.device("cuda:0")
params, _ = make_scene(N, W, H, seed=0, radius_px=r)
cam = jittered_cameras(1, W, H, seed=0, device=dev)[0]
bg = torch.zeros(3, device=dev)
pipe, opt = PipelineParams(), OptimizationParams()
tgt = GaussianModel(3, device=dev); tgt.create_from_params(perturb(params))
with torch.no_grad():
    gt = render(cam, tgt, pipe, bg, surface_maps=False)["render"].clamp(0, 1).contiguous()
del tgt
m = GaussianModel(3, device=dev); m.create_from_params(params); m.training_setup(opt)
lib = _lib.lib()
res = {"preset": name}
for probe in ("0", "7"):
    os.environ["GSR_K6_PROBE"] = probe; os.environ["GSR_K7_PROBE"] = probe
    _lib.profile_reset(); _lib.profile_enable(("render_fwd", "render_bwd"))
    for _ in range(12):
        pkg = render(cam, m, pipe, bg, surface_maps=False)
        total, _ = training_losses(pkg, gt, opt, 10000, cam, pipe)
        total.backward()
        m.optimizer.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    prof = _lib.profile_read()
    res[f"probe{probe}_ms"] = {k: round(prof[k][0] / max(prof[k][1], 1), 4) for k in ("render_fwd", "render_bwd")}
n_tiles = ((W + 15) // 16) * ((H + 15) // 16)
n_waves = 8 * ((n_tiles + 7) // 8) * 4
buf = np.zeros(5 * n_waves, dtype=np.uint64)
lib.gsr_probe_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.gsr_probe_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) == 0
st = buf.reshape(-1, 5).astype(np.int64)
ok = st[:, 1] > 0
res["render_fwd"] = {"waves": int(ok.sum()), "clock_mhz": round(float((st[ok, 4] - st[ok, 3]).sum() / (st[ok, 1] - st[ok, 0]).sum() * 100.0), 1),
                     "wave_cycles_sum": int((st[ok, 4] - st[ok, 3]).sum())}
buf = np.zeros(2 * n_waves, dtype=np.uint64)
lib.gsr_probe_read_stamps_bwd.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.gsr_probe_read_stamps_bwd(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) == 0
st = buf.reshape(-1, 2).astype(np.int64)
ok = st[:, 1] > 0
res["render_bwd"] = {"waves": int(ok.sum()), "clock_mhz": round(float(st[ok, 0].sum() / st[ok, 1].sum() * 100.0), 1),
                     "wave_cycles_sum": int(st[ok, 0].sum())}
# issue rate against 2 cycles per wave64 instruction per SIMD (MI355X_MICROARCH.md, v_fma_f32 row) at the clock held
pmc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
if name == "headline" and os.path.exists(pmc):
    t = json.load(open(pmc))
    for k in ("render_fwd", "render_bwd"):
        valu = (t.get("kernels", {}).get(k) or {}).get("valu_insts")
        if valu:
            ms = res["probe0_ms"][k]
            clk = res[k]["clock_mhz"] * 1e6
            res[k]["valu_wave_instr_per_launch"] = valu
            res[k]["cycles_per_valu_instr_per_simd"] = round(1024 * clk * ms * 1e-3 / valu, 3)
            res[k]["issue_frac_of_2_cycles_per_instr"] = round(2.0 / (1024 * clk * ms * 1e-3 / valu), 3)
print(json.dumps(res))
