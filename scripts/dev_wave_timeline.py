"""Developer aid (make PROBES=1): when do the waves of ONE render_fwd launch run?  Every wave stamps its start and end
on the 100 MHz s_memrealtime clock with its hardware id (GSR_K6_PROBE=7); this script reads the stamps of the last launch
and prints the duration statistics, the occupancy of the wave slots over time and the spread over XCDs / CUs.
Usage: python scripts/dev_wave_timeline.py [preset[:radius]]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gaussmart_amd import _lib
from gaussmart_amd.synthetic import make_scene, jittered_cameras
from gaussmart_amd.gaussian_model import GaussianModel
from gaussmart_amd.gaussian_renderer import render
from gaussmart_amd.params import OptimizationParams, PipelineParams

name = sys.argv[1] if len(sys.argv) > 1 else "headline"
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
m = GaussianModel(3, device=dev); m.create_from_params(params); m.training_setup(opt)
os.environ["GSR_K6_PROBE"] = "7"
for _ in range(4):
    pkg = render(cam, m, pipe, bg, surface_maps=False)
    pkg["render"].sum().backward()
    m.optimizer.zero_grad(set_to_none=True)
torch.cuda.synchronize()
lib = _lib.lib()
n_tiles = ((W + 15) // 16) * ((H + 15) // 16)
per_xcd = (n_tiles + 7) // 8
n_waves = 8 * per_xcd * 4
buf = np.zeros(5 * n_waves, dtype=np.uint64)
lib.gsr_probe_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = lib.gsr_probe_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes)
assert rc == 0
st = buf.reshape(-1, 5)
ok = st[:, 1] > 0
t0, t1, hid = st[ok, 0].astype(np.int64), st[ok, 1].astype(np.int64), st[ok, 2]
base = t0.min()
t0 = (t0 - base) / 100.0; t1 = (t1 - base) / 100.0          # microseconds
dur = t1 - t0
span = t1.max()
cyc = (st[ok, 4].astype(np.int64) - st[ok, 3].astype(np.int64)).sum()
print(f"shader clock held during the launch: {cyc / (dur.sum() * 100.0) * 100.0:.0f} MHz (sum of s_memtime spans / sum of s_memrealtime spans)")
print(f"{name} r={r}: {ok.sum()} waves, launch span {span:.1f} us; wave duration mean {dur.mean():.1f} us, "
      f"p10 {np.percentile(dur, 10):.1f}, median {np.median(dur):.1f}, p90 {np.percentile(dur, 90):.1f}, max {dur.max():.1f}")
print(f"sum of wave durations / (span x 7168 slots) = {dur.sum() / (span * 7168):.3f}")
# occupancy over time
edges = np.linspace(0, span, 41)
occ = []
for a, b in zip(edges[:-1], edges[1:]):
    occ.append((np.clip(np.minimum(t1, b) - np.maximum(t0, a), 0, None)).sum() / (b - a))
print("active waves per 2.5% of the launch:", " ".join(f"{o:.0f}" for o in occ))
xcc = (hid >> np.uint64(32)).astype(np.int64)
hw = (hid & np.uint64(0xFFFFFFFF)).astype(np.int64)
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
for x in range(8):
    sel = xcc == x
    if sel.any():
        print(f"xcc {x}: {sel.sum()} waves, busy {dur[sel].sum() / span / 896:.3f} of its 896 slots, last wave ends at {t1[sel].max():.1f} us, "
              f"first start {t0[sel].min():.1f}")
cuid = xcc * 1000 + se * 100 + sh * 16 + cu
u, inv = np.unique(cuid, return_inverse=True)
busy = np.bincount(inv, weights=dur) / span / 28
print(f"{len(u)} distinct (xcc, se, sh, cu); slot occupancy per CU: min {busy.min():.3f} median {np.median(busy):.3f} max {busy.max():.3f}")
ends = np.array([t1[inv == i].max() for i in range(len(u))])
print(f"per-CU time of last wave end: min {ends.min():.1f} median {np.median(ends):.1f} max {ends.max():.1f} us")
