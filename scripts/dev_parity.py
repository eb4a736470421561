"""Developer scratch: HIP vs oracle statistics on a small scene (run on the GPU box)."""
import math, sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import surfel_ref as O
from gaussmart_amd.synthetic import make_scene, activate
from gaussmart_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer, rasterize_debug

def osettings(cam, deg, dt, bg):
    return O.Settings(cam.image_height, cam.image_width, math.tan(cam.FoVx/2), math.tan(cam.FoVy/2),
        torch.tensor(bg, dtype=dt), 1.0, cam.world_view_transform.cpu().to(dt), cam.full_proj_transform.cpu().to(dt), deg, cam.camera_center.cpu().to(dt))
def hsettings(cam, deg, bg, dev):
    return GaussianRasterizationSettings(cam.image_height, cam.image_width, math.tan(cam.FoVx/2), math.tan(cam.FoVy/2),
        torch.tensor(bg, dtype=torch.float32, device=dev), 1.0, cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), deg, cam.camera_center.to(dev), False, False)

N, W, H = int(os.environ.get("N", 2000)), int(os.environ.get("W", 256)), int(os.environ.get("H", 256))
dev = torch.device("cuda:0")
p, cam = make_scene(N, W, H, seed=0)
a = activate(p)
bg = (0.2, 0.4, 0.6)
t = time.time()
dbg = rasterize_debug(a["means3D"].to(dev), a["opacities"].to(dev), a["shs"].to(dev), None, a["scales"].to(dev), a["rotations"].to(dev), None, raster_settings=hsettings(cam, 3, bg, dev))
print("hip fwd ok", time.time()-t, "D=", dbg["num_rendered"], "visible", int((dbg["radii"]>0).sum()))

# ---- K1 vs oracle fp32 preprocess
S32 = osettings(cam, 3, torch.float32, bg)
geom = O.preprocess(a["means3D"], a["scales"], a["rotations"], a["opacities"], a["shs"], None, None, S32)
radii_h = dbg["radii"].cpu()
print("radii mismatches:", int((radii_h != geom.radii).sum()), "of", N)
vi = geom.vis_idx
sp = dbg["splat"].cpu()[vi]
ref = torch.cat([geom.Tm.reshape(-1, 9), geom.xy, geom.normal, a["opacities"][vi], geom.rgb], 1)
err = (sp[:, :18] - ref).abs() / (ref.abs() + 1e-3)
print("splat max rel err per field:", err.max(0).values.numpy().round(7))

# ---- binning bit-exact vs numpy on HIP's own K1 output
gx = (W + 15)//16; gy = (H+15)//16
spl = dbg["splat"].cpu().numpy(); rad = radii_h.numpy()
rect = np.zeros((N, 4), np.int32)
def tdiv(v): return np.trunc(v / 16).astype(np.int64)
vis = rad > 0
cx, cy = spl[:, 9], spl[:, 10]
with np.errstate(all="ignore"):
    rect[:, 0] = np.clip(tdiv(cx - rad), 0, gx); rect[:, 1] = np.clip(tdiv(cy - rad), 0, gy)
    rect[:, 2] = np.clip(tdiv(cx + rad + 15), 0, gx); rect[:, 3] = np.clip(tdiv(cy + rad + 15), 0, gy)
rect[~vis] = 0
keys, plist = O.bin_tiles(None, rad, rect, dbg["depth_key"].cpu().numpy().view(np.float32).copy(), gx)
ranges = O.tile_ranges(keys, gx*gy)
pl_h = dbg["point_list"].cpu().numpy().astype(np.uint32)
rg_h = dbg["ranges"].cpu().numpy().astype(np.uint32)
print("D oracle", len(plist), "point_list equal:", np.array_equal(pl_h, plist), "ranges equal:", np.array_equal(rg_h, ranges))

# ---- K6 vs oracle render on HIP's geometry (fp64 math on fp32 inputs)
def geom_from_splat(spl, dt):
    s = torch.from_numpy(spl).to(dt)
    return s[:, 0:9].reshape(-1, 3, 3).contiguous(), s[:, 9:11].contiguous(), s[:, 11:14].contiguous(), s[:, 14].contiguous(), s[:, 15:18].contiguous()
for dt in (torch.float64,):
    S = osettings(cam, 3, dt, bg)
    gT, gxy, gn, go, gc = geom_from_splat(spl, dt)
    t = time.time()
    out = O.render_tiles(gT, gxy, gn, go, gc, torch.from_numpy(plist.astype(np.int64)), ranges, S, margins=True)
    print("oracle render", dt, time.time()-t)
    col_h, am_h = dbg["color"].cpu().to(dt), dbg["allmap"].cpu().to(dt)
    m = out.margins
    stable = (m["m_alpha"] > 1e-3) & (m["m_term"] > 1e-3) & (m["m_rho"] > 1e-3)
    print("stable pixel fraction", stable.float().mean().item())
    def rep(name, x, y, mask):
        d = (x - y).abs()
        sc = y.abs().max().item() + 1e-12
        print(f"  {name}: max|d| all {d.max().item():.3e} stable {d[..., mask].max().item():.3e}  scale {sc:.3e}  rel(stable) {d[..., mask].max().item()/sc:.3e}")
    rep("color", col_h, out.color, stable)
    names = ["depth", "alpha", "nx", "ny", "nz", "median", "dist"]
    for c in range(7):
        mk = stable & (m["m_med"] > 1e-4) if c == 5 else stable
        rep(names[c], am_h[c], out.allmap[c], mk)
    rep("final_T", dbg["final_T"].cpu().to(dt), out.final_T, stable)
    nc = dbg["n_contrib"].cpu().to(torch.int64)
    nc[1][nc[1] == 0xFFFFFFFF] = -1
    nc1 = dbg["n_contrib"][1].cpu(); 
    print("  n_contrib mismatches (stable):", int(((nc[0] != out.n_contrib[0]) & stable).sum()), " all:", int((nc[0] != out.n_contrib[0]).sum()))
    print("  med_contrib mismatches (stable):", int(((nc1.to(torch.int64) != out.n_contrib[1]) & stable & (m["m_med"] > 1e-4)).sum()))

# ---- end-to-end backward vs oracle (fp64 on same inputs)
for flags in (3, 0):
    torch.manual_seed(1)
    wc = torch.randn(3, H, W); wa = torch.randn(7, H, W) * torch.tensor([1, 1, 1, 1, 1, 1, 1.0])[:, None, None]
    # HIP
    inp = {k: v.clone().to(dev).requires_grad_(True) for k, v in a.items()}
    m2d = torch.zeros(N, 3, device=dev, requires_grad=True)
    rast = GaussianRasterizer(hsettings(cam, 3, bg, dev), flags=flags)
    c, r, am = rast(means3D=inp["means3D"], means2D=m2d, shs=inp["shs"], colors_precomp=None, opacities=inp["opacities"], scales=inp["scales"], rotations=inp["rotations"], cov3D_precomp=None)
    ((c * wc.to(dev)).sum() + (am * wa.to(dev)).sum()).backward()
    torch.cuda.synchronize()
    # oracle fp64
    S = osettings(cam, 3, torch.float64, bg)
    oin = {k: v.clone().double().requires_grad_(True) for k, v in a.items()}
    om2d = torch.zeros(N, 3, dtype=torch.float64, requires_grad=True)
    t = time.time()
    oc, orr, oam = O.rasterize(oin["means3D"], om2d, oin["opacities"], oin["shs"], None, oin["scales"], oin["rotations"], None, settings=S, flags=flags)
    ((oc * wc.double()).sum() + (oam * wa.double()).sum()).backward()
    print(f"flags={flags} oracle fwd+bwd {time.time()-t:.1f}s; fwd color max diff {(c.detach().cpu().double()-oc).abs().max().item():.3e}")
    for k in ["means3D", "opacities", "shs", "scales", "rotations"]:
        gh = inp[k].grad.cpu().double(); go_ = oin[k].grad
        d = (gh - go_).abs()
        sc = go_.abs().max().item()
        # per-Gaussian relative
        rown = go_.reshape(N, -1).abs().max(1).values
        rel_row = d.reshape(N, -1).max(1).values / (rown + 1e-6 * sc)
        print(f"  d{k}: max|d| {d.max().item():.3e} scale {sc:.3e} normwise {d.max().item()/sc:.3e}  per-row rel median {rel_row.median().item():.2e} p99 {rel_row.quantile(0.99).item():.2e} max {rel_row.max().item():.2e}")
    gh = m2d.grad.cpu().double(); go_ = om2d.grad
    print(f"  dmeans2D: max|d| {(gh-go_).abs().max().item():.3e} scale {go_.abs().max().item():.3e}")
