"""BASELINE config 1: the host-side mirror of the reference's callers (GaussianModel, render(),
the training step) exercised on CPU with the ORACLE standing in for the HIP operator (test
infrastructure only: the product has no such switch)."""
import numpy as np
import pytest
import torch

from gaussmart_amd import gaussian_renderer
from gaussmart_amd.gaussian_model import GaussianModel
from gaussmart_amd.params import OptimizationParams, PipelineParams
from gaussmart_amd.synthetic import make_scene, perturb
from gaussmart_amd.trainer import training_step, densification_step
from oracle import surfel_ref as O


@pytest.fixture()
def oracle_backend(monkeypatch):
    monkeypatch.setattr(gaussian_renderer, "GaussianRasterizer", O.OracleRasterizer)


def _model(n=2000, w=256, h=256, seed=0):
    params, cam = make_scene(n, w, h, seed=seed, device="cpu")
    m = GaussianModel(3, device="cpu")
    m.create_from_params(params)
    m.use_fused_adam = False
    m.training_setup(OptimizationParams())
    return m, cam, params


def test_config1_forward_backward_2k_256(oracle_backend):
    m, cam, params = _model()
    pipe, bg = PipelineParams(), torch.zeros(3)
    pkg = gaussian_renderer.render(cam, m, pipe, bg)
    assert set(pkg) == {"render", "viewspace_points", "visibility_filter", "radii", "rend_alpha", "rend_normal",
                        "rend_dist", "surf_depth", "surf_normal", "allmap"}
    assert pkg["render"].shape == (3, 256, 256) and pkg["rend_normal"].shape == (3, 256, 256)
    assert pkg["surf_depth"].shape == (1, 256, 256) and pkg["radii"].dtype == torch.int32
    assert torch.equal(pkg["visibility_filter"], pkg["radii"] > 0)
    loss = pkg["render"].mean() + pkg["rend_dist"].mean() + (pkg["rend_normal"] * pkg["surf_normal"]).sum(0).mean()
    loss.backward()
    for p in m.parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all()
    g2d = pkg["viewspace_points"].grad
    assert g2d.shape == (2000, 3) and torch.all(g2d[:, 2] == 0) and g2d.abs().sum() > 0


def test_training_reduces_loss(oracle_backend):
    m, cam, params = _model(300, 64, 64, seed=2)
    pipe, bg, opt = PipelineParams(), torch.zeros(3), OptimizationParams()
    tgt = GaussianModel(3, device="cpu")
    tgt.create_from_params(perturb(params))
    with torch.no_grad():
        gt = gaussian_renderer.render(cam, tgt, pipe, bg)["render"].clamp(0, 1)
    losses = []
    for it in range(1, 9):
        pkg, parts = training_step(m, cam, gt, opt, pipe, bg, 8000 + it)
        losses.append(float(parts["loss"]))
    assert losses[-1] < losses[0]


def test_densification_bookkeeping(oracle_backend):
    m, cam, params = _model(300, 64, 64, seed=3)
    pipe, bg, opt = PipelineParams(), torch.zeros(3), OptimizationParams()
    gt = torch.rand(3, 64, 64)
    n0 = m.get_xyz.shape[0]
    pkg, _ = training_step(m, cam, gt, opt, pipe, bg, 600, step_optimizer=False)
    densification_step(m, pkg, opt, 600, cameras_extent=5.0)
    assert float(m.denom.sum()) == 0.0          # the densify step consumed and reset the statistics
    n1 = m.get_xyz.shape[0]
    for t in (m._features_dc, m._features_rest, m._opacity, m._scaling, m._rotation, m.max_radii2D, m._segments):
        assert t.shape[0] == n1
    st = m.optimizer.state[m._xyz]
    assert st == {} or st["exp_avg"].shape[0] == n1
    m.optimizer.step()     # the optimiser still works on the re-created parameters


def test_compute_cov3d_python_path_matches(oracle_backend):
    m, cam, _ = _model(200, 64, 64, seed=5)
    bg = torch.zeros(3)
    a = gaussian_renderer.render(cam, m, PipelineParams(), bg)
    b = gaussian_renderer.render(cam, m, PipelineParams(compute_cov3D_python=True), bg)
    torch.testing.assert_close(a["render"], b["render"], atol=2e-5, rtol=0)
    torch.testing.assert_close(a["rend_alpha"], b["rend_alpha"], atol=2e-5, rtol=0)


def test_ply_roundtrip_and_column_order(tmp_path):
    m, _, _ = _model(50, 32, 32)
    m._segments = torch.arange(50)
    path = str(tmp_path / "point_cloud" / "iteration_7" / "point_cloud.ply")
    m.save_ply(path)
    head = open(path, "rb").read(4096).split(b"end_header")[0].decode()
    props = [l.split()[-1] for l in head.splitlines() if l.startswith("property")]
    assert props[:6] == ["x", "y", "z", "nx", "ny", "nz"] and props[6:9] == ["f_dc_0", "f_dc_1", "f_dc_2"]
    assert props[9] == "f_rest_0" and props[53] == "f_rest_44" and props[54] == "opacity"
    assert props[55:57] == ["scale_0", "scale_1"] and props[57:61] == ["rot_0", "rot_1", "rot_2", "rot_3"]
    assert props[61] == "segment" and len(props) == 62
    m2 = GaussianModel(3, device="cpu")
    m2.load_ply(path)
    for a, b in zip(m.parameters(), m2.parameters()):
        assert torch.equal(a.detach(), b.detach())
    assert torch.equal(m2._segments, torch.arange(50))
    # f_rest is stored channel-major (transpose(1,2).flatten): column f_rest_0 is coefficient 1 of RED
    names, data = GaussianModel._read_ply(path)
    np.testing.assert_array_equal(data[:, names.index("f_rest_0")], m._features_rest[:, 0, 0].detach().numpy())
    np.testing.assert_array_equal(data[:, names.index("f_rest_15")], m._features_rest[:, 0, 1].detach().numpy())


def test_capture_restore(oracle_backend):
    m, cam, _ = _model(100, 32, 32)
    opt = OptimizationParams()
    training_step(m, cam, torch.rand(3, 32, 32), opt, PipelineParams(), torch.zeros(3), 8001)
    snap = m.capture()
    m2 = GaussianModel(3, device="cpu")
    m2.use_fused_adam = False
    m2.restore(snap, opt)
    assert torch.equal(m2.get_xyz, m.get_xyz) and m2.active_sh_degree == m.active_sh_degree
    assert torch.equal(m2.optimizer.state[m2._xyz]["exp_avg"], m.optimizer.state[m._xyz]["exp_avg"])


def test_create_from_pcd_uses_knn_for_scales():
    class PCD:
        points = np.random.default_rng(0).normal(size=(64, 3))
        colors = np.random.default_rng(1).uniform(size=(64, 3))
    from scipy.spatial import cKDTree
    d, _ = cKDTree(PCD.points).query(PCD.points, k=4)
    ref = torch.tensor((d[:, 1:] ** 2).mean(1), dtype=torch.float32)
    m = GaussianModel(3, device="cpu")
    m.create_from_pcd(PCD, 1.0, dist2_fn=lambda pts: ref)
    torch.testing.assert_close(m._scaling[:, 0], torch.log(torch.sqrt(ref)))
    assert m._scaling.shape == (64, 2) and m._features_rest.shape == (64, 15, 3) and m._features_dc.shape == (64, 1, 3)
    torch.testing.assert_close(m.get_opacity, torch.full((64, 1), 0.1))


def test_train_in_chunks_equals_one_uninterrupted_run(oracle_backend, tmp_path):
    """train_cli.py calls train() once per save point.  With one TrainState and `final_iteration` = the end of the
    schedule the split run must be the uninterrupted one bit for bit: every intermediate stop steps the optimiser
    (reference train.py:214-216 skips the step only on the schedule's last iteration) and the view sampler carries on.
    The checkpoint written at a stop reloads with the weights-only loader."""
    from gaussmart_amd.camera import Camera
    from gaussmart_amd.synthetic import jittered_cameras
    from gaussmart_amd.trainer import train, TrainState
    pipe, bg = PipelineParams(), torch.zeros(3)
    opt = OptimizationParams(iterations=8012)
    cams = jittered_cameras(3, 48, 48, seed=4, device="cpu")
    for i, c in enumerate(cams):
        c.original_image = torch.rand(3, 48, 48, generator=torch.Generator().manual_seed(i))

    def fresh():
        params, _ = make_scene(120, 48, 48, seed=6, device="cpu")
        m = GaussianModel(3, device="cpu"); m.use_fused_adam = False
        m.create_from_params(params); m.training_setup(opt)
        return m

    one = fresh()
    train(one, cams, opt, pipe, bg, first_iter=8000, iterations=8012)
    split = fresh()
    st = TrainState(seed=0)
    train(split, cams, opt, pipe, bg, first_iter=8000, iterations=8005, final_iteration=8012, state=st)
    torch.save((split.capture(), 8005), tmp_path / "chk.pth")
    snap, it = torch.load(tmp_path / "chk.pth", weights_only=True)
    assert it == 8005 and torch.equal(snap[1], split.get_xyz)
    train(split, cams, opt, pipe, bg, first_iter=8005, iterations=8012, final_iteration=8012, state=st)
    for a, b in zip(one.parameters(), split.parameters()):
        assert torch.equal(a, b)
    # the last iteration of the schedule leaves no stale gradient behind
    assert all(p.grad is None for p in split.parameters())
