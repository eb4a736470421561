"""Host-side helpers vs golden vectors captured from the reference's CPU-importable functions
(tests/golden/make_golden.py; reference file:line cited there)."""
import os

import numpy as np
import torch

from conftest import GOLDEN
from gaussmart_amd import camera, general, losses, sh
from oracle import surfel_ref as O


def test_eval_sh_matches_reference():
    g = np.load(os.path.join(GOLDEN, "sh.npz"))
    s, d = torch.from_numpy(g["sh"]), torch.from_numpy(g["dirs"])
    for deg in range(4):
        np.testing.assert_allclose(sh.eval_sh(deg, s, d).numpy(), g[f"eval_deg{deg}"], rtol=0, atol=1e-13)
    rgb = torch.from_numpy(g["rgb"])
    np.testing.assert_allclose(sh.RGB2SH(rgb).numpy(), g["rgb2sh"], atol=1e-15)
    np.testing.assert_allclose(sh.SH2RGB(rgb).numpy(), g["sh2rgb"], atol=1e-15)


def test_oracle_sh_colour_matches_reference_eval_sh():
    """The oracle's SH->RGB (same formula the HIP kernel implements) against the reference's
    eval_sh + 0.5 clamp (gaussian_renderer/__init__.py:88-91)."""
    g = np.load(os.path.join(GOLDEN, "sh.npz"))
    s, d = torch.from_numpy(g["sh"]), torch.from_numpy(g["dirs"])   # [n,3,16], unit dirs
    campos = torch.tensor([0.3, -0.2, 0.1], dtype=torch.float64)
    means = campos + 2.5 * d
    for deg in range(4):
        rgb, clamped = O.eval_sh_rgb(deg, s.transpose(1, 2).contiguous(), means, campos)
        ref = np.maximum(g[f"eval_deg{deg}"] + 0.5, 0.0)
        np.testing.assert_allclose(rgb.numpy(), ref, atol=1e-12)
        assert np.array_equal(clamped.numpy(), (g[f"eval_deg{deg}"] + 0.5) < 0)


def test_camera_matrices_match_reference():
    c = np.load(os.path.join(GOLDEN, "camera.npz"))
    for i in range(4):
        w = camera.getWorld2View2(c[f"R{i}"], c[f"T{i}"], c[f"trans{i}"], float(c[f"scale{i}"]))
        np.testing.assert_array_equal(w, c[f"w2v{i}"])
        p = camera.getProjectionMatrix(0.01, 100.0, *c[f"fov{i}"]).numpy()
        np.testing.assert_allclose(p, c[f"proj{i}"], rtol=1e-7, atol=0)
        fx, fy = c[f"fov{i}"]
        np.testing.assert_allclose([camera.fov2focal(fx, 640), camera.focal2fov(camera.fov2focal(fy, 480), 480)],
                                   c[f"focal{i}"], rtol=1e-12)


def test_camera_class_conventions():
    c = np.load(os.path.join(GOLDEN, "camera.npz"))
    cam = camera.Camera(0, c["R1"], c["T1"], 0.8, 0.65, None, data_device="cpu", width=64, height=48,
                        trans=c["trans1"], scale=float(c["scale1"]))
    np.testing.assert_array_equal(cam.world_view_transform.numpy(), c["w2v1"].T)
    P = camera.getProjectionMatrix(0.01, 100.0, 0.8, 0.65)
    np.testing.assert_allclose(cam.full_proj_transform.numpy(), c["w2v1"].T @ P.numpy().T, rtol=1e-6, atol=1e-7)
    # clip w equals view-space z (P[3,2] = 1)
    pt = torch.tensor([0.3, -0.4, 2.0, 1.0])
    assert abs(float((pt @ cam.full_proj_transform)[3] - (pt @ cam.world_view_transform)[2])) < 1e-6
    np.testing.assert_allclose(cam.camera_center.numpy(), np.linalg.inv(c["w2v1"].T)[3, :3], atol=1e-6)


def test_losses_match_reference():
    g = np.load(os.path.join(GOLDEN, "loss.npz"))
    a, b = torch.from_numpy(g["a"]), torch.from_numpy(g["b"])
    np.testing.assert_allclose(losses.l1_loss(a, b).numpy(), g["l1"], rtol=1e-6)
    np.testing.assert_allclose(losses.ssim(a, b).numpy(), g["ssim"], rtol=1e-5)
    np.testing.assert_allclose(losses.ssim(a[None], b[None], size_average=False).numpy(), g["ssim_per"], rtol=1e-5)
    np.testing.assert_allclose(losses.psnr(a[None], b[None]).numpy(), g["psnr"], rtol=1e-6)


def test_scalar_helpers_match_reference():
    g = np.load(os.path.join(GOLDEN, "scalar.npz"))
    np.testing.assert_allclose(general.inverse_sigmoid(torch.from_numpy(g["x"])).numpy(), g["inv_sigmoid"], rtol=1e-6)
    f = general.get_expon_lr_func(lr_init=0.00016, lr_final=0.0000016, lr_delay_mult=0.01, max_steps=30000)
    f2 = general.get_expon_lr_func(lr_init=1e-2, lr_final=1e-4, lr_delay_steps=500, lr_delay_mult=0.1, max_steps=2000)
    np.testing.assert_allclose([f(int(s)) for s in g["steps"]], g["lr"], rtol=1e-12)
    np.testing.assert_allclose([f2(int(s)) for s in g["steps"]], g["lr2"], rtol=1e-12)


def test_quaternion_convention_wxyz():
    # utils/general_utils.py:78-99: (w,x,y,z); a 90 degree turn about z maps x to y
    q = torch.tensor([[np.cos(np.pi / 4), 0.0, 0.0, np.sin(np.pi / 4)]], dtype=torch.float64)
    for R in (general.build_rotation(q)[0], O.quat_to_rotmat(q)[0]):
        np.testing.assert_allclose((R @ torch.tensor([1.0, 0, 0], dtype=torch.float64)).numpy(), [0, 1, 0], atol=1e-12)
