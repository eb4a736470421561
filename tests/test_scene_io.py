"""COLMAP / NeRF-synthetic scene input vs what the reference's own loader read from the same files
(tests/golden/colmap_small*, captured by tests/golden/make_golden_colmap.py), plus the Scene container."""
import json
import os
import shutil

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from gaussmart_amd import scene_io as S
from gaussmart_amd.camera import getWorld2View2

G = np.load(os.path.join(GOLDEN, "colmap_small.npz"))
ROOT = os.path.join(GOLDEN, "colmap_small")


@pytest.mark.parametrize("tag", ["bin", "txt"])
def test_model_readers_match_reference(tag):
    d = os.path.join(ROOT, tag, "sparse", "0")
    ri, re_, rp = ((S.read_intrinsics_binary, S.read_extrinsics_binary, S.read_points3D_binary) if tag == "bin" else
                   (S.read_intrinsics_text, S.read_extrinsics_text, S.read_points3D_text))
    cams, imgs = ri(os.path.join(d, f"cameras.{tag}")), re_(os.path.join(d, f"images.{tag}"))
    xyz, rgb, err = rp(os.path.join(d, f"points3D.{tag}"))
    assert sorted(cams) == G[f"{tag}_cam_ids"].tolist()
    for k in cams:
        assert [cams[k].width, cams[k].height] == G[f"{tag}_cam{k}_wh"].tolist()
        np.testing.assert_array_equal(cams[k].params, G[f"{tag}_cam{k}_params"])
        assert cams[k].model == str(G[f"{tag}_cam{k}_model"])
    assert sorted(imgs) == G[f"{tag}_img_ids"].tolist()
    for k, im in imgs.items():
        np.testing.assert_array_equal(im.qvec, G[f"{tag}_img{k}_qvec"])
        np.testing.assert_array_equal(im.tvec, G[f"{tag}_img{k}_tvec"])
        assert im.camera_id == int(G[f"{tag}_img{k}_cam"]) and im.name == str(G[f"{tag}_img{k}_name"])
        np.testing.assert_array_equal(im.xys.reshape(-1, 2), G[f"{tag}_img{k}_xys"])
        np.testing.assert_array_equal(im.point3D_ids, G[f"{tag}_img{k}_p3d"])
        np.testing.assert_allclose(S.qvec2rotmat(im.qvec), G[f"{tag}_img{k}_R"], atol=1e-15)
    np.testing.assert_array_equal(xyz, G[f"{tag}_xyz"])
    np.testing.assert_array_equal(rgb, G[f"{tag}_rgb"])
    np.testing.assert_array_equal(err, G[f"{tag}_err"])


def test_colmap_scene_info_and_ply_roundtrip(tmp_path):
    root = tmp_path / "scene"
    shutil.copytree(os.path.join(ROOT, "bin"), root)
    info = S.readColmapSceneInfo(str(root), eval=True, llffhold=2, open_images=False)
    names = sorted(f"img_{i:03d}" for i in range(5))
    assert [c.image_name for c in info.test_cameras] == names[0::2]
    assert [c.image_name for c in info.train_cameras] == names[1::2]
    # camera convention: R = qvec2rotmat(q)^T, T = tvec; focal -> fov per model (PINHOLE / SIMPLE_PINHOLE)
    c = {ci.image_name: ci for ci in info.train_cameras + info.test_cameras}["img_000"]
    np.testing.assert_allclose(c.R, G["bin_img1_R"].T)
    np.testing.assert_allclose(c.FovX, 2 * np.arctan(640 / (2 * 500.0)))
    np.testing.assert_allclose(c.FovY, 2 * np.arctan(480 / (2 * 510.0)))
    c2 = {ci.image_name: ci for ci in info.train_cameras + info.test_cameras}["img_001"]     # SIMPLE_PINHOLE, 800x600, f=700
    np.testing.assert_allclose([c2.FovX, c2.FovY], [2 * np.arctan(800 / 1400.0), 2 * np.arctan(600 / 1400.0)])
    # NeRF++ normalisation: radius = 1.1 * max distance of a train camera centre from their mean
    centers = np.stack([np.linalg.inv(getWorld2View2(ci.R, ci.T))[:3, 3] for ci in info.train_cameras])
    np.testing.assert_allclose(info.nerf_normalization["radius"], 1.1 * np.linalg.norm(centers - centers.mean(0), axis=1).max(), rtol=1e-6)
    np.testing.assert_allclose(info.nerf_normalization["translate"], -centers.mean(0), rtol=1e-6, atol=1e-7)
    # the converted point cloud keeps positions (f32) and colours (u8 / 255), segments = 0
    np.testing.assert_array_equal(info.point_cloud.points, G["bin_xyz"].astype(np.float32))
    np.testing.assert_allclose(info.point_cloud.colors, G["bin_rgb"] / 255.0)
    assert os.path.exists(info.ply_path) and np.all(info.point_cloud.segments == 0)
    head = open(info.ply_path, "rb").read(400).decode("ascii", "replace")
    assert "property uchar red" in head and "property int segment" in head and "property float nx" in head


def test_nerf_synthetic_reader_and_scene(tmp_path):
    from PIL import Image
    from scipy.spatial import cKDTree
    from gaussmart_amd.gaussian_model import GaussianModel
    root = tmp_path / "lego"
    os.makedirs(root / "train"); os.makedirs(root / "test")
    rng = np.random.default_rng(0)

    def frames(split, n):
        out = []
        for i in range(n):
            m = np.eye(4); m[:3, 3] = rng.normal(size=3) * 2 + np.array([0, 0, 4.0])
            Image.fromarray(rng.integers(0, 255, size=(40, 60, 4), dtype=np.uint8), "RGBA").save(root / split / f"r_{i}.png")
            out.append({"file_path": f"./{split}/r_{i}", "transform_matrix": m.tolist()})
        return {"camera_angle_x": 0.7, "frames": out}
    json.dump(frames("train", 4), open(root / "transforms_train.json", "w"))
    json.dump(frames("test", 2), open(root / "transforms_test.json", "w"))
    info = S.load_scene_info(str(root), eval=True, white_background=True)
    assert len(info.train_cameras) == 4 and len(info.test_cameras) == 2 and info.point_cloud.points.shape == (100_000, 3)
    c = info.train_cameras[0]
    assert (c.width, c.height) == (60, 40) and abs(c.FovX - 0.7) < 1e-12
    # Blender -> COLMAP axes
    w2c = np.eye(4); w2c[:3, :3] = c.R.T; w2c[:3, 3] = c.T
    m = np.array(json.load(open(root / "transforms_train.json"))["frames"][0]["transform_matrix"]); m[:3, 1:3] *= -1
    np.testing.assert_allclose(w2c, np.linalg.inv(m), atol=1e-12)

    def dist2(pts):          # CPU stand-in for the HIP 3-NN kernel (test only)
        p = pts.cpu().numpy().astype(np.float64)
        d, _ = cKDTree(p).query(p, k=4)
        return torch.tensor((d[:, 1:] ** 2).mean(1), dtype=torch.float32)
    g = GaussianModel(3, device="cpu")
    sc = S.Scene(str(root), g, eval=True, white_background=True, data_device="cpu", resolution_scales=(1.0, 2.0), dist2_fn=dist2)
    assert len(sc.getTrainCameras()) == 4 and sc.getTrainCameras(2.0)[0].image_width == 30
    cam = sc.getTrainCameras()[0]
    assert cam.original_image.shape == (3, 40, 60) and float(cam.original_image.max()) <= 1.0
    assert sc.cameras_extent > 0 and g.get_xyz.shape == (100_000, 3) and g.spatial_lr_scale == sc.cameras_extent
    # save / reload an iteration
    sc.model_path = str(tmp_path / "out")
    sc.save(7)
    g2 = GaussianModel(3, device="cpu")
    sc2 = S.Scene(str(root), g2, model_path=str(tmp_path / "out"), load_iteration=-1, eval=True, white_background=True,
                  data_device="cpu", dist2_fn=dist2)
    assert sc2.loaded_iter == 7 and torch.equal(g2._xyz, g._xyz)
