"""Scene input: COLMAP sparse models (binary / text), NeRF-synthetic `transforms_*.json`, the
point-cloud PLY the reference seeds Gaussians from, and the `Scene` container.

Counterpart of scene/colmap_loader.py:43-242, scene/dataset_readers.py:26-341,
utils/camera_utils.py:19-82 and scene/__init__.py:21-94 of the reference.  File formats are COLMAP's
(public); the readers are checked against the reference's own readers on small committed models
(tests/golden/colmap_*/, tests/test_scene_io.py).  The SAM/segment-specific branches of the
reference (cleaned point cloud lookup, mask areas) are out of scope; a `segment` column is carried
through when present.
"""
import json
import os
import struct
from pathlib import Path
from typing import NamedTuple, Optional

import numpy as np
import torch

from .camera import Camera, getWorld2View2, focal2fov, fov2focal
from .sh import SH2RGB

# model id -> (name, number of parameters)   [COLMAP src/base/camera_models.h]
CAMERA_MODELS = {0: ("SIMPLE_PINHOLE", 3), 1: ("PINHOLE", 4), 2: ("SIMPLE_RADIAL", 4), 3: ("RADIAL", 5),
                 4: ("OPENCV", 8), 5: ("OPENCV_FISHEYE", 8), 6: ("FULL_OPENCV", 12), 7: ("FOV", 5),
                 8: ("SIMPLE_RADIAL_FISHEYE", 4), 9: ("RADIAL_FISHEYE", 5), 10: ("THIN_PRISM_FISHEYE", 12)}


class ColmapCamera(NamedTuple):
    id: int
    model: str
    width: int
    height: int
    params: np.ndarray


class ColmapImage(NamedTuple):
    id: int
    qvec: np.ndarray
    tvec: np.ndarray
    camera_id: int
    name: str
    xys: np.ndarray
    point3D_ids: np.ndarray


def qvec2rotmat(q):
    w, x, y, z = (float(v) for v in q)
    return np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y],
                     [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
                     [2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]])


# ----------------------------------------------------------------------------------- binary model
def read_intrinsics_binary(path):
    cams = {}
    buf = memoryview(Path(path).read_bytes())
    (n,), off = struct.unpack_from("<Q", buf, 0), 8
    for _ in range(n):
        cam_id, model_id, width, height = struct.unpack_from("<iiQQ", buf, off)
        off += 24
        name, n_par = CAMERA_MODELS[model_id]
        params = np.frombuffer(buf, dtype="<f8", count=n_par, offset=off).copy()
        off += 8 * n_par
        cams[cam_id] = ColmapCamera(cam_id, name, int(width), int(height), params)
    return cams


def read_extrinsics_binary(path):
    imgs = {}
    raw = Path(path).read_bytes()
    buf = memoryview(raw)
    (n,), off = struct.unpack_from("<Q", buf, 0), 8
    for _ in range(n):
        vals = struct.unpack_from("<idddddddi", buf, off)
        off += 64
        end = raw.index(b"\x00", off)
        name = raw[off:end].decode("utf-8")
        off = end + 1
        (n2d,) = struct.unpack_from("<Q", buf, off)
        off += 8
        rec = np.frombuffer(buf, dtype=np.dtype([("x", "<f8"), ("y", "<f8"), ("id", "<i8")]), count=n2d, offset=off)
        off += 24 * n2d
        imgs[vals[0]] = ColmapImage(vals[0], np.array(vals[1:5]), np.array(vals[5:8]), vals[8], name,
                                    np.column_stack([rec["x"], rec["y"]]) if n2d else np.zeros((0, 2)),
                                    rec["id"].astype(np.int64).copy())
    return imgs


def read_points3D_binary(path):
    buf = memoryview(Path(path).read_bytes())
    (n,), off = struct.unpack_from("<Q", buf, 0), 8
    xyz, rgb, err = np.empty((n, 3)), np.empty((n, 3)), np.empty((n, 1))
    for i in range(n):
        v = struct.unpack_from("<QdddBBBd", buf, off)
        off += 43
        (track,) = struct.unpack_from("<Q", buf, off)
        off += 8 + 8 * track
        xyz[i], rgb[i], err[i] = v[1:4], v[4:7], v[7]
    return xyz, rgb, err


# ----------------------------------------------------------------------------------- text model
def _data_lines(path):
    with open(path, "r") as f:
        for line in f:
            line = line.strip()
            if line and not line.startswith("#"):
                yield line


def read_intrinsics_text(path):
    cams = {}
    for line in _data_lines(path):
        e = line.split()
        if e[1] != "PINHOLE":
            raise AssertionError("While the loader support other types, the rest of the code assumes PINHOLE")
        cams[int(e[0])] = ColmapCamera(int(e[0]), e[1], int(e[2]), int(e[3]), np.array([float(v) for v in e[4:]]))
    return cams


def read_extrinsics_text(path):
    imgs = {}
    with open(path, "r") as f:
        lines = [l.rstrip("\n") for l in f]
    i = 0
    while i < len(lines):
        line = lines[i].strip()
        i += 1
        if not line or line.startswith("#"):
            continue
        e = line.split()
        pts = lines[i].split() if i < len(lines) else []
        i += 1
        xys = np.column_stack([[float(v) for v in pts[0::3]], [float(v) for v in pts[1::3]]]) if pts else np.zeros((0, 2))
        imgs[int(e[0])] = ColmapImage(int(e[0]), np.array([float(v) for v in e[1:5]]), np.array([float(v) for v in e[5:8]]),
                                      int(e[8]), e[9], xys, np.array([int(v) for v in pts[2::3]], dtype=np.int64))
    return imgs


def read_points3D_text(path):
    rows = [l.split() for l in _data_lines(path)]
    n = len(rows)
    xyz, rgb, err = np.empty((n, 3)), np.empty((n, 3)), np.empty((n, 1))
    for i, e in enumerate(rows):
        xyz[i] = [float(v) for v in e[1:4]]
        rgb[i] = [int(v) for v in e[4:7]]
        err[i] = float(e[7])
    return xyz, rgb, err


# ----------------------------------------------------------------------------------- point-cloud PLY
class BasicPointCloud(NamedTuple):
    points: np.ndarray
    colors: np.ndarray
    normals: np.ndarray
    segments: np.ndarray


_PLY_TYPES = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "uint8": "u1",
              "int": "<i4", "int32": "<i4", "uint": "<u4", "short": "<i2", "ushort": "<u2", "char": "i1"}


def read_ply_vertices(path):
    """-> structured array of the `vertex` element of a binary_little_endian PLY."""
    with open(path, "rb") as f:
        fields, count, fmt, in_vertex = [], 0, None, False
        while True:
            line = f.readline()
            if not line:
                raise ValueError("PLY header without end_header")
            t = line.decode("ascii", "replace").split()
            if not t:
                continue
            if t[0] == "format":
                fmt = t[1]
            elif t[0] == "element":
                in_vertex = t[1] == "vertex"
                if in_vertex:
                    count = int(t[2])
            elif t[0] == "property" and in_vertex:
                if t[1] == "list":
                    raise ValueError("list properties on vertices are not supported")
                fields.append((t[2], _PLY_TYPES[t[1]]))
            elif t[0] == "end_header":
                break
        if fmt != "binary_little_endian":
            raise ValueError("only binary_little_endian PLY files are supported")
        dt = np.dtype(fields)
        return np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count)


def fetchPly(path):
    v = read_ply_vertices(path)
    pos = np.vstack([v["x"], v["y"], v["z"]]).T
    col = np.vstack([v["red"], v["green"], v["blue"]]).T / 255.0
    nrm = np.vstack([v["nx"], v["ny"], v["nz"]]).T if "nx" in v.dtype.names else np.zeros_like(pos)
    seg = v["segment"].astype(np.int32) if "segment" in v.dtype.names else np.zeros(len(pos), dtype=np.int32)
    return BasicPointCloud(points=pos, colors=col, normals=nrm, segments=seg)


def storePly(path, xyz, rgb, segments=None):
    """x y z nx ny nz (f4) red green blue (u1) segment (i4): scene/dataset_readers.py:169-184."""
    n = len(xyz)
    dt = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4"),
                   ("red", "u1"), ("green", "u1"), ("blue", "u1"), ("segment", "<i4")])
    el = np.zeros(n, dtype=dt)
    el["x"], el["y"], el["z"] = np.asarray(xyz, dtype=np.float64).T
    el["red"], el["green"], el["blue"] = np.asarray(rgb).astype(np.uint8).T
    if segments is not None:
        el["segment"] = segments
    names = {"<f4": "float", "u1": "uchar", "<i4": "int"}
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n
    header += "".join(f"property {names[dt.fields[k][0].str.replace('|', '')]} {k}\n" for k in dt.names) + "end_header\n"
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(el.tobytes())


# ----------------------------------------------------------------------------------- scene description
class CameraInfo(NamedTuple):
    uid: int
    R: np.ndarray
    T: np.ndarray
    FovY: float
    FovX: float
    image: object          # PIL image (or None when images are absent)
    image_path: str
    image_name: str
    width: int
    height: int


class SceneInfo(NamedTuple):
    point_cloud: Optional[BasicPointCloud]
    train_cameras: list
    test_cameras: list
    nerf_normalization: dict
    ply_path: str


def getNerfppNorm(cam_infos):
    """Scene centre / radius from the camera centres (scene/dataset_readers.py:45-66)."""
    centers = np.hstack([np.linalg.inv(getWorld2View2(c.R, c.T))[:3, 3:4] for c in cam_infos])
    center = centers.mean(axis=1, keepdims=True)
    diagonal = np.linalg.norm(centers - center, axis=0).max()
    return {"translate": -center.flatten(), "radius": diagonal * 1.1}


def readColmapCameras(cam_extrinsics, cam_intrinsics, images_folder, open_images=True):
    infos = []
    for key in cam_extrinsics:
        ext = cam_extrinsics[key]
        intr = cam_intrinsics[ext.camera_id]
        R = qvec2rotmat(ext.qvec).T
        T = np.array(ext.tvec)
        if intr.model == "SIMPLE_PINHOLE":
            fovy, fovx = focal2fov(intr.params[0], intr.height), focal2fov(intr.params[0], intr.width)
        elif intr.model == "PINHOLE":
            fovy, fovx = focal2fov(intr.params[1], intr.height), focal2fov(intr.params[0], intr.width)
        else:
            raise AssertionError("Colmap camera model not handled: only undistorted datasets (PINHOLE or "
                                 "SIMPLE_PINHOLE cameras) supported!")
        image_path = os.path.join(images_folder, os.path.basename(ext.name))
        image = None
        if open_images:
            from PIL import Image
            image = Image.open(image_path)
        infos.append(CameraInfo(intr.id, R, T, fovy, fovx, image, image_path,
                                os.path.basename(image_path).split(".")[0], intr.width, intr.height))
    return infos


def _create_once(ply_path, make):
    """The converted point cloud is written by ONE process: under torchrun every rank loads the scene, and concurrent
    writers / readers of the same file race.  Rank 0 writes (to a temporary name, then renames), the others wait at a
    barrier before reading."""
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if not multi or dist.get_rank() == 0:
        if not os.path.exists(ply_path):
            tmp = ply_path + f".tmp{os.getpid()}"
            make(tmp)
            os.replace(tmp, ply_path)
    if multi:
        dist.barrier()


def readColmapSceneInfo(path, images=None, eval=False, llffhold=8, open_images=True):
    sparse = os.path.join(path, "sparse/0")
    if os.path.exists(os.path.join(sparse, "images.bin")):
        ext, intr = read_extrinsics_binary(os.path.join(sparse, "images.bin")), read_intrinsics_binary(os.path.join(sparse, "cameras.bin"))
    else:
        ext, intr = read_extrinsics_text(os.path.join(sparse, "images.txt")), read_intrinsics_text(os.path.join(sparse, "cameras.txt"))
    cams = sorted(readColmapCameras(ext, intr, os.path.join(path, images or "images"), open_images), key=lambda c: c.image_name)
    if eval:
        train = [c for i, c in enumerate(cams) if i % llffhold != 0]
        test = [c for i, c in enumerate(cams) if i % llffhold == 0]
    else:
        train, test = cams, []
    ply_path = os.path.join(sparse, "points3D.ply")
    def make(dst):
        try:
            xyz, rgb, _ = read_points3D_binary(os.path.join(sparse, "points3D.bin"))
        except FileNotFoundError:
            xyz, rgb, _ = read_points3D_text(os.path.join(sparse, "points3D.txt"))
        storePly(dst, xyz, rgb, np.zeros(len(xyz), dtype=np.int32))
    _create_once(ply_path, make)
    return SceneInfo(fetchPly(ply_path), train, test, getNerfppNorm(train), ply_path)


def readCamerasFromTransforms(path, transformsfile, white_background, extension=".png", open_images=True):
    infos = []
    with open(os.path.join(path, transformsfile)) as f:
        contents = json.load(f)
    fovx = contents["camera_angle_x"]
    for idx, frame in enumerate(contents["frames"]):
        c2w = np.array(frame["transform_matrix"], dtype=np.float64)
        c2w[:3, 1:3] *= -1                               # OpenGL/Blender axes -> COLMAP axes
        w2c = np.linalg.inv(c2w)
        R, T = w2c[:3, :3].T, w2c[:3, 3]
        image_path = os.path.join(path, frame["file_path"] + extension)
        image, (w, h) = None, (contents.get("w", 800), contents.get("h", 800))
        if open_images:
            from PIL import Image
            rgba = np.array(Image.open(image_path).convert("RGBA")) / 255.0
            bg = np.ones(3) if white_background else np.zeros(3)
            arr = rgba[:, :, :3] * rgba[:, :, 3:4] + bg * (1 - rgba[:, :, 3:4])
            image = Image.fromarray(np.array(arr * 255.0, dtype=np.uint8), "RGB")
            w, h = image.size
        infos.append(CameraInfo(idx, R, T, focal2fov(fov2focal(fovx, w), h), fovx, image, image_path,
                                Path(image_path).stem, w, h))
    return infos


def readNerfSyntheticInfo(path, white_background=False, eval=False, extension=".png", open_images=True, seed=0):
    train = readCamerasFromTransforms(path, "transforms_train.json", white_background, extension, open_images)
    test = readCamerasFromTransforms(path, "transforms_test.json", white_background, extension, open_images)
    if not eval:
        train, test = train + test, []
    ply_path = os.path.join(path, "points3d.ply")
    def make(dst):
        rng = np.random.default_rng(seed)                # random points inside the Blender scene bounds
        xyz = rng.random((100_000, 3)) * 2.6 - 1.3
        storePly(dst, xyz, SH2RGB(rng.random((100_000, 3)) / 255.0) * 255)
    _create_once(ply_path, make)
    return SceneInfo(fetchPly(ply_path), train, test, getNerfppNorm(train), ply_path)


def load_scene_info(path, images=None, eval=False, white_background=False, open_images=True):
    if os.path.exists(os.path.join(path, "sparse")):
        return readColmapSceneInfo(path, images, eval, open_images=open_images)
    if os.path.exists(os.path.join(path, "transforms_train.json")):
        return readNerfSyntheticInfo(path, white_background, eval, open_images=open_images)
    raise ValueError(f"could not recognise the scene type of {path}")


# ----------------------------------------------------------------------------------- cameras / Scene
def pil_to_torch(pil_image, resolution):
    t = torch.from_numpy(np.array(pil_image.resize(resolution))) / 255.0
    return t.permute(2, 0, 1) if t.dim() == 3 else t.unsqueeze(-1).permute(2, 0, 1)


def load_camera(cam_info, uid, resolution=-1, resolution_scale=1.0, data_device="cuda"):
    """utils/camera_utils.py:19-54: explicit downscale factors 1/2/4/8, or width capped at 1600."""
    w, h = cam_info.image.size if cam_info.image is not None else (cam_info.width, cam_info.height)
    if resolution in (1, 2, 4, 8):
        res = (round(w / (resolution_scale * resolution)), round(h / (resolution_scale * resolution)))
    else:
        down = (w / 1600 if w > 1600 else 1.0) if resolution == -1 else w / resolution
        res = (int(w / (down * resolution_scale)), int(h / (down * resolution_scale)))
    image = alpha = None
    if cam_info.image is not None:
        img = pil_to_torch(cam_info.image, res)
        image, alpha = img[:3], (img[3:4] if img.shape[0] == 4 else None)
    return Camera(cam_info.uid, cam_info.R, cam_info.T, cam_info.FovX, cam_info.FovY, image, alpha, cam_info.image_name,
                  uid, data_device=data_device, width=res[0], height=res[1])


class Scene:
    """scene/__init__.py:21-94: cameras at the requested resolution scales, cameras_extent, and a
    GaussianModel seeded from the point cloud (or loaded from a saved iteration)."""

    def __init__(self, source_path, gaussians, model_path=None, load_iteration=None, images=None, eval=False,
                 white_background=False, resolution=-1, resolution_scales=(1.0,), data_device="cuda", shuffle=True, seed=0,
                 dist2_fn=None):
        self.model_path, self.gaussians, self.loaded_iter = model_path, gaussians, None
        info = load_scene_info(source_path, images, eval, white_background)
        self.cameras_extent = info.nerf_normalization["radius"]
        if load_iteration is not None and model_path:
            if load_iteration == -1:
                its = [int(d.split("_")[-1]) for d in os.listdir(os.path.join(model_path, "point_cloud"))]
                load_iteration = max(its)
            self.loaded_iter = load_iteration
        train, test = list(info.train_cameras), list(info.test_cameras)
        if shuffle:
            import random
            rnd = random.Random(seed)
            rnd.shuffle(train)
            rnd.shuffle(test)
        self.train_cameras = {s: [load_camera(c, i, resolution, s, data_device) for i, c in enumerate(train)] for s in resolution_scales}
        self.test_cameras = {s: [load_camera(c, i, resolution, s, data_device) for i, c in enumerate(test)] for s in resolution_scales}
        if self.loaded_iter:
            gaussians.load_ply(os.path.join(model_path, "point_cloud", f"iteration_{self.loaded_iter}", "point_cloud.ply"))
        else:
            gaussians.create_from_pcd(info.point_cloud, self.cameras_extent, dist2_fn=dist2_fn)

    def save(self, iteration):
        self.gaussians.save_ply(os.path.join(self.model_path, "point_cloud", f"iteration_{iteration}", "point_cloud.ply"))

    def getTrainCameras(self, scale=1.0):
        return self.train_cameras[scale]

    def getTestCameras(self, scale=1.0):
        return self.test_cameras[scale]
