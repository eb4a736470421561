"""Camera matrices in the reference's transposed / row-vector convention.

Counterparts of utils/graphics_utils.py:39-72 and scene/cameras.py:17-72 of the reference:
`world_view_transform = W2C^T`, `full_proj_transform = world_view_transform @ P^T`,
`P[3,2] = 1` (clip w = view z), `camera_center = inverse(world_view_transform)[3,:3]`.
Device is a parameter (the reference hard-codes "cuda").
"""
import math

import numpy as np
import torch

ZNEAR, ZFAR = 0.01, 100.0


def getWorld2View2(R, t, translate=np.zeros(3), scale=1.0):
    """R is camera-to-world rotation (COLMAP R^T), t world-to-camera translation."""
    w2c = np.eye(4)
    w2c[:3, :3] = np.asarray(R).T
    w2c[:3, 3] = t
    c2w = np.linalg.inv(w2c)
    c2w[:3, 3] = (c2w[:3, 3] + translate) * scale
    return np.linalg.inv(c2w).astype(np.float32)


def getProjectionMatrix(znear, zfar, fovX, fovY):
    tx, ty = math.tan(fovX / 2), math.tan(fovY / 2)
    right, top = tx * znear, ty * znear
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (2 * right)
    P[1, 1] = 2.0 * znear / (2 * top)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def fov2focal(fov, pixels):
    return pixels / (2 * math.tan(fov / 2))


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))


class MiniCam:
    """Everything render() reads from a camera (scene/cameras.py:61-72)."""

    def __init__(self, width, height, fovy, fovx, znear, zfar, world_view_transform,
                 full_proj_transform):
        self.image_width, self.image_height = width, height
        self.FoVy, self.FoVx = fovy, fovx
        self.znear, self.zfar = znear, zfar
        self.world_view_transform = world_view_transform
        self.full_proj_transform = full_proj_transform
        self.camera_center = torch.inverse(world_view_transform)[3][:3].contiguous()


class Camera(torch.nn.Module):
    """scene/cameras.py:17-59 with an explicit device."""

    def __init__(self, colmap_id, R, T, FoVx, FoVy, image, gt_alpha_mask=None, image_name="",
                 uid=0, trans=np.zeros(3), scale=1.0, data_device="cuda", width=None, height=None,
                 keep_image_on_device=True):
        super().__init__()
        self.uid, self.colmap_id, self.image_name = uid, colmap_id, image_name
        self.R, self.T, self.FoVx, self.FoVy = R, T, FoVx, FoVy
        self.data_device = torch.device(data_device)
        if image is not None:
            # DEVIATION from scene/cameras.py:40, which keeps the clamped image on the HOST ("move to device at dataloader
            # to reduce VRAM requirement") and lets train.py:112 copy it up every iteration: with 288 GB of HBM the
            # images stay resident by default (a 1600x1200 RGB frame is 23 MB, so ~40 frames per GB);
            # keep_image_on_device=False restores the reference's behaviour (trainer.train() moves the frame per iteration)
            self.original_image = image.clamp(0.0, 1.0)
            if keep_image_on_device:
                self.original_image = self.original_image.to(self.data_device)
            self.image_width, self.image_height = image.shape[2], image.shape[1]
        else:
            self.original_image = None
            self.image_width, self.image_height = width, height
        self.gt_alpha_mask = gt_alpha_mask.to(self.data_device) if gt_alpha_mask is not None else None
        self.zfar, self.znear = ZFAR, ZNEAR
        self.trans, self.scale = trans, scale
        dev = self.data_device
        self.world_view_transform = torch.tensor(getWorld2View2(R, T, trans, scale)).transpose(0, 1).contiguous().to(dev)
        self.projection_matrix = getProjectionMatrix(self.znear, self.zfar, FoVx, FoVy).transpose(0, 1).to(dev)
        self.full_proj_transform = self.world_view_transform @ self.projection_matrix
        self.camera_center = self.world_view_transform.inverse()[3, :3].contiguous()


def look_at_camera(eye, target, up, fovx, width, height, device="cpu", image=None, uid=0):
    """Convenience constructor for synthetic scenes (+z forward, +y down, as COLMAP)."""
    eye, target, up = (np.asarray(v, dtype=np.float64) for v in (eye, target, up))
    fwd = target - eye
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, up)                       # x_cam = right, y_cam = down, z_cam = forward
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd], axis=1)        # camera-to-world
    T = -R.T @ eye
    fovy = focal2fov(fov2focal(fovx, width), height)
    return Camera(uid, R, T, fovx, fovy, image, None, f"synthetic_{uid}", uid, data_device=device,
                  width=width, height=height)
