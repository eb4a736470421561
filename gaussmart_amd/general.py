"""Small numeric helpers shared by the model and the trainer.

Counterparts of utils/general_utils.py:18-110 of the reference (device is a parameter there the
reference hard-codes "cuda"); `inverse_sigmoid` and `get_expon_lr_func` are checked against
golden vectors.
"""
import numpy as np
import torch


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """Log-linear interpolation lr_init -> lr_final with an optional sine warm-up."""

    def helper(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        delay = 1.0
        if lr_delay_steps > 0:
            delay = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
        t = np.clip(step / max_steps, 0, 1)
        return delay * np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t)

    return helper


def build_rotation(r: torch.Tensor) -> torch.Tensor:
    """(w,x,y,z) quaternions [N,4] (normalised here) -> rotation matrices [N,3,3]."""
    q = r / r.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(-1, 3, 3)


def build_scaling_rotation(s: torch.Tensor, r: torch.Tensor) -> torch.Tensor:
    """R @ diag(s) for s [N,3]."""
    return build_rotation(r) * s[:, None, :]
