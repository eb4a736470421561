"""Real spherical-harmonics helpers (degree <= 3).

Counterpart of the reference's utils/sh_utils.py:24-117; checked against golden vectors of
`eval_sh`, `RGB2SH`, `SH2RGB` in tests/test_golden.py.  Written as basis * coefficients so the
same basis function serves the HIP kernel's documentation and the Python colour path.
"""
import torch

Y00 = 0.28209479177387814
Y1 = 0.4886025119029199
Y2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
      -1.0925484305920792, 0.5462742152960396)
Y3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
      -0.4570457994644658, 1.445305721320277, -0.5900435899266435)


def sh_basis(deg: int, dirs: torch.Tensor) -> torch.Tensor:
    """dirs [...,3] unit vectors -> basis [..., (deg+1)^2] with the reference's sign convention."""
    if not 0 <= deg <= 3:
        raise ValueError("SH degree must be in 0..3")
    x, y, z = dirs[..., 0], dirs[..., 1], dirs[..., 2]
    b = [torch.full_like(x, Y00)]
    if deg >= 1:
        b += [-Y1 * y, Y1 * z, -Y1 * x]
    if deg >= 2:
        xx, yy, zz = x * x, y * y, z * z
        b += [Y2[0] * x * y, Y2[1] * y * z, Y2[2] * (2 * zz - xx - yy), Y2[3] * x * z,
              Y2[4] * (xx - yy)]
    if deg >= 3:
        b += [Y3[0] * y * (3 * xx - yy), Y3[1] * x * y * z, Y3[2] * y * (4 * zz - xx - yy),
              Y3[3] * z * (2 * zz - 3 * xx - 3 * yy), Y3[4] * x * (4 * zz - xx - yy),
              Y3[5] * z * (xx - yy), Y3[6] * x * (xx - 3 * yy)]
    return torch.stack(b, dim=-1)


def eval_sh(deg: int, sh: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """sh [..., C, K>=(deg+1)^2], dirs [..., 3] -> [..., C] (same contract as the reference)."""
    k = (deg + 1) ** 2
    return (sh[..., :k] * sh_basis(deg, dirs)[..., None, :]).sum(-1)


def RGB2SH(rgb):
    return (rgb - 0.5) / Y00


def SH2RGB(sh):
    return sh * Y00 + 0.5
