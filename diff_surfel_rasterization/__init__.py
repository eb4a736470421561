"""Drop-in for the reference's `diff_surfel_rasterization` package
(imported at gaussian_renderer/__init__.py:14 of alevalve/gaussmart): put this repository's root
on PYTHONPATH in place of `pip install submodules/diff-surfel-rasterization`."""
from gaussmart_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                      rasterize_gaussians)
