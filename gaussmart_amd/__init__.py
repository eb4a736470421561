"""gaussmart_amd -- MI355X-native differentiable 2D-Gaussian-surfel rasterizer + 3-NN kernel.

Drop-in for the two native operators alevalve/gaussmart trains against
(`diff_surfel_rasterization`, `simple_knn._C.distCUDA2`; call sites
gaussian_renderer/__init__.py:14,37-53,97-106 and scene/gaussian_model.py:22,261 of the
reference), plus device-parametric counterparts of their immediate callers.
The compute path is hand-written HIP for gfx950 behind a C ABI (include/gsr.h); there is no
CPU fallback: importing the operators without the built library raises.
"""
__version__ = "0.1.0"
