"""Default hyper-parameters, copied as VALUES from the reference's argparse groups
(arguments/__init__.py:47-96 of alevalve/gaussmart); the argparse plumbing itself is out of scope."""
from dataclasses import dataclass


@dataclass
class ModelParams:
    sh_degree: int = 3
    white_background: bool = False
    data_device: str = "cuda"
    uniform_upsampling: bool = False


@dataclass
class PipelineParams:
    convert_SHs_python: bool = False
    compute_cov3D_python: bool = False
    depth_ratio: float = 0.0
    debug: bool = False
    # not in the reference: feed the rasterizer the raw parameters and fuse exp / sigmoid / normalize
    # and the dc|rest concatenation into its kernels (same results, ~0.4 ms less per 1M-Gaussian step)
    fused_activations: bool = True
    # not in the reference: the backward returns dL/drgb [N,3] and the Adam kernel of the SH tensors rebuilds
    # basis(dir) x dL/drgb itself (48 -> 3 gradient floats per Gaussian written, read and exchanged between GPUs)
    factored_sh_grad: bool = True
    # not in the reference: while no regularizer is active the fused trainer asks the forward for the colour image alone
    # (GSR_FLAG_COLOR_ONLY: allmap neither accumulated nor written; the reference computes it and multiplies it by zero)
    color_only_when_unregularized: bool = True
    # not in the reference: on a HIP device the trainer applies the objective as one fused autograd node on allmap; True keeps
    # the reference's own formulation instead (render() derives the maps in torch, utils/loss_utils.py's L1 + SSIM, the two
    # regularizers as train.py:132-143) -- what a PYTHONPATH swap under the reference's train.py runs
    reference_objective: bool = False
    # not in the reference: render()'s five derived maps (rend_normal, surf_depth, surf_normal, ...) from one HIP launch each way
    # instead of the reference's torch post-processing (gaussian_renderer/__init__.py:117-156) -- same tensors, same gradients
    fused_surface_maps: bool = False


@dataclass
class OptimizationParams:
    iterations: int = 30_000
    position_lr_init: float = 0.00016
    position_lr_final: float = 0.0000016
    position_lr_delay_mult: float = 0.01
    position_lr_max_steps: int = 30_000
    feature_lr: float = 0.0025
    opacity_lr: float = 0.05
    scaling_lr: float = 0.005
    rotation_lr: float = 0.001
    percent_dense: float = 0.01
    lambda_dssim: float = 0.2
    lambda_dist: float = 0.0
    lambda_normal: float = 0.05
    lambda_segment: float = 0.05
    opacity_cull: float = 0.05
    densification_interval: int = 100
    opacity_reset_interval: int = 3000
    densify_from_iter: int = 500
    densify_until_iter: int = 15_000
    densify_grad_threshold: float = 0.0002
