"""GaussianModel: the SoA parameter store the rasterizer is fed from.

Counterpart of scene/gaussian_model.py:26-553 of the reference with the same attribute /
method names, activations (exp / sigmoid / normalize, :37-43), Adam groups and learning rates
(:282-295), densify / prune / reset logic (:398-553), checkpoint tuple (:66-101) and PLY column
layout incl. the extra `segment` column (:305-396).  Differences: the device is a constructor
argument (the reference hard-codes "cuda"), PLY I/O is self-contained (no plyfile), and the
SAM-mask segment augmentation of create_from_pcd (:132-258) is out of scope.
"""
import os

import numpy as np
import torch
from torch import nn

from .general import inverse_sigmoid, get_expon_lr_func, build_rotation, build_scaling_rotation
from .sh import RGB2SH


class GaussianModel:
    def setup_functions(self):
        def build_covariance_from_scaling_rotation(center, scaling, scaling_modifier, rotation):
            # 4x4 splat->world, rows (tu,0),(tv,0),(n,0),(p,1)   [scene/gaussian_model.py:29-35]
            s3 = torch.cat([scaling * scaling_modifier, torch.ones_like(scaling)], dim=-1)[:, :3]
            RS = build_scaling_rotation(s3, rotation).permute(0, 2, 1)
            trans = torch.zeros((center.shape[0], 4, 4), dtype=torch.float, device=center.device)
            trans[:, :3, :3] = RS
            trans[:, 3, :3] = center
            trans[:, 3, 3] = 1
            return trans

        self.scaling_activation = torch.exp
        self.scaling_inverse_activation = torch.log
        self.covariance_activation = build_covariance_from_scaling_rotation
        self.opacity_activation = torch.sigmoid
        self.inverse_opacity_activation = inverse_sigmoid
        self.rotation_activation = torch.nn.functional.normalize

    def __init__(self, sh_degree: int, uniform_upsampling: bool = False, device="cuda"):
        self.device = torch.device(device)
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        e = torch.empty(0, device=self.device)
        self._xyz = self._features_dc = self._features_rest = e
        self._scaling = self._rotation = self._opacity = e
        self.max_radii2D = self.xyz_gradient_accum = self.denom = e
        self._segments = torch.empty(0, dtype=torch.long, device=self.device)
        self.optimizer = None
        self.percent_dense = 0
        self.spatial_lr_scale = 0
        self.uniform_upsampling = uniform_upsampling
        self.use_fused_adam = True
        # hand-over slots between the rasterizer operator and the optimiser step of THIS model (pending side-stream
        # update, factored SH gradient): per model, nothing at module level
        from .rasterizer import RasterState
        self.raster_state = RasterState()
        self.setup_functions()

    # ------------------------------------------------------------------ checkpoint tuple
    def capture(self):
        return (self.active_sh_degree, self._xyz, self._features_dc, self._features_rest, self._scaling,
                self._rotation, self._opacity, self.max_radii2D, self.xyz_gradient_accum, self.denom,
                self._segments, self.optimizer.state_dict(), float(self.spatial_lr_scale))

    def restore(self, model_args, training_args):
        (self.active_sh_degree, self._xyz, self._features_dc, self._features_rest, self._scaling,
         self._rotation, self._opacity, self.max_radii2D, xyz_gradient_accum, denom, segments,
         opt_dict, self.spatial_lr_scale) = model_args
        self.training_setup(training_args)
        self.xyz_gradient_accum = xyz_gradient_accum
        self.denom = denom
        self._segments = segments
        self.optimizer.load_state_dict(opt_dict)

    # ------------------------------------------------------------------ getters
    @property
    def get_scaling(self):
        return self.scaling_activation(self._scaling)

    @property
    def get_rotation(self):
        return self.rotation_activation(self._rotation)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_opacity(self):
        return self.opacity_activation(self._opacity)

    def get_covariance(self, scaling_modifier=1):
        return self.covariance_activation(self.get_xyz, self.get_scaling, scaling_modifier, self._rotation)

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    # ------------------------------------------------------------------ construction
    def _install(self, xyz, f_dc, f_rest, scaling, rotation, opacity, segments=None):
        dev = self.device
        mk = lambda t: nn.Parameter(t.to(dev).float().contiguous().requires_grad_(True))
        self._xyz, self._features_dc, self._features_rest = mk(xyz), mk(f_dc), mk(f_rest)
        self._scaling, self._rotation, self._opacity = mk(scaling), mk(rotation), mk(opacity)
        n = self._xyz.shape[0]
        self._segments = segments.to(dev).long() if segments is not None else torch.zeros(n, dtype=torch.long, device=dev)
        self.max_radii2D = torch.zeros(n, device=dev)

    def create_from_params(self, params: dict, spatial_lr_scale: float = 1.0, active_sh_degree=None):
        """From raw tensors in storage layout (gaussmart_amd.synthetic.make_scene)."""
        self.spatial_lr_scale = spatial_lr_scale
        self._install(params["xyz"], params["features_dc"], params["features_rest"], params["scaling"],
                      params["rotation"], params["opacity"])
        self.active_sh_degree = self.max_sh_degree if active_sh_degree is None else active_sh_degree

    def create_from_pcd(self, pcd, spatial_lr_scale: float, dist2_fn=None):
        """pcd has .points [N,3], .colors [N,3] in [0,1] (and optionally .segments).
        scene/gaussian_model.py:166-275 without the mask-area augmentation."""
        self.spatial_lr_scale = spatial_lr_scale
        dev = self.device
        pts = torch.tensor(np.asarray(pcd.points)).float().to(dev)
        col = RGB2SH(torch.tensor(np.asarray(pcd.colors)).float().to(dev))
        feats = torch.zeros((pts.shape[0], 3, (self.max_sh_degree + 1) ** 2), device=dev)
        feats[:, :3, 0] = col
        seg = getattr(pcd, "segments", None)
        seg = torch.tensor(np.asarray(seg)).long() if seg is not None else None
        if dist2_fn is None:
            from .knn import distCUDA2 as dist2_fn
        dist2 = torch.clamp_min(dist2_fn(pts), 0.0000001)
        scales = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 2)
        rots = torch.rand((pts.shape[0], 4), device=dev)
        opac = inverse_sigmoid(0.1 * torch.ones((pts.shape[0], 1), dtype=torch.float, device=dev))
        self._install(pts, feats[:, :, 0:1].transpose(1, 2), feats[:, :, 1:].transpose(1, 2), scales, rots, opac, seg)

    # ------------------------------------------------------------------ optimiser
    def training_setup(self, training_args):
        self.percent_dense = training_args.percent_dense
        n, dev = self.get_xyz.shape[0], self.device
        self.xyz_gradient_accum = torch.zeros((n, 1), device=dev)
        self.denom = torch.zeros((n, 1), device=dev)
        groups = [
            {"params": [self._xyz], "lr": training_args.position_lr_init * self.spatial_lr_scale, "name": "xyz"},
            {"params": [self._features_dc], "lr": training_args.feature_lr, "name": "f_dc"},
            {"params": [self._features_rest], "lr": training_args.feature_lr / 20.0, "name": "f_rest"},
            {"params": [self._opacity], "lr": training_args.opacity_lr, "name": "opacity"},
            {"params": [self._scaling], "lr": training_args.scaling_lr, "name": "scaling"},
            {"params": [self._rotation], "lr": training_args.rotation_lr, "name": "rotation"},
        ]
        if self.use_fused_adam and dev.type == "cuda":
            from .fused_adam import FusedAdam          # one HIP launch for all six groups
            self.optimizer = FusedAdam(groups, lr=0.0, eps=1e-15)
        else:
            self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
        self.xyz_scheduler_args = get_expon_lr_func(
            lr_init=training_args.position_lr_init * self.spatial_lr_scale,
            lr_final=training_args.position_lr_final * self.spatial_lr_scale,
            lr_delay_mult=training_args.position_lr_delay_mult, max_steps=training_args.position_lr_max_steps)

    def update_learning_rate(self, iteration):
        for group in self.optimizer.param_groups:
            if group["name"] == "xyz":
                lr = float(self.xyz_scheduler_args(iteration))   # a plain float: the schedule computes in NumPy, and a NumPy
                group["lr"] = lr                                  # scalar in the optimiser state would make the checkpoint
                return lr                                         # (capture()) unloadable with torch.load(weights_only=True)

    def parameters(self):
        return [self._xyz, self._features_dc, self._features_rest, self._opacity, self._scaling, self._rotation]

    # ------------------------------------------------------------------ PLY (binary little endian)
    def construct_list_of_attributes(self):
        names = ["x", "y", "z", "nx", "ny", "nz"]
        names += [f"f_dc_{i}" for i in range(self._features_dc.shape[1] * self._features_dc.shape[2])]
        names += [f"f_rest_{i}" for i in range(self._features_rest.shape[1] * self._features_rest.shape[2])]
        names.append("opacity")
        names += [f"scale_{i}" for i in range(self._scaling.shape[1])]
        names += [f"rot_{i}" for i in range(self._rotation.shape[1])]
        return names

    def save_ply(self, path):
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        xyz = self._xyz.detach().cpu().numpy()
        cols = [xyz, np.zeros_like(xyz),
                self._features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy(),
                self._features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy(),
                self._opacity.detach().cpu().numpy(), self._scaling.detach().cpu().numpy(),
                self._rotation.detach().cpu().numpy(),
                self._segments.detach().cpu().numpy()[:, None].astype(np.float32)]
        names = self.construct_list_of_attributes() + ["segment"]
        data = np.concatenate(cols, axis=1).astype("<f4")
        header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % data.shape[0]
        header += "".join(f"property float {n}\n" for n in names) + "end_header\n"
        with open(path, "wb") as f:
            f.write(header.encode("ascii"))
            f.write(data.tobytes())

    @staticmethod
    def _read_ply(path):
        with open(path, "rb") as f:
            names, count, fmt = [], 0, None
            while True:
                line = f.readline().decode("ascii").strip()
                if line.startswith("format"):
                    fmt = line.split()[1]
                elif line.startswith("element vertex"):
                    count = int(line.split()[-1])
                elif line.startswith("property"):
                    _, typ, name = line.split()
                    if typ not in ("float", "float32"):
                        raise ValueError(f"unsupported PLY property type {typ}")
                    names.append(name)
                elif line == "end_header":
                    break
            if fmt != "binary_little_endian":
                raise ValueError("only binary_little_endian PLY files are supported")
            data = np.frombuffer(f.read(count * len(names) * 4), dtype="<f4").reshape(count, len(names))
        return names, data

    def load_ply(self, path):
        names, data = self._read_ply(path)
        col = {n: data[:, i] for i, n in enumerate(names)}
        xyz = np.stack([col["x"], col["y"], col["z"]], axis=1)
        opac = col["opacity"][:, None]
        f_dc = np.stack([col["f_dc_0"], col["f_dc_1"], col["f_dc_2"]], axis=1)[:, :, None]
        seg = col["segment"] if "segment" in col else np.zeros(xyz.shape[0], dtype=np.int32)
        by_idx = lambda pre: sorted((n for n in names if n.startswith(pre)), key=lambda x: int(x.split("_")[-1]))
        extra = by_idx("f_rest_")
        assert len(extra) == 3 * (self.max_sh_degree + 1) ** 2 - 3
        f_extra = np.stack([col[n] for n in extra], axis=1).reshape(xyz.shape[0], 3, (self.max_sh_degree + 1) ** 2 - 1)
        scales = np.stack([col[n] for n in by_idx("scale_")], axis=1)
        rots = np.stack([col[n] for n in by_idx("rot")], axis=1)
        t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float)
        self._install(t(xyz), t(f_dc).transpose(1, 2), t(f_extra).transpose(1, 2), t(scales), t(rots), t(opac),
                      torch.tensor(np.asarray(seg)).long())
        self.active_sh_degree = self.max_sh_degree

    # ------------------------------------------------------------------ optimiser surgery
    def reset_opacity(self):
        new = self.inverse_opacity_activation(torch.min(self.get_opacity, torch.ones_like(self.get_opacity) * 0.01))
        self._opacity = self.replace_tensor_to_optimizer(new, "opacity")["opacity"]

    def replace_tensor_to_optimizer(self, tensor, name):
        out = {}
        for group in self.optimizer.param_groups:
            if group["name"] == name:
                stored = self.optimizer.state.get(group["params"][0], None)
                if stored is not None:
                    stored["exp_avg"] = torch.zeros_like(tensor)
                    stored["exp_avg_sq"] = torch.zeros_like(tensor)
                    del self.optimizer.state[group["params"][0]]
                group["params"][0] = nn.Parameter(tensor.requires_grad_(True))
                if stored is not None:
                    self.optimizer.state[group["params"][0]] = stored
                out[group["name"]] = group["params"][0]
        return out

    def _prune_optimizer(self, mask, extra=()):
        """Keeps the rows of `mask` in every parameter, its Adam moments and the `extra` per-Gaussian tensors with ONE
        fused compaction (compaction.py) instead of ~21 boolean-index operations; returns (params by name, extras)."""
        from .compaction import compact_rows
        groups = self.optimizer.param_groups
        olds = [g["params"][0] for g in groups]
        states = [self.optimizer.state.get(p, None) for p in olds]
        batch = [p.detach() for p in olds]
        for st in states:
            if st is not None:
                batch += [st["exp_avg"], st["exp_avg_sq"]]
        batch += list(extra)
        new = compact_rows(batch, mask)
        out, k = {}, len(olds)
        for i, (group, old, st) in enumerate(zip(groups, olds, states)):
            new_p = nn.Parameter(new[i].requires_grad_(True))
            if st is not None:
                st["exp_avg"], st["exp_avg_sq"] = new[k], new[k + 1]
                k += 2
                del self.optimizer.state[old]
                self.optimizer.state[new_p] = st
            group["params"][0] = new_p
            out[group["name"]] = new_p
        return out, new[k:]

    def _adopt(self, t):
        self._xyz, self._features_dc, self._features_rest = t["xyz"], t["f_dc"], t["f_rest"]
        self._opacity, self._scaling, self._rotation = t["opacity"], t["scaling"], t["rotation"]

    def prune_points(self, mask):
        keep = ~mask
        params, extras = self._prune_optimizer(keep, (self.xyz_gradient_accum, self.denom, self.max_radii2D, self._segments))
        self._adopt(params)
        self.xyz_gradient_accum, self.denom, self.max_radii2D, self._segments = extras

    def cat_tensors_to_optimizer(self, tensors_dict):
        out = {}
        for group in self.optimizer.param_groups:
            assert len(group["params"]) == 1
            ext = tensors_dict[group["name"]]
            stored = self.optimizer.state.get(group["params"][0], None)
            new_p = nn.Parameter(torch.cat((group["params"][0], ext), dim=0).requires_grad_(True))
            if stored is not None:
                stored["exp_avg"] = torch.cat((stored["exp_avg"], torch.zeros_like(ext)), dim=0)
                stored["exp_avg_sq"] = torch.cat((stored["exp_avg_sq"], torch.zeros_like(ext)), dim=0)
                del self.optimizer.state[group["params"][0]]
                self.optimizer.state[new_p] = stored
            group["params"][0] = new_p
            out[group["name"]] = new_p
        return out

    def densification_postfix(self, new_xyz, new_features_dc, new_features_rest, new_opacities, new_scaling,
                              new_rotation, new_segments):
        d = {"xyz": new_xyz, "f_dc": new_features_dc, "f_rest": new_features_rest, "opacity": new_opacities,
             "scaling": new_scaling, "rotation": new_rotation}
        self._adopt(self.cat_tensors_to_optimizer(d))
        self._segments = torch.cat([self._segments, new_segments], dim=0)
        n, dev = self.get_xyz.shape[0], self.device
        self.xyz_gradient_accum = torch.zeros((n, 1), device=dev)
        self.denom = torch.zeros((n, 1), device=dev)
        self.max_radii2D = torch.zeros(n, device=dev)

    def densify_and_split(self, grads, grad_threshold, scene_extent, N=2, generator=None):
        n_init = self.get_xyz.shape[0]
        padded = torch.zeros(n_init, device=self.device)
        padded[:grads.shape[0]] = grads.squeeze()
        sel = padded >= grad_threshold
        sel = torch.logical_and(sel, torch.max(self.get_scaling, dim=1).values > self.percent_dense * scene_extent)
        stds = self.get_scaling[sel].repeat(N, 1)
        stds = torch.cat([stds, 0 * torch.ones_like(stds[:, :1])], dim=-1)
        samples = torch.normal(mean=torch.zeros_like(stds), std=stds, generator=generator)
        rots = build_rotation(self._rotation[sel]).repeat(N, 1, 1)
        new_xyz = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + self.get_xyz[sel].repeat(N, 1)
        new_scaling = self.scaling_inverse_activation(self.get_scaling[sel].repeat(N, 1) / (0.8 * N))
        self.densification_postfix(new_xyz, self._features_dc[sel].repeat(N, 1, 1), self._features_rest[sel].repeat(N, 1, 1),
                                   self._opacity[sel].repeat(N, 1), new_scaling, self._rotation[sel].repeat(N, 1),
                                   self._segments[sel].repeat(N))
        prune = torch.cat((sel, torch.zeros(N * int(sel.sum()), device=self.device, dtype=torch.bool)))
        self.prune_points(prune)

    def densify_and_clone(self, grads, grad_threshold, scene_extent):
        sel = torch.norm(grads, dim=-1) >= grad_threshold
        sel = torch.logical_and(sel, torch.max(self.get_scaling, dim=1).values <= self.percent_dense * scene_extent)
        self.densification_postfix(self._xyz[sel], self._features_dc[sel], self._features_rest[sel], self._opacity[sel],
                                   self._scaling[sel], self._rotation[sel], self._segments[sel])

    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size, generator=None):
        grads = self.xyz_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        self.densify_and_clone(grads, max_grad, extent)
        self.densify_and_split(grads, max_grad, extent, generator=generator)
        prune_mask = (self.get_opacity < min_opacity).squeeze()
        if max_screen_size:
            big_vs = self.max_radii2D > max_screen_size
            big_ws = self.get_scaling.max(dim=1).values > 0.1 * extent
            prune_mask = torch.logical_or(torch.logical_or(prune_mask, big_vs), big_ws)
        self.prune_points(prune_mask)

    def add_densification_stats(self, viewspace_point_tensor, update_filter):
        # norm over all three components: the rasterizer guarantees .z == 0  [:551-553]
        self.xyz_gradient_accum[update_filter] += torch.norm(viewspace_point_tensor.grad[update_filter], dim=-1, keepdim=True)
        self.denom[update_filter] += 1

    def update_densification_stats(self, viewspace_point_tensor, radii):
        """max_radii2D + add_densification_stats of one iteration (train.py:199-203).  On a HIP device: ONE kernel and
        no host synchronisation (boolean-mask indexing synchronises on nonzero()); elsewhere the reference's ops."""
        grad = viewspace_point_tensor.grad
        fused = (radii.is_cuda and grad is not None and grad.is_contiguous() and radii.dtype == torch.int32
                 and all(t.is_contiguous() and t.dtype == torch.float32 for t in (self.max_radii2D, self.xyz_gradient_accum, self.denom)))
        if not fused:
            vis = radii > 0
            self.max_radii2D[vis] = torch.max(self.max_radii2D[vis], radii[vis].to(self.max_radii2D.dtype))
            self.add_densification_stats(viewspace_point_tensor, vis)
            return
        import ctypes as C
        from . import _lib
        with torch.cuda.device(radii.device):
            _lib.check(_lib.lib().gsr_densify_stats(
                radii.shape[0], C.c_void_p(radii.data_ptr()), C.c_void_p(grad.data_ptr()), C.c_void_p(self.max_radii2D.data_ptr()),
                C.c_void_p(self.xyz_gradient_accum.data_ptr()), C.c_void_p(self.denom.data_ptr()),
                C.c_void_p(torch.cuda.current_stream(radii.device).cuda_stream)))
