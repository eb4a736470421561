"""View-parallel data parallelism: one GaussianModel replica per GPU, each rank renders a
different camera, per-Gaussian gradients are summed across ranks with ONE flat all-reduce
(RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).

The reference has no distributed code at all; this is the build's
multi-GPU row (section 8(e)): 58 f32 per Gaussian per step in a single bucket, plus three small
reductions of the densification statistics right before a densify step so every replica takes
identical clone / split / prune decisions.
"""
import torch
import torch.distributed as dist


class ViewParallel:
    def __init__(self, gaussians, process_group=None, average=True):
        self.g = gaussians
        self.pg = process_group
        self.average = average
        self._bucket = None

    @property
    def world_size(self):
        return dist.get_world_size(self.pg) if dist.is_available() and dist.is_initialized() else 1

    @property
    def rank(self):
        return dist.get_rank(self.pg) if dist.is_available() and dist.is_initialized() else 0

    def shard_views(self, views, epoch_seed: int):
        """Same shuffled order on every rank; rank r takes perm[r::world]."""
        gen = torch.Generator().manual_seed(epoch_seed)
        perm = torch.randperm(len(views), generator=gen).tolist()
        return [views[i] for i in perm[self.rank::self.world_size]]

    def allreduce_gradients(self):
        """Sum (or average) the six parameter gradients across ranks through one flat bucket."""
        if self.world_size == 1:
            return
        params = self.g.parameters()
        grads = [p.grad for p in params]
        if any(gr is None for gr in grads):
            raise RuntimeError("allreduce_gradients() called before backward()")
        total = sum(gr.numel() for gr in grads)
        if self._bucket is None or self._bucket.numel() != total or self._bucket.device != grads[0].device:
            self._bucket = torch.empty(total, dtype=grads[0].dtype, device=grads[0].device)
        views, off = [], 0
        for gr in grads:
            v = self._bucket[off:off + gr.numel()].view_as(gr)
            v.copy_(gr)
            views.append(v)
            off += gr.numel()
        dist.all_reduce(self._bucket, op=dist.ReduceOp.SUM, group=self.pg)
        if self.average:
            self._bucket.mul_(1.0 / self.world_size)
        for p, v in zip(params, views):
            p.grad.copy_(v)

    def sync_densification_stats(self):
        """xyz_gradient_accum / denom are summed, max_radii2D is max-reduced."""
        if self.world_size == 1:
            return
        dist.all_reduce(self.g.xyz_gradient_accum, op=dist.ReduceOp.SUM, group=self.pg)
        dist.all_reduce(self.g.denom, op=dist.ReduceOp.SUM, group=self.pg)
        dist.all_reduce(self.g.max_radii2D, op=dist.ReduceOp.MAX, group=self.pg)

    def replicated_generator(self, iteration: int, device):
        """Identical RNG stream on every rank for the split samples (scene/gaussian_model.py:504)."""
        gen = torch.Generator(device=device)
        gen.manual_seed(0x5EED + iteration)
        return gen
