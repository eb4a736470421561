"""View-parallel data parallelism: one GaussianModel replica per GPU, each rank renders a
different camera, per-Gaussian gradients are averaged across ranks with ONE grouped all-reduce
(RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).

The reference has no distributed code at all; this is the build's
multi-GPU row (section 8(e)): 58 f32 per Gaussian per step in a single grouped collective, plus three small
reductions of the densification statistics right before a densify step so every replica takes
identical clone / split / prune decisions.
"""
import torch
import torch.distributed as dist


class ViewParallel:
    def __init__(self, gaussians, process_group=None, average=True, force=False):
        self.g = gaussians
        self.pg = process_group
        self.average = average
        self.force = force          # run the collectives even at world size 1 (single-GPU rehearsal)

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
        """Average (or sum) the six parameter gradients across ranks IN PLACE.  On RCCL the six
        all-reduces are issued as one group (ncclGroupStart/End through torch's coalescing
        manager), i.e. one fused collective over 58 floats per Gaussian without a flatten /
        un-flatten copy of the 232 MB (at 1 M Gaussians) bucket; on gloo (CPU tests) they run one
        after the other."""
        if self.world_size == 1 and not self.force:
            return
        grads = [p.grad for p in self.g.parameters()]
        if any(gr is None for gr in grads):
            raise RuntimeError("allreduce_gradients() called before backward()")
        grads = [gr if gr.is_contiguous() else gr.contiguous() for gr in grads]
        backend = dist.get_backend(self.pg)
        use_avg = self.average and backend == "nccl"
        op = dist.ReduceOp.AVG if use_avg else dist.ReduceOp.SUM
        done = False
        if backend == "nccl" and hasattr(dist, "_coalescing_manager"):
            try:
                with dist._coalescing_manager(group=self.pg, device=grads[0].device, async_ops=False):
                    for gr in grads:
                        dist.all_reduce(gr, op=op, group=self.pg)
                done = True
            except (RuntimeError, TypeError, AttributeError):
                done = False
        if not done:
            for gr in grads:
                dist.all_reduce(gr, op=op, group=self.pg)
        if self.average and not use_avg:
            torch._foreach_mul_(grads, 1.0 / self.world_size)
        for p, gr in zip(self.g.parameters(), grads):
            if p.grad is not gr:
                p.grad = gr

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
