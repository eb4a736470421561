"""View-parallel data parallelism: one GaussianModel replica per GPU, each rank renders a
different camera, per-Gaussian gradients are averaged across ranks with ONE grouped all-reduce
(RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).

The reference has no distributed code at all; this is the build's
multi-GPU row (section 8(e)): 58 f32 per Gaussian per step in a single grouped collective, plus three small
reductions of the densification statistics right before a densify step so every replica takes
identical clone / split / prune decisions.
"""
import os

import torch
import torch.distributed as dist


class ViewParallel:
    def __init__(self, gaussians, process_group=None, average=True, force=False, pipelined=None, overlap_local=False):
        self.g = gaussians
        self.pg = process_group
        self.average = average
        self.force = force          # run the collectives even at world size 1 (single-GPU rehearsal)
        # pipelined step (RCCL + HIP only): see reduce_and_step(); GSR_DP_PIPELINE=0 switches it off
        self.pipelined = (os.environ.get("GSR_DP_PIPELINE", "1") != "0") if pipelined is None else bool(pipelined)
        # with ONE rank and nothing to exchange, still run the factored SH update on the side stream (beside the next
        # forward's sorting and binning)
        self.overlap_local = bool(overlap_local)
        self._side = None           # side stream of the SH update
        self._pending = None        # event: SH update of the previous step finished
        # opt-in accounting of EXPOSED communication (bench.py --gpus N): event pairs around every point where the
        # step's own stream waits for a collective; exposed_ms() adds them up.  Off by default (two events per wait).
        self.probe = False
        self._probe_events = []
        self._gathered = None       # factored step: all-gathered colour-gradient records (grow-only)
        self._xyz_snap = None       # factored pipelined step: the positions the backward saw

    def _wait_on_main(self, work, dev):
        """The current stream waits for collective `work`; with the probe on, how long it actually stalls there is
        recorded (0 when the collective had already finished: fully hidden)."""
        if not self.probe or not torch.cuda.is_available():
            work.wait()
            return
        s = torch.cuda.current_stream(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        work.wait()
        e1.record(s)
        self._probe_events.append((e0, e1))

    def exposed_ms(self, reset=True):
        """Sum (ms) of the main-stream stalls on collectives since the last call; synchronises the events."""
        total = 0.0
        for e0, e1 in self._probe_events:
            e1.synchronize()
            total += e0.elapsed_time(e1)
        n = len(self._probe_events)
        if reset:
            self._probe_events = []
        return total, n

    def replica_checksum(self):
        """Order-independent 64-bit fingerprints of every parameter tensor (sum of the raw bit patterns), for the
        replica-equality check of a view-parallel run: replicas must stay BIT-identical."""
        with torch.no_grad():
            self.finish()
            return torch.stack([p.detach().contiguous().view(torch.int32).to(torch.int64).sum() for p in self.g.parameters()])

    def replicas_identical(self):
        """True when every rank holds bit-identical parameters (all ranks must call this)."""
        if self.world_size == 1:
            return True
        mine = self.replica_checksum()
        if dist.get_backend(self.pg) != "nccl":
            mine = mine.cpu()
        ref = mine.clone()
        dist.broadcast(ref, src=0, group=self.pg)
        same = torch.tensor([1 if torch.equal(ref, mine) else 0], dtype=torch.int64, device=mine.device)
        dist.all_reduce(same, op=dist.ReduceOp.MIN, group=self.pg)
        return bool(int(same.item()))

    @torch.no_grad()
    def resync_from_rank0(self, optimizer=None):
        """Overwrite every replica's parameters (and the Adam moments / step counts of `optimizer`) with rank 0's."""
        if self.world_size == 1:
            return
        self.finish()
        tensors = [p.data for p in self.g.parameters()]
        if optimizer is not None:
            for p in self.g.parameters():
                st = optimizer.state.get(p, {})
                tensors += [st[k] for k in ("exp_avg", "exp_avg_sq") if k in st]
        host = dist.get_backend(self.pg) != "nccl"
        for t in tensors:
            if host and t.is_cuda:
                c = t.cpu()
                dist.broadcast(c, src=0, group=self.pg)
                t.copy_(c)
            else:
                dist.broadcast(t, src=0, group=self.pg)
        if optimizer is not None:
            for p in self.g.parameters():
                st = optimizer.state.get(p, {})
                if "step" in st:
                    c = st["step"].detach().clone().cpu().reshape(1).double()
                    if not host:
                        c = c.to(p.device)
                    dist.broadcast(c, src=0, group=self.pg)
                    st["step"].fill_(float(c.item()))
        # the parameters were rewritten through `.data` (no version bump): a colour cache built from the old ones is stale
        for opt in {id(o): o for o in (optimizer, getattr(self.g, "optimizer", None)) if o is not None}.values():
            if hasattr(opt, "invalidate_color_cache"):
                opt.invalidate_color_cache()

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

    @staticmethod
    def flat_gradient(grads):
        """The fused rasterizer backward writes the six parameter gradients into ONE buffer
        ([xyz | f_dc | opacity | scaling | rotation | f_rest], rasterizer.py).  Returns a 1-D tensor over that
        buffer when `grads` are exactly such adjacent views (in any order, up to 3 floats of alignment padding
        between them), else None."""
        try:
            st = grads[0].untyped_storage()
            if any(g.untyped_storage().data_ptr() != st.data_ptr() or not g.is_contiguous() or g.dtype != grads[0].dtype
                   for g in grads):
                return None
            spans = sorted((g.storage_offset(), g.numel()) for g in grads)
            pos = spans[0][0]
            for off, n in spans:
                if not 0 <= off - pos < 4:       # segments may be padded to 16 bytes
                    return None
                pos = off + n
            return torch.empty(0, dtype=grads[0].dtype, device=grads[0].device).set_(st, spans[0][0], (pos - spans[0][0],))
        except (RuntimeError, AttributeError):
            return None

    def allreduce_gradients(self):
        """Average (or sum) the six parameter gradients across ranks IN PLACE: one all-reduce over the flat
        58-floats-per-Gaussian buffer the fused backward produced (232 MB at 1 M Gaussians, no flatten / un-flatten
        copy); when the gradients are separate tensors, six all-reduces issued as one RCCL group."""
        if self.world_size == 1 and not self.force:
            return
        grads = [p.grad for p in self.g.parameters()]
        if any(gr is None for gr in grads):
            raise RuntimeError("allreduce_gradients() called before backward()")
        flat = self.flat_gradient(grads)
        if flat is not None:
            use_avg = self.average and dist.get_backend(self.pg) == "nccl"
            dist.all_reduce(flat, op=dist.ReduceOp.AVG if use_avg else dist.ReduceOp.SUM, group=self.pg)
            if self.average and not use_avg:
                flat.mul_(1.0 / self.world_size)
            return
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

    # ------------------------------------------------------------------ factored SH gradient (13 floats per Gaussian)
    # The SH gradient of one view is the outer product basis(view direction) x dL/drgb, so instead of all-reducing
    # 48 floats per Gaussian the ranks ALL-GATHER their [N,3] colour gradients (+ camera position) and every rank
    # rebuilds   mean_r basis(dir_r) x g_r   inside its Adam kernel (fused_adam.FusedAdam.step_sh_factored), in rank
    # order -- identical bits on every replica.  Wire volume per Gaussian and rank: 40 B all-reduced + 12 B gathered
    # per peer, instead of 232 B all-reduced.  xGMI is point-to-point (one ~77 GB/s link per direction per GPU pair), so
    # the volume is what bounds a view-parallel step once the step itself takes 2.3 ms.
    def _gather_records(self, rec, async_op):
        world = self.world_size
        stride = rec.record.numel()
        need = world * stride
        if self._gathered is None or self._gathered.numel() < need or self._gathered.device != rec.record.device:
            self._gathered = torch.empty(need + need // 8, dtype=torch.float32, device=rec.record.device)
        out = self._gathered[:need]
        if dist.get_backend(self.pg) == "nccl":
            work = dist.all_gather_into_tensor(out, rec.record, group=self.pg, async_op=async_op)
        else:
            work = dist.all_gather([out[i * stride:(i + 1) * stride] for i in range(world)], rec.record, group=self.pg,
                                   async_op=async_op)
        return out, work

    @torch.no_grad()
    def exchange_factored(self, rec):
        """Blocking exchange of a factored backward (rasterizer.ColorGradRecord): all-reduce of the geometry gradients
        in place, all-gather of the colour-gradient records; the result is parked on `rec` for
        trainer.optimizer_step()."""
        rec.exchanged, rec.gathered, rec.n_views, rec.grad_scale = True, None, 1, 1.0
        if self.world_size == 1 and not self.force:
            return
        self.finish()
        use_avg = self.average and dist.get_backend(self.pg) == "nccl"
        probe = self.probe and rec.head.is_cuda
        if probe:       # blocking exchange: everything between these two events is exposed
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(rec.head.device))
        dist.all_reduce(rec.head, op=dist.ReduceOp.AVG if use_avg else dist.ReduceOp.SUM, group=self.pg)
        if self.average and not use_avg:
            rec.head.mul_(1.0 / self.world_size)
        rec.gathered, _ = self._gather_records(rec, False)
        if probe:
            e1.record(torch.cuda.current_stream(rec.head.device))
            self._probe_events.append((e0, e1))
        rec.n_views = self.world_size
        rec.grad_scale = 1.0 / self.world_size if self.average else 1.0

    @torch.no_grad()     # rec.xyz is the parameter itself: the snapshot copy must not enter an autograd graph
    def _factored_step(self, optimizer, rec):
        g = self.g
        f_dc, f_rest = g._features_dc, g._features_rest
        stride, deg = rec.record.numel(), rec.sh_degree
        active = self.world_size > 1 or self.force          # ranks to exchange with
        if not active and not (self.overlap_local and self.pipelined and rec.head.is_cuda):
            optimizer.step_with_sh(f_dc, f_rest, rec, rec.record, 1)
            return
        self.finish()                                     # the previous step's SH update (normally long done)
        if active and not (self.pipelined and rec.head.is_cuda and dist.get_backend(self.pg) == "nccl"):
            self.exchange_factored(rec)
            optimizer.step_with_sh(f_dc, f_rest, rec, rec.gathered, rec.n_views, rec.grad_scale)
            return
        # Pipelined: the main stream waits only for the 40 MB geometry all-reduce, updates xyz / opacity / scaling /
        # rotation and starts the next forward; the all-gather of the colour gradients and the SH update run on the
        # side stream against a snapshot of the positions, and the next forward puts its SH colour pass behind them on
        # that stream (RasterState.set_pending_param_event).  With `overlap_local` the same overlap is used without any
        # exchange: the bandwidth-bound SH update runs beside the latency-bound sorting / binning of the next forward.
        from . import rasterizer
        dev = rec.head.device
        w_head = w_rec = None
        records, n_views, scale = rec.record, 1, 1.0
        if active:
            n_views = self.world_size
            scale = 1.0 / n_views if self.average else 1.0
            op = dist.ReduceOp.AVG if self.average else dist.ReduceOp.SUM
            w_head = dist.all_reduce(rec.head, op=op, group=self.pg, async_op=True)
            records, w_rec = self._gather_records(rec, True)
        if self._xyz_snap is None or self._xyz_snap.shape != rec.xyz.shape or self._xyz_snap.device != dev:
            self._xyz_snap = torch.empty_like(rec.xyz)
        if w_head is not None:
            self._wait_on_main(w_head, dev)               # current stream waits for the geometry collective only
        if self._side is None:
            self._side = torch.cuda.Stream(device=dev)
        next_view = getattr(optimizer, "next_view", None)
        # geometry tensors (the features have no .grad: skipped); the same launch leaves the positions the backward saw in
        # the snapshot (FusedAdam.step(keep_old=...): no copy launch beside it)
        optimizer.step(keep_old=(rec.xyz, self._xyz_snap))
        geo_done = torch.cuda.Event()                     # snapshot written; and -- for the next view's colour, evaluated
        geo_done.record(torch.cuda.current_stream(dev))   # from the NEW positions -- the geometry step done
        with torch.cuda.stream(self._side):
            self._side.wait_event(geo_done)
            if w_rec is not None:
                w_rec.wait()
            optimizer.step_sh_factored(f_dc, f_rest, self._xyz_snap, records, n_views, stride, deg, scale, stream=self._side,
                                       next_view=next_view, xyz_next=rec.xyz if next_view is not None else None)
            ev = torch.cuda.Event()
            ev.record(self._side)
        rec.flat.record_stream(self._side)
        self._pending = ev
        self._raster_state().set_pending_param_event(ev, self._side)

    def reduce_and_step(self, optimizer, rec=None):
        """all-reduce + optimiser step of one iteration.  `rec`: the factored SH gradient of this iteration's backward
        (trainer.optimizer_step passes it), handled by _factored_step().

        Pipelined form (RCCL, HIP tensors, flat gradient buffer, FusedAdam): the buffer is reduced as two
        collectives -- "geometry + dc" (13 floats per Gaussian) and "SH rest" (45) -- and the main stream only
        waits for the first: it updates xyz / f_dc / opacity / scaling / rotation and is free to start the next
        forward, whose geometry, sorting and binning phase (~0.4 ms at 1 M Gaussians) does not read the SH
        coefficients.  The second collective and the Adam update of f_rest run on a side stream; the next forward
        waits for them right before its SH colour pass (RasterState.set_pending_param_event).  Anything else that
        touches the parameters must call finish() first (densification, saving, evaluation renders do).
        Falls back to allreduce_gradients() + optimizer.step() whenever a precondition is missing."""
        if rec is not None:
            if rec.exchanged:
                raise RuntimeError("this factored gradient was already exchanged (training_step(step_optimizer=False)); "
                                   "finish the iteration with trainer.optimizer_step(gaussians)")
            return self._factored_step(optimizer, rec)
        params = list(self.g.parameters())
        grads = [p.grad for p in params]
        active = self.world_size > 1 or self.force
        from .fused_adam import FusedAdam
        ok = (active and self.pipelined and all(gr is not None and gr.is_cuda for gr in grads)
              and dist.get_backend(self.pg) == "nccl" and isinstance(optimizer, FusedAdam))
        flat = self.flat_gradient(grads) if ok else None
        rest_p = getattr(self.g, "_features_rest", None)
        if flat is None or rest_p is None or rest_p.grad is None or rest_p.grad.numel() == 0:
            self.finish()
            self.allreduce_gradients()
            optimizer.step()
            return
        rest_g = rest_p.grad
        # f_rest must be the tail of the buffer
        if rest_g.storage_offset() + rest_g.numel() != flat.storage_offset() + flat.numel():
            self.finish()
            self.allreduce_gradients()
            optimizer.step()
            return
        from . import rasterizer
        dev = flat.device
        self.finish()                                     # the previous step's SH update (normally long done)
        n_head = flat.numel() - rest_g.numel()
        head, tail = flat[:n_head], flat[n_head:]
        op = dist.ReduceOp.AVG if self.average else dist.ReduceOp.SUM
        # the SH tail travels as two collectives, so that the Adam update of its first half overlaps the transfer of
        # the second
        n_tail = tail.numel()
        cut = (n_tail // 2) & ~3                         # keep both halves 16-byte aligned
        w_head = dist.all_reduce(head, op=op, group=self.pg, async_op=True)
        w_t1 = dist.all_reduce(tail[:cut], op=op, group=self.pg, async_op=True) if cut > 0 else None
        w_t2 = dist.all_reduce(tail[cut:], op=op, group=self.pg, async_op=True)
        self._wait_on_main(w_head, dev)                   # current stream waits for the first collective only
        optimizer.step(only=[p for p in params if p is not rest_p])
        if self._side is None:
            self._side = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(self._side):
            if w_t1 is not None:
                w_t1.wait()
                optimizer.step_slice(rest_p, 0, cut, stream=self._side, count_step=True)
            w_t2.wait()
            optimizer.step_slice(rest_p, cut, n_tail, stream=self._side, count_step=w_t1 is None)
            ev = torch.cuda.Event()
            ev.record(self._side)
        flat.record_stream(self._side)                    # its memory may be reused only after the side stream is done
        self._pending = ev
        self._raster_state().set_pending_param_event(ev, self._side)

    def _raster_state(self):
        """The model's hand-over slots with the operator (rasterizer.RasterState); created on models that lack one."""
        st = getattr(self.g, "raster_state", None)
        if st is None:
            from .rasterizer import RasterState
            st = self.g.raster_state = RasterState()
        return st

    def finish(self):
        """Make the current stream wait for the outstanding SH update of a pipelined step, if any."""
        if self._pending is not None:
            dev = self.g.parameters()[0].device
            self._raster_state().pop_pending()                     # un-park it (no-op if a forward consumed it)
            torch.cuda.current_stream(dev).wait_event(self._pending)
            self._pending = None

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
