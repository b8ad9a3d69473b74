"""View-parallel training support: every rank holds a full replica of the Gaussian parameters,
renders its own views, and the per-Gaussian gradients are summed across ranks with ONE flat
all-reduce (RCCL over xGMI on MI355X; `gloo` in the CPU tests).

This capability is new relative to the reference, which is single-GPU.  The
message is one fp32 bucket of 59 floats per Gaussian at SH degree 3 (3 xyz + 3 dc + 45 rest +
1 opacity + 3 scale + 4 rotation = 236 B): parameter .grad tensors are VIEWS into that bucket, so
autograd accumulates straight into it and no flatten/unflatten copy is needed around the collective.
"""
import os

import torch
import torch.distributed as dist


class FlatGradBucket:
    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        # every slice starts on a 16-byte boundary: in direct mode the HIP backward stores float4 rows into the slices
        # (dL/drotation, the features_rest rows), whatever P is
        pad4 = lambda n: (n + 3) & ~3
        n = sum(pad4(p.numel()) for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        self.views = []
        for p in self.params:
            v = self.flat[off:off + p.numel()].view_as(p)
            p.grad = v
            self.views.append(v)
            off += pad4(p.numel())

    def sinks(self):
        """{parameter: its slice of the bucket} for diff_gaussian_rasterization.set_grad_sinks (direct mode)."""
        return dict(zip(self.params, self.views))

    def detach_grads(self):
        """Direct mode: param.grad = None, so that autograd ADOPTS the alias of the bucket slice the rasterizer's
        backward wrote the gradient into (no zero-fill of the bucket, no accumulation pass)."""
        for p in self.params:
            p.grad = None

    def zero(self):
        self.flat.zero_()
        for p, v in zip(self.params, self.views):   # re-attach in case an optimiser set grads to None
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v

    def all_reduce(self, group=None, average_over=None, async_op=False):
        """Sum over ranks (then divide by `average_over` views if given)."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            if average_over:
                self.flat.div_(average_over)
            return None
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op:
            return work
        if average_over:
            self.flat.div_(average_over)
        return None


class PipelinedGradExchange:
    """Gradient exchange overlapped with rendering ("on a dedicated stream, overlappable"): two flat
    buckets used alternately — while the all-reduce of view k runs on the communicator's stream, view k+1 is rendered
    into the other bucket.  This is the gradient-accumulation pipeline of a trainer whose optimizer step covers >= 2
    views per GPU (SURVEY's partitioning: GPU g renders views {g, g+N, ...} of the iteration's batch); with one view
    per optimizer step use FlatGradBucket.all_reduce directly.

        ex = PipelinedGradExchange(params, world)
        for each view:  ex.begin_view(); loss = criterion(render(...)); loss.backward(); ex.end_view()
        (begin_view() BEFORE the forward: the rasterizer snapshots the registered sinks when it is called)
        ex.drain()                      # every exchange finished (stream-level), buckets hold the averaged grads

    xGMI is per-link bound, so the exchange of a 236 MB bucket costs about as much as a whole view at 8 GPUs; hiding
    it behind the next view is what keeps view-parallel scaling near-linear."""

    def __init__(self, params, world=None, group=None, direct=False):
        """direct=True: the rasterizer's backward writes each view's gradients straight into the current bucket
        (diff_gaussian_rasterization.set_grad_sinks; needs the raw / chained entry, i.e. the reference's getters or
        render_fused) — ONE view per bucket use, no zero-fill and no accumulation pass over the 59*P floats."""
        self.direct = direct
        self.group = group
        # (MSGS_EXCHANGE_FORCE=1: issue the collectives even with a single rank — lets one GPU exercise the RCCL path)
        self.active = dist.is_available() and dist.is_initialized() and (
            dist.get_world_size(group) > 1 or os.environ.get("MSGS_EXCHANGE_FORCE") == "1")
        self.world = world if world is not None else (dist.get_world_size(group) if self.active else 1)
        self.buckets = [FlatGradBucket(params), FlatGradBucket(params)]
        self.pending = [None, None]
        self.k = 0
        # ncclAvg folds the 1/world into the collective; gloo (CPU tests) has no AVG -> SUM and divide after the wait
        self.avg_op = None
        if self.active and dist.get_backend(group) == "nccl":
            try:
                probe = torch.ones(1, device=self.buckets[0].flat.device)
                dist.all_reduce(probe, op=dist.ReduceOp.AVG, group=group)
                self.avg_op = dist.ReduceOp.AVG
            except Exception:
                self.avg_op = None

    @property
    def current(self):
        return self.buckets[self.k % 2]

    def _finish(self, i):
        w = self.pending[i]
        if w is None:
            return
        w.wait()                                   # NCCL: the current stream waits; the host does not
        if self.avg_op is None and self.world > 1:
            self.buckets[i].flat.div_(self.world)
        self.pending[i] = None

    def begin_view(self):
        i = self.k % 2
        self._finish(i)                            # the exchange issued two views ago used this bucket
        if self.direct:
            import diff_gaussian_rasterization as dgr
            self.buckets[i].detach_grads()
            dgr.set_grad_sinks(self.buckets[i].sinks())
        else:
            self.buckets[i].zero()                 # also points every p.grad at this bucket

    def end_view(self):
        i = self.k % 2
        b = self.buckets[i]
        if self.direct:
            import diff_gaussian_rasterization as dgr
            dgr.set_grad_sinks(None)
            for p, v in zip(b.params, b.views):    # the backward must have delivered every gradient into the bucket
                if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                    raise RuntimeError("PipelinedGradExchange(direct=True): a gradient did not land in the bucket "
                                       "(the rasterizer was not called through its raw / chained entry)")
        if self.active:
            op = self.avg_op if self.avg_op is not None else dist.ReduceOp.SUM
            self.pending[i] = dist.all_reduce(b.flat, op=op, group=self.group, async_op=True)
        elif self.world > 1:
            b.flat.div_(self.world)
        self.k += 1
        return b

    def drain(self):
        for i in (0, 1):
            self._finish(i)


class MultiViewStepExchange:
    """Gradient exchange for an optimizer step that covers k >= 2 views PER RANK (BASELINE config C4 on fewer than 8 GPUs:
    8 / N views per GPU and step).  The rank renders its views through multi_view.ViewPipeline — two in flight, their
    gradients summed inside the per-Gaussian backward kernel straight into the slices of ONE flat bucket
    (diff_gaussian_rasterization.GradAccumulator over FlatGradBucket's views: no per-view zero-fill, no `grad += g` passes) —
    and the bucket crosses the ranks ONCE per step: the dense 59-floats-per-Gaussian all-reduce is paid per step, not per
    view, i.e. 1 / k of FlatGradBucket.all_reduce after every view.

        ex = MultiViewStepExchange(model, total_views)
        ex.step(pipeline, cams_of_this_rank, pipe, bg, backward_fn, **render_settings)   # then optimizer.step()

    `model` carries the reference's leaf names (_xyz, _features_dc, _features_rest, _opacity, _scaling, _rotation)."""

    LEAVES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")

    def __init__(self, model, total_views, group=None, make_accumulator=None):
        """make_accumulator(leaves, dest): injection point for the CPU (gloo) tests of the exchange logic; the default is
        diff_gaussian_rasterization.GradAccumulator (HIP, no CPU fallback)"""
        self.model = model
        self.group = group
        self.total_views = int(total_views)
        leaves = [getattr(model, n) for n in self.LEAVES]
        self.bucket = FlatGradBucket(leaves)
        self.bucket.detach_grads()
        if make_accumulator is None:
            import diff_gaussian_rasterization as dgr
            make_accumulator = dgr.GradAccumulator
        self.acc = make_accumulator(leaves, dest=self.bucket.views)

    def step(self, pipeline, cams, pipe, bg_color, backward_fn, **kw):
        """all views of this rank (forward + backward, gradients into the bucket), then one all-reduce; afterwards every
        leaf's .grad is the bucket slice holding the gradient averaged over `total_views`"""
        self.bucket.detach_grads()
        out = pipeline.train_views(cams, self.model, pipe, bg_color, backward_fn, accumulator=self.acc, **kw)
        for p_, v in zip(self.bucket.params, self.bucket.views):
            if p_.grad is None or p_.grad.data_ptr() != v.data_ptr():
                raise RuntimeError("MultiViewStepExchange: a gradient did not land in the bucket (the rasterizer was not called "
                                   "through its raw / chained entry)")
        self.bucket.all_reduce(group=self.group, average_over=self.total_views)
        return out


class FactoredGradExchange:
    """Gradient exchange for ONE view per GPU per optimizer step (BASELINE config C4: 8 views over 8 GPUs), where
    nothing can hide a dense 59-floats-per-Gaussian all-reduce (every gradient is final only after the last backward
    kernel and the optimizer needs it before the next forward).  It moves 2.6x fewer bytes instead:

      * 48 of the 59 floats are the SH gradient, which for ONE view is the outer product of the 16 SH basis values of
        the viewing direction (a function of the replicated means and the view's camera centre, which every rank can
        evaluate) and the 3 floats dL/drgb.  The rasterizer's backward therefore delivers only dL/drgb [P,3]
        (diff_gaussian_rasterization.set_grad_sinks(..., sh_factor=...)); the ranks ALL-GATHER {dL/drgb | camera
        centre} (12 B per Gaussian per rank) and each rank rebuilds  (1/N) sum_v basis_v x drgb_v  with one kernel
        (msgs_sh_grad_from_views: the same products the backward would have formed, added in view order, so every rank
        holds bit-identical results);
      * the remaining 11 floats (xyz 3, opacity 1, scaling 3, rotation 4) go through one flat all-reduce (ncclAvg).

    Per GPU and step at 1 M Gaussians and 8 ranks: 7 x 12 MB received by the all-gather + 2 x 7/8 x 44 MB for the
    all-reduce = 161 MB, against 413 MB for the dense bucket.  The all-gather starts as soon as the factors are written
    — the library emits them with a kernel of its own in front of the per-Gaussian backward and signals an event
    (msgs_grads_t::factors_ready) — and so overlaps that kernel; the SH rows are rebuilt while the all-reduce of the
    small bucket is still in flight.  Exact: no quantisation, the same float32 terms in a different (fixed) summation order.

        ex = FactoredGradExchange(model, world)
        ex.begin_view(camera_center)          # BEFORE the forward: the sinks are snapshotted by the render call
        loss = criterion(render(...)); loss.backward()
        ex.end_view(); ex.finish()            # then optimizer.step()

    `model` carries the reference's leaf names (_xyz, _features_dc, _features_rest, _opacity, _scaling, _rotation) and
    active_sh_degree.  The rasterizer must be called through its raw / chained entry (the reference's getters or
    render_fused).  `reconstruct` / `set_sinks` are injection points for the CPU (gloo) tests of the exchange logic;
    the defaults are the HIP implementations and there is no CPU fallback."""

    SMALL = ("_xyz", "_opacity", "_scaling", "_rotation")

    def __init__(self, model, world=None, group=None, reconstruct=None, set_sinks=None, visible_rows=False,
                 visible_rows_max_fraction=0.85):
        """visible_rows: reduce only the rows of the small bucket that SOME rank rendered this step (end_view then needs
        each rank's visibility mask): the ranks all-gather their [P] visibility bytes, form the union, pack the union's
        rows of the 11 non-SH floats into one dense [U, 11] buffer, all-reduce that and scatter it back — rows no view
        rendered are zero on every rank and stay where they are.  Worth it at 2-4 ranks on scenes where a view sees a
        fraction of the model (C3-like frusta: 56 % rendered per view); when the union exceeds
        `visible_rows_max_fraction` of the model (C4: every ring camera sees 99.9 % of the ball) the step falls back to the
        dense small bucket.  Costs one host synchronisation per step (the size of the union)."""
        self.model = model
        self.group = group
        self.visible_rows = bool(visible_rows)
        self.visible_rows_max_fraction = float(visible_rows_max_fraction)
        self._packed = None                 # (idx, packed buffer) of the step in flight
        self.last_union_rows = None
        self.active = dist.is_available() and dist.is_initialized() and (
            dist.get_world_size(group) > 1 or os.environ.get("MSGS_EXCHANGE_FORCE") == "1")
        self.world = world if world is not None else (dist.get_world_size(group) if self.active else 1)
        self.n_rows = dist.get_world_size(group) if self.active else 1          # rows of the gathered buffer
        self.small = FlatGradBucket([getattr(model, n) for n in self.SMALL])
        self.small.detach_grads()
        xyz = model._xyz
        self.P = int(xyz.shape[0])
        dev = xyz.device
        row = 3 * self.P + 4                                                    # {drgb [P,3] | campos [3] | pad}
        self.send = torch.zeros(row, dtype=torch.float32, device=dev)
        self.gathered = torch.zeros(self.n_rows, row, dtype=torch.float32, device=dev) if self.active else \
            self.send.view(1, row)
        self.g_dc = torch.empty_like(model._features_dc)
        self.g_rest = torch.empty_like(model._features_rest)
        self.pending = []
        self.avg_op = None
        if self.active and dist.get_backend(group) == "nccl":
            try:
                dist.all_reduce(torch.ones(1, device=dev), op=dist.ReduceOp.AVG, group=group)
                self.avg_op = dist.ReduceOp.AVG
            except Exception:
                self.avg_op = None
        # overlap of the factor all-gather with the per-Gaussian backward (K9): the library writes the factors with a
        # kernel of its own in front of K9 and records `ready` behind it; the all-gather is issued from a side stream that
        # only waits for that event (HIP path only: the injected test doubles have no event)
        self.side = self.ready = None
        if reconstruct is None or set_sinks is None:
            import diff_gaussian_rasterization as dgr
            reconstruct = reconstruct or dgr.sh_grad_from_views
            if set_sinks is None and dev.type == "cuda":
                self.side = torch.cuda.Stream(device=dev)
                self.ready = torch.cuda.Event()
                self.ready.record(torch.cuda.current_stream(dev))       # creates the native handle
            set_sinks = set_sinks or dgr.set_grad_sinks
        self._reconstruct, self._set_sinks = reconstruct, set_sinks

    def begin_view(self, camera_center=None):
        """Call BEFORE the forward (render) of the view: the rasterizer snapshots the registered sinks at forward time.
        camera_center (optional here, else in end_view): written into the exchange row now, on the main stream, i.e.
        ordered before the backward and its `factors ready` event."""
        for n in ("_features_dc", "_features_rest"):
            getattr(self.model, n).grad = None
        self.small.detach_grads()
        self._cc_written = camera_center is not None
        if camera_center is not None:
            self.send[3 * self.P:3 * self.P + 3] = camera_center.to(self.send.device, torch.float32).reshape(3)
        if self.ready is not None:
            self._set_sinks(self.small.sinks(), sh_factor=self.send[:3 * self.P].view(self.P, 3), factors_ready=self.ready)
        else:
            self._set_sinks(self.small.sinks(), sh_factor=self.send[:3 * self.P].view(self.P, 3))

    def _small_exchange(self, visibility, op):
        """the small bucket's all-reduce: dense, or — visible_rows — over the rows rendered somewhere this step"""
        b = self.small
        self._packed = None
        if self.visible_rows and visibility is not None:
            vis8 = visibility.reshape(-1).to(torch.uint8).contiguous()
            allvis = torch.empty(self.n_rows, self.P, dtype=torch.uint8, device=vis8.device)
            dist.all_gather_into_tensor(allvis.view(-1), vis8, group=self.group)
            idx = allvis.max(dim=0).values.nonzero().squeeze(1)          # host sync: the size of the union
            self.last_union_rows = int(idx.numel())
            if self.last_union_rows <= self.visible_rows_max_fraction * self.P:
                packed = torch.cat([v.reshape(self.P, -1).index_select(0, idx) for v in b.views], dim=1).contiguous()
                self._packed = (idx, packed)
                return dist.all_reduce(packed, op=op, group=self.group, async_op=True)
        return dist.all_reduce(b.flat, op=op, group=self.group, async_op=True)

    def end_view(self, camera_center=None, visibility=None):
        """after backward(): issue both collectives (asynchronously; they run on the communicator's stream).
        visibility: this view's `visibility_filter` ([P] bool), needed by the visible_rows exchange only"""
        self._set_sinks(None)
        b = self.small
        for p, v in zip(b.params, b.views):
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                raise RuntimeError("FactoredGradExchange: a gradient did not land in the bucket — either begin_view() was "
                                   "called AFTER the forward (the sinks are snapshotted by the render call: begin_view(); "
                                   "render(); backward(); end_view()), or the rasterizer was not called through its raw / "
                                   "chained entry")
        written = getattr(self, "_cc_written", False)
        if camera_center is None and not written:
            raise ValueError("FactoredGradExchange: the view's camera_center was given neither to begin_view nor to end_view")
        if self.active:
            op = self.avg_op if self.avg_op is not None else dist.ReduceOp.SUM
            cc, converted = None, False
            if not written:
                converted = camera_center.device != self.send.device or camera_center.dtype != torch.float32
                cc = camera_center.to(self.send.device, torch.float32).reshape(3)
            if self.side is not None:
                # the factors are final once `ready` has fired (before K9 ends): gather them from the side stream,
                # concurrently with K9; the small bucket's all-reduce follows K9 on the main stream
                main = torch.cuda.current_stream(self.send.device)
                with torch.cuda.stream(self.side):
                    self.side.wait_event(self.ready)
                    if converted:                   # a real copy / cast was enqueued on the main stream: wait for it
                        self.side.wait_stream(main)
                    if cc is not None:
                        self.send[3 * self.P:3 * self.P + 3] = cc
                    ag = dist.all_gather_into_tensor(self.gathered.view(-1), self.send, group=self.group, async_op=True)
                self.send.record_stream(self.side)
                self.gathered.record_stream(self.side)
                ar = self._small_exchange(visibility, op)
                self.pending = [ag, ar]
                self._main = main
            else:
                if cc is not None:
                    self.send[3 * self.P:3 * self.P + 3] = cc
                self.pending = [dist.all_gather_into_tensor(self.gathered.view(-1), self.send, group=self.group, async_op=True),
                                self._small_exchange(visibility, op)]
        elif not written:
            self.send[3 * self.P:3 * self.P + 3] = camera_center.to(self.send.device, torch.float32).reshape(3)

    def finish(self):
        """wait for the exchange (stream-level for NCCL) and rebuild the SH gradient; afterwards every leaf's .grad
        holds the gradient averaged over the `world` views of this step"""
        if self.pending:
            self.pending[0].wait()                 # the factors of every rank have arrived (stream-level wait for NCCL)
        # the SH rows are rebuilt while the small bucket's all-reduce is still in flight
        self._reconstruct(self.model._xyz.detach(), self.gathered, self.n_rows, int(self.model.active_sh_degree),
                          1.0 / self.world, self.g_dc, self.g_rest)
        for w in self.pending[1:]:
            w.wait()
        self.pending = []
        if self._packed is not None:            # visible-rows exchange: the reduced rows go back where they came from
            idx, packed = self._packed
            self._packed = None
            if self.world > 1 and self.avg_op is None:
                packed.div_(self.world)
            col = 0
            for v in self.small.views:
                v2 = v.reshape(self.P, -1)
                v2.index_copy_(0, idx, packed[:, col:col + v2.shape[1]])
                col += v2.shape[1]
        elif self.world > 1 and (not self.active or self.avg_op is None):
            self.small.flat.div_(self.world)
        self.model._features_dc.grad = self.g_dc
        self.model._features_rest.grad = self.g_rest

    def bytes_per_step(self):
        """bytes a GPU receives per step: all-gather of the factors + ring-equivalent all-reduce of the small bucket"""
        n = max(self.n_rows, 1)
        return 4 * ((n - 1) * self.send.numel() + 2 * (n - 1) * self.small.flat.numel() // n)

    def bytes_last_step_visible_rows(self):
        """the same count for the last step when it went through the visible-rows exchange (None otherwise): factors +
        visibility bytes + the packed [U, 11] all-reduce"""
        if not self.visible_rows or self.last_union_rows is None or \
                self.last_union_rows > self.visible_rows_max_fraction * self.P:
            return None
        n = max(self.n_rows, 1)
        return 4 * (n - 1) * self.send.numel() + (n - 1) * self.P + 4 * 2 * (n - 1) * (11 * self.last_union_rows) // n


def gather_pixel_size_observations(visibility_filter, pixel_sizes, reso_lvl, group=None):
    """All-gather of what update_pixel_sizes (scene/gaussian_model.py:663-686) consumes from one view: returns
    (obs [N,P] float32, levels [N] int64) with obs[v] = pixel_sizes of rank v's view where visible, -1 where not
    (pixel sizes are >= 0, so the sign carries the mask) — 4 bytes per Gaussian per rank.

    The reference's update is ORDER DEPENDENT (every view that sees a Gaussian first decays max by 0.95 / relaxes min
    by 1.05, then folds its observation in), so two ways to apply N concurrent views are provided:
      * sequential (exact): apply the N observations in rank order with the single-view update — bit-identical on every
        rank and to a single-GPU run that renders the same N views in that order
        (train_epilogue.update_training_stats per row; see apply_pixel_size_observations_sequential);
      * batched (documented approximation, batched_pixel_size_update below): ONE decay per iteration, folded with the
        MAX / MIN over the views — differs from the sequential result by at most the factor 0.95^(k-1) resp. 1.05^(k-1)
        for a Gaussian seen by k > 1 views of the iteration."""
    obs = torch.where(visibility_filter, pixel_sizes.to(torch.float32), torch.full_like(pixel_sizes, -1.0, dtype=torch.float32))
    lvl = torch.tensor([int(reso_lvl)], dtype=torch.int64, device=obs.device)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return obs[None], lvl
    n = dist.get_world_size(group)
    allobs = torch.empty(n, obs.numel(), dtype=torch.float32, device=obs.device)
    dist.all_gather_into_tensor(allobs.view(-1), obs.contiguous(), group=group)
    lvls = torch.empty(n, dtype=torch.int64, device=obs.device)
    dist.all_gather_into_tensor(lvls, lvl, group=group)
    return allobs, lvls


def batched_pixel_size_update(max_pixel_sizes, min_pixel_sizes, target_reso_lvl, obs, reso_lvl, reso_lvls):
    """The batched variant of update_pixel_sizes for N views of ONE iteration rendered at the same level (in place):
        seen   = any_v obs[v] >= 0  and  target_reso_lvl == reso_lvl
        max    <- max(0.95 max, MAX_v obs[v])                       where seen, reso_lvl > 0
        grown  = clip(1.05 min, -1)
        min    <- valid ? (grown < 0 ? MIN_v+ obs : min(grown, MIN_v+ obs)) : grown     where seen, reso_lvl < L - 1
    with MIN_v+ over the views' valid (> 0) observations.  Equal to the reference's update when one view sees the
    Gaussian; with k views it applies the decay once instead of k times (see gather_pixel_size_observations)."""
    vis = obs >= 0
    seen = vis.any(dim=0) & (target_reso_lvl == reso_lvl)
    if reso_lvl > 0:
        ps_max = torch.where(vis, obs, torch.full_like(obs, -1.0)).max(dim=0).values
        new = torch.maximum(max_pixel_sizes * 0.95, ps_max)
        max_pixel_sizes.copy_(torch.where(seen, new, max_pixel_sizes))
    if reso_lvl < reso_lvls - 1:
        valid = vis & (obs > 0)
        ps_min = torch.where(valid, obs, torch.full_like(obs, float("inf"))).min(dim=0).values
        has = valid.any(dim=0)
        grown = torch.clip(min_pixel_sizes * 1.05, -1)
        new = torch.where(has, torch.where(grown < 0, ps_min, torch.minimum(grown, ps_min)), grown)
        min_pixel_sizes.copy_(torch.where(seen, new, min_pixel_sizes))


def apply_pixel_size_observations_sequential(model, obs, levels, update_fn=None):
    """Exact multi-view update: the single-view update of gaussian_model.py:663-686 for each gathered row in rank order.
    update_fn(model, visibility [P] bool, pixel_sizes [P], reso_lvl) defaults to the HIP statistics kernel
    (train_epilogue.update_training_stats with only the pixel-size group selected)."""
    if update_fn is None:
        from train_epilogue import update_training_stats

        def update_fn(model, vis, ps, lvl):
            radii = vis.to(torch.int32).contiguous()            # the kernel masks by radii > 0
            update_training_stats(model, None, radii, ps.contiguous(), lvl, update_pixel_sizes=True, densify=False)
    for v in range(obs.shape[0]):
        vis = obs[v] >= 0
        update_fn(model, vis, torch.where(vis, obs[v], torch.zeros_like(obs[v])), int(levels[v]))


def all_reduce_densification_stats(grad_norm_sum, vis_count, max_radii, group=None):
    """Training statistics that must stay equivalent between 1 and N GPUs:
    sum of per-view ||viewspace_points.grad[:, :2]|| and visibility counts (SUM; the norm is taken per
    view BEFORE summing, scene/gaussian_model.py:698-701) and max_radii2D (MAX, train.py:249)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return grad_norm_sum, vis_count, max_radii
    dist.all_reduce(grad_norm_sum, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(vis_count, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(max_radii, op=dist.ReduceOp.MAX, group=group)
    return grad_norm_sum, vis_count, max_radii


def views_for_rank(n_views, rank, world_size):
    """View v goes to rank v mod world_size."""
    return [v for v in range(n_views) if v % world_size == rank]
