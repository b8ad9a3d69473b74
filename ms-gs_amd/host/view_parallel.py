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
        for each view:  ex.begin_view(); loss.backward(); ex.end_view()
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
