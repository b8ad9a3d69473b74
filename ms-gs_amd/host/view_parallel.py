"""View-parallel training support: every rank holds a full replica of the Gaussian parameters,
renders its own views, and the per-Gaussian gradients are summed across ranks with ONE flat
all-reduce (RCCL over xGMI on MI355X; `gloo` in the CPU tests).

This capability is new relative to the reference, which is single-GPU.  The
message is one fp32 bucket of 59 floats per Gaussian at SH degree 3 (3 xyz + 3 dc + 45 rest +
1 opacity + 3 scale + 4 rotation = 236 B): parameter .grad tensors are VIEWS into that bucket, so
autograd accumulates straight into it and no flatten/unflatten copy is needed around the collective.
"""
import torch
import torch.distributed as dist


class FlatGradBucket:
    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        self.views = []
        for p in self.params:
            v = self.flat[off:off + p.numel()].view_as(p)
            p.grad = v
            self.views.append(v)
            off += p.numel()

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
