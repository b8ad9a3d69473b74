"""Several views of ONE model in flight on the GPU at once: the two-stream view pipeline.

Every multi-view loop of the reference is dependency-free between its views — the insertion sweeps over all training
cameras (/root/reference/train.py:282-299,337-341), evaluation (:488-496), render.py:37-49, render_traj.py:99-105 —
and so are the views of one optimizer step when a rank holds several (BASELINE config C4 at 1 / 2 / 4 GPUs: 8 / 4 / 2
views per GPU per step).  The reference renders them one after the other.  One view leaves half of the machine idle at
any moment: its blend kernels are bound by instruction issue with HBM at 7 %, its per-Gaussian and binning kernels by
HBM or by launch latency with the vector units idle.  This module keeps TWO views in flight on two
HIP streams from one host thread:

  * forwards only LAUNCH (diff_gaussian_rasterization.deferred_forward -> msgs_forward_launch): the host does not wait
    for a view's instance count before it enqueues the next view; it collects the count (msgs_forward_finish) when the
    view's consumer or backward is due, by which time it has long landed;
  * software pipeline of depth two: the forward of view i+1 is enqueued BEFORE the backward of view i, on the other
    stream, so that at any time one stream is in a forward (per-Gaussian stage, sorts, blend forward) while the other is
    in a backward (blend backward, per-Gaussian backward);
  * the model's getters (exp / sigmoid / normalize / cat, /root/reference/scene/gaussian_model.py:127-153) are evaluated
    ONCE per sweep and shared by its views (`share_getters`), instead of once per render() call;
  * the views' gradients are summed INSIDE the per-Gaussian backward kernel into one bucket that becomes the parameters'
    .grad (diff_gaussian_rasterization.GradAccumulator): no `param.grad += g` passes after every view (six kernels,
    0.11 ms per view at 1 M Gaussians) and no zero rows for Gaussians a view did not render.

Results are bit-identical to rendering the same views one after the other on one stream
(tests/test_multi_view_gpu.py): every view runs the same kernels on its own buffers; only their interleaving on the
GPU changes.
"""
import torch

import diff_gaussian_rasterization as dgr
from gaussian_renderer import render as _render

_GETTERS = ("get_features", "get_opacity", "get_scaling", "get_rotation")
# the leaf parameters of the reference's GaussianModel (/root/reference/scene/gaussian_model.py:53-58)
LEAF_NAMES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")


class SharedGetters:
    """Proxy of a GaussianModel-like object whose four activated getters are evaluated once and then reused: the
    activations do not depend on the camera, a sweep over n views needs them once, not n times (torch.cat of the SH
    leaves alone is 0.11 ms and 384 MB of traffic at 1 M Gaussians).  The cached tensors keep their autograd history, so
    the op still recognises them as the reference's getters and chains their backward."""

    def __init__(self, pc):
        object.__setattr__(self, "_pc", pc)
        object.__setattr__(self, "_cache", {})

    def prime(self):
        for name in _GETTERS:
            getattr(self, name)
        return self

    def __getattr__(self, name):
        if name in _GETTERS:
            cache = object.__getattribute__(self, "_cache")
            if name not in cache:
                cache[name] = getattr(object.__getattribute__(self, "_pc"), name)
            return cache[name]
        return getattr(object.__getattribute__(self, "_pc"), name)

    def __setattr__(self, name, value):
        setattr(object.__getattribute__(self, "_pc"), name, value)


class ViewPipeline:
    """View lanes (one HIP stream each) + the launch order that keeps a forward and a backward in flight together.

    n_streams: 2 by default; 3 lanes measured 2-3 % faster for fwd+bwd at 1 M Gaussians / 1080p and 3 % slower forward-only.
    priorities: optional per-lane stream priorities (torch convention: lower = more urgent).
    What was measured and NOT kept (profiles/r4_notes.md): the blend kernels of every lane on low-priority streams of their
    own (no effect on who gets free wave slots, +20-40 us of event fences per kernel), and CU-masked blend streams that keep
    a few CUs per XCD free for the other lanes' kernels (cross-queue fences of masked queues cost hundreds of us)."""

    def __init__(self, device, n_streams=2, priorities=None):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("ViewPipeline needs a HIP device (there is no CPU path)")
        pr = list(priorities) if priorities is not None else [0] * n_streams
        if len(pr) != n_streams or n_streams < 1:
            raise ValueError("one priority per stream")
        self.streams = [torch.cuda.Stream(self.device, priority=p) for p in pr]

    # -- stream plumbing -------------------------------------------------------------------------------------------
    def _fork(self):
        cur = torch.cuda.current_stream(self.device)
        ev = torch.cuda.Event()
        ev.record(cur)
        for s in self.streams:
            s.wait_event(ev)
        return cur

    def _join(self, cur):
        for s in self.streams:
            cur.wait_stream(s)

    @staticmethod
    def _hand_over(obj, cur):
        """tensors allocated on a side stream and handed to the caller: tell the caching allocator that `cur` uses them.
        Covers what a view returns — a tensor, a result dict, a tuple / list of those — and the .grad of the leaf tensors
        among them (`viewspace_points.grad` is allocated by the backward on the lane's stream and read on the caller's by
        the statistics update, /root/reference/train.py:247-250)."""
        if torch.is_tensor(obj):
            if obj.is_cuda:
                obj.record_stream(cur)
                g = obj.grad if obj.is_leaf else None
                if g is not None and g.is_cuda:
                    g.record_stream(cur)
        elif isinstance(obj, dict):
            for v in obj.values():
                ViewPipeline._hand_over(v, cur)
        elif isinstance(obj, (tuple, list)):
            for v in obj:
                ViewPipeline._hand_over(v, cur)

    def _model(self, pc, share_getters, pipe=None, training=False):
        """share_getters in a TRAINING sweep is only sound when the op chains the getters' backward itself: the
        getters' autograd nodes are then never run.  If autograd had to run them, the first view's backward would free the graph
        the other views share.  So the sweep shares them only when the op will recognise them (checked here on the model's own
        tensors) and the pipeline settings keep the SH / covariance inside the op; train_views verifies every view afterwards."""
        if not share_getters:
            return pc
        shared = SharedGetters(pc).prime()
        if training:
            ok = bool(dgr.chain_reference_getters) and pipe is not None and not getattr(pipe, "convert_SHs_python", False) \
                and not getattr(pipe, "compute_cov3D_python", False) and torch.is_grad_enabled()
            ok = ok and dgr._match_reference_getters(shared.get_xyz, shared.get_features, shared.get_opacity,
                                                     shared.get_scaling, shared.get_rotation) is not None
            if not ok:
                return pc
        return shared

    # -- forward only ----------------------------------------------------------------------------------------------
    def render_views(self, cams, pc, pipe, bg_color, consume=None, render_fn=_render, share_getters=True, **settings):
        """Forward-only renders of `cams` (call it under torch.no_grad() for evaluation / sweeps), views alternating over
        the streams, view i+1 launched before view i's instance count is waited for.

        consume(i, pkg), when given, runs in view i's stream context right after the view is resolved (its results are
        final in that stream's order) and its return value is collected instead of the result dict — for sweeps over
        hundreds of cameras that only reduce each view (train.py:282-299) and must not keep every image alive.
        Returns the list of result dicts (or of consume()'s return values), usable on the caller's stream."""
        n = len(cams)
        ns = len(self.streams)
        model = self._model(pc, share_getters, pipe, training=torch.is_grad_enabled())
        cur = self._fork()
        out = [None] * n
        pkgs = [None] * n
        try:
            with dgr.deferred_forward() as pending:
                def launch(i):
                    with torch.cuda.stream(self.streams[i % ns]):
                        before = len(pending)
                        pkgs[i] = (render_fn(cams[i], model, pipe, bg_color, **settings), pending[before:])
                if n:
                    launch(0)
                for i in range(n):
                    if i + 1 < n:
                        launch(i + 1)
                    pkg, mine = pkgs[i]
                    pkgs[i] = None
                    for p_ in mine:
                        p_.resolve()
                    if consume is not None:
                        with torch.cuda.stream(self.streams[i % ns]):
                            out[i] = consume(i, pkg)
                    else:
                        out[i] = pkg
        finally:
            self._join(cur)                       # also when a view raised: the lanes may still be reading the model
        for o in out:
            self._hand_over(o, cur)
        return out

    # -- forward + backward ----------------------------------------------------------------------------------------
    def train_views(self, cams, pc, pipe, bg_color, backward_fn, render_fn=_render, share_getters=True,
                    accumulate_in_kernel=True, accumulator=None, **settings):
        """render() + backward of every view of one optimizer step; gradients accumulate into the parameters' .grad.

        accumulate_in_kernel: the views' leaf gradients are summed inside the per-Gaussian backward kernel into one bucket
        (diff_gaussian_rasterization.GradAccumulator) that becomes .grad at the end, instead of autograd's
        `param.grad += g` passes after every view (same order of additions, same bits); it applies to the calls the op
        recognises as the reference's getters or that use the raw-parameter entry — any other call is accumulated by
        autograd as usual.  accumulator: a caller-owned GradAccumulator (e.g. over the slices of a flat exchange bucket,
        view_parallel.MultiViewStepExchange) used instead of a fresh one, also for a single view.

        backward_fn(i, pkg) is called in view i's stream context once the view is resolved and must run the view's
        backward (e.g. `loss_of(pkg["render"], gt[i]).backward()` or `pkg["render"].backward(dL[i])`); its return value
        is collected.  Order of the enqueued work: fwd(0), fwd(1), bwd(0), fwd(2), bwd(1), ... — the forward of the next
        view on the other stream always goes out before a backward."""
        n = len(cams)
        ns = len(self.streams)
        model = self._model(pc, share_getters, pipe, training=True)
        acc = prev_acc = None
        if accumulator is not None or (accumulate_in_kernel and n > 1):
            acc = accumulator if accumulator is not None else dgr.GradAccumulator([getattr(pc, name) for name in LEAF_NAMES])
            acc.begin_step()                      # allocated on the caller's stream, ahead of the fork
            prev_acc = dgr.set_grad_accumulator(acc)
        cur = self._fork()
        out = [None] * n
        pkgs = [None] * n
        try:
            with dgr.deferred_forward() as pending:
                def launch(i):
                    with torch.cuda.stream(self.streams[i % ns]):
                        before = len(pending)
                        pkg = render_fn(cams[i], model, pipe, bg_color, **settings)
                        if model is not pc and n > 1:
                            fn = getattr(pkg.get("render"), "grad_fn", None)
                            # (the raw-parameter entry never reads the getters: sharing them is harmless there)
                            if fn is not None and not any(t in type(fn).__name__ for t in ("Chained", "Raw")):
                                raise RuntimeError(
                                    "ViewPipeline.train_views(share_getters=True): this render did not take the op's chained "
                                    "entry (override_color, a modified getter, ...), so autograd would run the shared getters' "
                                    "backward once per view and free their graph after the first: pass share_getters=False")
                        pkgs[i] = (pkg, pending[before:])
                if n:
                    launch(0)
                for i in range(n):
                    if i + 1 < n:
                        launch(i + 1)
                    pkg, mine = pkgs[i]
                    pkgs[i] = None
                    for p_ in mine:
                        p_.resolve()
                    with torch.cuda.stream(self.streams[i % ns]):
                        out[i] = backward_fn(i, pkg)
                    del pkg
        finally:
            if acc is not None:
                dgr.set_grad_accumulator(prev_acc)
            self._join(cur)                       # also when a view raised: the lanes may still be reading the model
        if acc is not None:
            acc.finish()
        for o in out:                             # whatever backward_fn returned: result dicts, losses, viewspace_points
            self._hand_over(o, cur)
        return out
