#!/usr/bin/env python3
"""bench.py — Mpixels/s of rasterizer forward+backward (BASELINE.json metric) on N MI355X GPUs.

A "step" is one pass of the hot path over one synthetic view per GPU:
    render() [preprocess -> depth sort -> scan -> emit -> tile sort -> ranges -> blend fwd]
    -> backward [blend bwd -> preprocess bwd -> torch activations]  (+ grad all-reduce when N > 1)
driven through the reference's call surface (gaussian_renderer.render -> GaussianRasterizer -> C ABI).

Workload (config.workload):
  N = 1 : BASELINE.json configs[2] "C3": 1M Gaussians, 1920x1080, SH degree 3, multi-scale fields with
          filter_small + filter_large, fade_size 0 (train.py:124-125), the seeded synthetic scene of
          scenes.config("C3"), all inputs resident in HBM before the timed region.
  N > 1 : BASELINE.json configs[3] "C4": ONE shared 1M-Gaussian set (scenes.ball_scene, seed 4) replicated on every
          GPU and the 8 ring cameras of scenes.config_c4(); in step k rank r renders view (k*N + r) mod 8 — ONE view
          per GPU per step, weak scaling (per-GPU work fixed) — and the per-Gaussian gradients of the N views are
          exchanged over RCCL INSIDE the step (an optimizer could step after every step): `value` uses
          view_parallel.FactoredGradExchange (all-gather of the [P,3] dL/drgb factors + all-reduce of the 11 non-SH
          floats per Gaussian, SH rows rebuilt on every rank: 161 MB per GPU per step at 8 ranks instead of 413 MB).
          The JSON line also carries, measured the same way (SURVEY 8(e) "with and without the all-reduce"):
          `exchange.dense_serial_allreduce` (one flat 59-float all-reduce after every view),
          `exchange.dense_pipelined` (that all-reduce overlapped with the NEXT view: only valid when an optimizer step
          covers >= 2 views per GPU) and `exchange.without_exchange`.

Output: ONE JSON line on rank 0 (see the driver contract in the task statement) carrying `roofline`
(dominant blend kernel: algorithmic bytes / HIP-event time / 8 TB/s) and `cpu_baseline` (the CPU oracle
timed on this box's host cores on a bounded sample).
"""
import argparse
import contextlib
import gc
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host")):
    if p not in sys.path:
        sys.path.insert(0, p)

# the hosts of this pool only support dmabuf IPC: without this RCCL's buffer exchange between the ranks fails in
# hipIpcGetMemHandle (already exported on the driver's boxes; kept here for any other launcher)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

# thread pools sized for the CPUs this container may USE (its cgroup quota), not the ones it can see: with one OpenMP thread
# per visible CPU (128 on a 256-CPU box granted 16) the quota is spent spinning and the kernel parks the whole process —
# the kernel-launching thread included — for 15-70 ms at a time (hostinfo.py; profiles/r3_notes.md "host stalls")
from hostinfo import limit_thread_pools, usable_cpus  # noqa: E402  (the pools are sized after the launch decision below)


_GC_DEFAULT = gc.isenabled()


@contextlib.contextmanager
def quiet_gc():
    """Timed regions run with Python's cyclic garbage collector paused: a full collection of this process's heap takes 30-70 ms
    and lands, at a position fixed by the allocation count, inside some timed window of ~10 ms (seen as a 5.9 ms 'step' at one
    pyramid level; tools/diag_host_stalls.py).  Interpreter housekeeping, not part of a step.  The collection itself is done by
    settle_gc() BEFORE the warm-up steps, never between warm-up and timing: a 35 ms pause there lets the GPU clock down and the
    first timed steps pay for the ramp (measured: +0.07 ms per step over a 20-step region).  The collector is switched back on
    when the region ends (host memory must not grow across the many timed loops of one run)."""
    gc.disable()
    try:
        yield
    finally:
        if _GC_DEFAULT:
            gc.enable()


def settle_gc():
    """collect now (ahead of a warm-up loop) and keep the collector off until the following timed region (quiet_gc) ends"""
    gc.collect()
    gc.disable()


def period_median(fn, steps, warmup, sync):
    """Median PERIOD of `steps` back-to-back calls of fn(): one HIP event in front of every call and one behind the last, on the
    current stream; period k = time from event k to event k + 1.  Unlike an event pair around each call this also counts the gap
    in which the GPU waits for a host that cannot keep up (C1 / C2-sized steps), and unlike one wall-clock interval over the
    whole loop it is a median: one host stall (a cgroup throttle, a page fault) inside a 20-step loop moves a mean by 30 %
    (round 4: train_iteration 1.72 ms in the driver's run against 1.35-1.39 ms everywhere else) and a median not at all.
    Returns (median ms, sorted periods)."""
    import torch
    settle_gc()
    for _ in range(warmup):
        fn()
    sync()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    with quiet_gc():
        for k in range(steps):
            evs[k].record()
            fn()
        evs[steps].record()
        sync()
    ts = sorted(evs[k].elapsed_time(evs[k + 1]) for k in range(steps))
    n = len(ts)
    med = ts[n // 2] if n % 2 else 0.5 * (ts[n // 2 - 1] + ts[n // 2])
    return med, ts


def _self_launch_if_needed():
    """`python3 bench.py --gpus N` with N > 1 and no launcher around it (no WORLD_SIZE in the environment): start the N
    ranks HERE, as fresh child processes — one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment,
    exactly what `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` would give them — relay rank 0's
    JSON line and exit with the worst child status.  Runs before torch is imported: this process never touches the GPU
    (a process that has initialised HIP must not be replaced, and does not need to be: it only waits)."""
    n = 1
    for i, a in enumerate(sys.argv):
        if a == "--gpus" and i + 1 < len(sys.argv):
            n = int(sys.argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    # the ranks share this container's CPU quota (torch.distributed.run gives its workers OMP_NUM_THREADS=1 for the same reason)
    per_rank = os.environ.get("OMP_NUM_THREADS", str(max(1, usable_cpus() // n - 1)))
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS=per_rank, MKL_NUM_THREADS=per_rank)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
    # supervise every rank: a rank that dies before or inside a collective leaves its peers waiting in the rendezvous or in
    # RCCL until the process-group timeout — on the first failure (or after MSGS_BENCH_TIMEOUT seconds) the others are stopped
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("MSGS_BENCH_TIMEOUT", "3600"))
    codes = [None] * n
    failed = None
    while any(c is None for c in codes):
        for i, p_ in enumerate(procs):
            if codes[i] is None:
                codes[i] = p_.poll()
        bad = [c for c in codes if c not in (None, 0)]
        if bad or time.time() > deadline:
            failed = bad[0] if bad else 124
            for i, p_ in enumerate(procs):          # fresh children of this process: stopping them is safe
                if codes[i] is None:
                    p_.terminate()
            t_kill = time.time() + 10.0
            for i, p_ in enumerate(procs):
                if codes[i] is None:
                    try:
                        codes[i] = p_.wait(timeout=max(0.1, t_kill - time.time()))
                    except subprocess.TimeoutExpired:
                        p_.kill()
                        codes[i] = p_.wait()
            break
        time.sleep(0.05)
    reader.join(timeout=5.0)
    sys.stdout.write("".join(c for c in chunks if c))
    sys.stdout.flush()
    if failed is not None:
        print(f"[bench launcher] a rank exited with status {failed}; the other ranks were stopped", file=sys.stderr)
        raise SystemExit(failed)
    raise SystemExit(0)


if __name__ == "__main__":
    _self_launch_if_needed()
HOST_THREADS = limit_thread_pools()

import numpy as np
import torch
import torch.distributed as dist

from hostinfo import single_thread_backward  # noqa: E402
# one GPU per process: autograd's backward runs on the calling thread (no hand-off to the device thread; hostinfo.py).
# MSGS_BENCH_MT_BACKWARD=1 keeps torch's default for an A/B.
SINGLE_THREAD_BACKWARD = os.environ.get("MSGS_BENCH_MT_BACKWARD", "0") != "1"
if SINGLE_THREAD_BACKWARD:
    single_thread_backward()

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def csrc_sha256():
    """hash of the kernel sources this tree builds (same function in tools/summarize_rocprof.py, which stamps the
    committed counter summaries with it)"""
    import hashlib
    root = os.path.join(ROOT, "ms-gs_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(root)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(root, name), "rb").read())
    return h.hexdigest()


def cpu_baseline(scenes, scene, settings, W, H, runs=2):
    """The CPU oracle (kind "port": there is no reference CPU path) on the SAME workload: the full scene
    rendered forward+backward at full resolution (i) on all host cores, best of `runs` (about 1-6 s per run), and
    (ii) once on a single thread (about 30 s) — SURVEY §8(d) asks for both."""
    from oracle import oracle_ctypes as oc
    cam = scenes.front_camera(W, H)
    bg = torch.zeros(3)
    dL = scenes.grad_seed(W, H, 2)
    cores = usable_cpus()                  # affinity ∩ cgroup quota (16 of the 256 visible CPUs on the MI355X boxes)

    d_ref = [None]

    def run(nt):
        t0 = time.perf_counter()
        r = oc.rasterize(scene, cam, settings, bg, num_threads=nt)
        t1 = time.perf_counter()
        oc.backward(r, dL, num_threads=nt)
        t2 = time.perf_counter()
        d_ref[0] = r.num_instances          # the reference's duplication count (3-sigma rect): D_ref of SURVEY 8(d)
        del r
        return (t2 - t0, t1 - t0, t2 - t1)
    best = min((run(cores) for _ in range(runs)), key=lambda t: t[0])
    one = run(1)
    dt = best[0]
    return {"value": round(W * H / 1e6 / dt, 4), "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "seconds": round(dt, 2), "fwd_s": round(best[1], 2), "bwd_s": round(best[2], 2),
            "single_thread": {"value": round(W * H / 1e6 / one[0], 4), "cores": 1, "seconds": round(one[0], 2),
                              "fwd_s": round(one[1], 2), "bwd_s": round(one[2], 2)},
            "instances_reference_duplication": d_ref[0],
            "scaling_note": f"a checker, not a tuned baseline: {cores} threads are {one[0] / dt:.1f}x one thread "
                            "(double atomics in the backward, serial sort stages); cores = the CPUs this container may "
                            f"use (cgroup quota), of {os.cpu_count()} visible",
            "sample": f"the whole workload (same scene, settings, {W}x{H}), forward+backward, float32 "
                      f"C++/OpenMP oracle: the {cores} usable host cores (best of {runs} runs) and one thread (one run)"}


def pyramid_timing(scenes, pc, settings, bg, dev, steps=11, warmup=3):
    """What MS-GS is for, measured the reference's way (train.py:488-496,541 logs render_time per scale; viewer.py:67-81 and
    render_traj.py:99-105 time a forward-only render() between synchronisations): the SAME 1 M-Gaussian scene at the pyramid
    levels k = 0..6 ((W, H) = (int(1920 / 2^k), int(1080 / 2^k)), utils/camera_utils.py:38-39):
      pyramid_ms          render() + backward() per level, training settings (filters on, fade 0)
      render_forward_ms   forward-only render() under no_grad: "filters_on" = the viewer's --anti_alias (filter_small =
                          filter_large = True, fade_size 1.0: viewer.py:59-72), "filters_off" = render.py's defaults"""
    from gaussian_renderer import PIPE, render, render_fused

    def timed(fn):
        return round(period_median(fn, steps, warmup, torch.cuda.synchronize)[0], 4)

    fb, on, off, sizes, vis, fused_fb, fused_on = [], [], [], [], [], [], []
    aa = dict(filter_small=True, filter_large=True, fade_size=1.0)
    plain = dict(filter_small=False, filter_large=False, fade_size=1.0)
    for k in range(7):
        W, H = int(1920 / 2 ** k), int(1080 / 2 ** k)
        cam = scenes.front_camera(W, H).to(dev)
        dL = scenes.grad_seed(W, H, 40 + k).to(dev)

        def train_step():
            for p_ in pc.parameters():
                p_.grad = None
            render(cam, pc, PIPE, bg, **settings)["render"].backward(dL)
        fb.append(timed(train_step))

        def fused_train_step():
            for p_ in pc.parameters():
                p_.grad = None
            render_fused(cam, pc, PIPE, bg, **settings)["render"].backward(dL)
        fused_fb.append(timed(fused_train_step))
        with torch.no_grad():
            fused_on.append(timed(lambda: render_fused(cam, pc, PIPE, bg, **aa)))
            on.append(timed(lambda: render(cam, pc, PIPE, bg, **aa)))
            off.append(timed(lambda: render(cam, pc, PIPE, bg, **plain)))
            vis.append(int((render(cam, pc, PIPE, bg, **settings)["radii"] > 0).sum().item()))
        sizes.append([W, H])
    for p_ in pc.parameters():
        p_.grad = None
    # the filters-off render at level 0 with and without the occlusion cut-off (review item 2): instance counts and times
    occ = None
    try:
        import diff_gaussian_rasterization as dgr
        W0, H0 = sizes[0]
        cam0_ = scenes.front_camera(W0, H0).to(dev)
        key0 = (dev.index, int(pc.get_xyz.shape[0]), W0, H0, 0, 0)

        def count_and_time(switch):
            prev_s = dgr._C.lib.msgs_set_occlusion(switch)
            try:
                dgr._last_instances.pop(key0, None)
                with torch.no_grad():
                    t_ = period_median(lambda: render(cam0_, pc, PIPE, bg, **plain), 5, 3, torch.cuda.synchronize)[0]
                return int(dgr._last_instances.get(key0, -1)), round(t_, 4)
            finally:
                dgr._C.lib.msgs_set_occlusion(prev_s)
        d_on, t_on = count_and_time(1)
        d_off, t_off = count_and_time(0)
        occ = {"instances": d_on, "ms": t_on, "instances_uncut": d_off, "ms_uncut": t_off,
               "what": "level 0, render.py's default flags: with the exact per-tile occlusion cut-off (default) and with it switched off "
                       "(msgs_set_occlusion(0)); bit-identical images (tests/test_occlusion_gpu.py)"}
        torch.cuda.empty_cache()
    except Exception as e:      # informational
        occ = {"error": repr(e)}
    # Informational, NOT the BASELINE scene: the same model after MS-GS's own bookkeeping has seen it.  The C3 recipe (SURVEY
    # 8(d)) gives min_pixel_sizes to half of the level-0 Gaussians only; in a trained MS-GS model update_pixel_sizes
    # (scene/gaussian_model.py:663-686) has given one to every level-0 Gaussian that was ever visible — its smallest observed
    # footprint at level 0 — which is what lets filter_small drop them at the coarser levels.  Here: every level-0 Gaussian
    # carries 0.8 x its level-0 pixel size of this camera; the inserted coarse-level Gaussians are untouched.
    trained = None
    try:
        cam0 = scenes.front_camera(1920, 1080).to(dev)
        with torch.no_grad():
            ps0 = render(cam0, pc, PIPE, bg, **plain)["pixel_sizes"]
        keep_min = pc.min_pixel_sizes
        lvl0 = pc.max_pixel_sizes < 0                  # level-0 Gaussians: no max_pixel_sizes in the C3 recipe
        pc.min_pixel_sizes = torch.where(lvl0 & (ps0 > 0), 0.8 * ps0, keep_min).contiguous()
        fb2, vis2 = [], []
        for k in range(7):
            W, H = int(1920 / 2 ** k), int(1080 / 2 ** k)
            cam = scenes.front_camera(W, H).to(dev)
            dL = scenes.grad_seed(W, H, 40 + k).to(dev)

            def train_step2():
                for p_ in pc.parameters():
                    p_.grad = None
                render(cam, pc, PIPE, bg, **settings)["render"].backward(dL)
            fb2.append(timed(train_step2))
            with torch.no_grad():
                vis2.append(int((render(cam, pc, PIPE, bg, **settings)["radii"] > 0).sum().item()))
        pc.min_pixel_sizes = keep_min
        for p_ in pc.parameters():
            p_.grad = None
        trained = {"ms": fb2, "rendered_gaussians": vis2,
                   "what": "informational, not the BASELINE scene: the same model with min_pixel_sizes on EVERY level-0 Gaussian "
                           "(0.8 x its level-0 pixel size), as MS-GS's update_pixel_sizes leaves a trained model — the case the "
                           "filters and the compacting sort are built for"}
    except Exception as e:      # informational only
        trained = {"error": repr(e)}
    return ({"levels": sizes, "ms": fb, "rendered_gaussians": vis, "render_fused_ms": fused_fb,
             "with_min_pixel_sizes_on_every_level0_gaussian": trained,
             "what": "render() + backward() per pyramid level k = 0..6 of the C3 scene, training settings; render_fused_ms = the "
                     "same through the raw-parameter entry (activations and SH concatenation inside the kernels)"},
            {"levels": sizes, "filters_on": on, "filters_off": off, "render_fused_filters_on": fused_on,
             "filters_off_occlusion": occ,
             "what": "forward-only render() under no_grad per pyramid level (viewer.py:67-81 convention): filters_on = "
                     "--anti_alias (both filters, fade_size 1.0), filters_off = render.py defaults; render_fused_filters_on = "
                     "the raw-parameter entry with the viewer's filters"})


def two_view_timing(pc, cam, bg, dL, settings, W, H, warmup, whole_step, views=8, rounds=7):
    """ms per view of sweeps over `views` views of the same model with two of them in flight (ViewPipeline: the forward of view
    i+1 enqueued before the backward of view i on the other stream, no host wait for the instance counts, getters once per sweep,
    gradients summed inside the per-Gaussian backward kernel), against the same sweep done the reference's way — one view after
    the other on one stream (train.py:282-299,337-341,488-496; the views of one optimizer step at C4 on fewer than 8 GPUs)."""
    from gaussian_renderer import PIPE, render, render_fused
    from multi_view import ViewPipeline
    cams = [cam] * views
    pipe = ViewPipeline(cam.world_view_transform.device, n_streams=2)

    def zero():
        for p_ in pc.parameters():
            p_.grad = None

    def timed(fn):      # median sweep period / views
        return round(period_median(fn, rounds, max(1, warmup // 2), torch.cuda.synchronize)[0] / views, 4)

    bwd = lambda i, pkg: pkg["render"].backward(dL)

    def serial(fn):
        zero()
        for c in cams:
            fn(c, pc, PIPE, bg, **settings)["render"].backward(dL)

    def piped(fn):
        zero()
        pipe.train_views(cams, pc, PIPE, bg, bwd, render_fn=fn, **settings)
    out = {"views_per_sweep": views, "lanes": 2,
           "serial_fwd_bwd_ms_per_view": timed(lambda: serial(render)),
           "fwd_bwd_ms_per_view": timed(lambda: piped(render)),
           "fused_serial_fwd_bwd_ms_per_view": timed(lambda: serial(render_fused)),
           "fused_fwd_bwd_ms_per_view": timed(lambda: piped(render_fused))}
    zero()
    with torch.no_grad():
        out["serial_forward_ms_per_view"] = timed(lambda: [render(c, pc, PIPE, bg, **settings) for c in cams])
        out["forward_ms_per_view"] = timed(lambda: pipe.render_views(cams, pc, PIPE, bg, **settings))
        out["fused_forward_ms_per_view"] = timed(lambda: pipe.render_views(cams, pc, PIPE, bg, render_fn=render_fused,
                                                                           share_getters=False, **settings))
    # the same per pyramid level k = 1..6 of the scene (what pyramid_ms / render_forward_ms report for the serial call pattern):
    # at the low levels a view is mostly latency-bound per-Gaussian and binning kernels, which two lanes overlap almost fully
    try:
        import scenes as _sc
        lv_fb, lv_f, lv_sizes = [out["fwd_bwd_ms_per_view"]], [out["forward_ms_per_view"]], [[W, H]]
        for k in range(1, 7):
            Wk, Hk = int(W / 2 ** k), int(H / 2 ** k)
            ck = [_sc.front_camera(Wk, Hk).to(cam.world_view_transform.device)] * views
            dk = _sc.grad_seed(Wk, Hk, 40 + k).to(cam.world_view_transform.device)

            def piped_k():
                zero()
                pipe.train_views(ck, pc, PIPE, bg, lambda i, pkg: pkg["render"].backward(dk), **settings)
            lv_fb.append(timed(piped_k))
            zero()
            with torch.no_grad():
                lv_f.append(timed(lambda: pipe.render_views(ck, pc, PIPE, bg, **settings)))
            lv_sizes.append([Wk, Hk])
        out["pyramid"] = {"levels": lv_sizes, "fwd_bwd_ms_per_view": lv_fb, "forward_ms_per_view": lv_f,
                          "what": "two lanes per pyramid level k = 0..6, training settings (compare pyramid_ms.ms and "
                                  "render_forward_ms of the serial call pattern)"}
    except Exception as e:      # informational
        out["pyramid"] = {"error": repr(e)}
    zero()
    out["value_fwd_bwd"] = round(W * H / 1e6 / (out["fwd_bwd_ms_per_view"] * 1e-3), 3)
    out["unit"] = "Mpixels/s"
    if whole_step and whole_step.get("algorithmic_bytes"):
        b = whole_step["algorithmic_bytes"]
        out["whole_step"] = {"algorithmic_bytes": int(b), "GBps": round(b / (out["fwd_bwd_ms_per_view"] * 1e-3) / 1e9, 1),
                             "frac": round(b / (out["fwd_bwd_ms_per_view"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    out["what"] = ("sweeps over 8 copies of the C3 view, ms per view: the serial loop (one stream, gradients accumulated by autograd "
                   "over the sweep) vs host/multi_view.ViewPipeline (two lanes, deferred forwards, getters once per sweep, gradients "
                   "summed inside the per-Gaussian backward); bit-identical results (tests/test_multi_view_gpu.py); informational — "
                   "`value` stays one view per step")
    return out


def algorithmic_bytes(P, W, H, D, D_trav, V):
    """SURVEY 8(d) per-kernel ALGORITHMIC bytes of one forward + backward (SH degree 3): P Gaussians, V rendered, D tile
    instances, D_trav = sum over tiles of the longest list prefix a pixel of the tile walks, N pixels."""
    N = W * H
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    S_sh = 12 * 16
    tb = max(1, math.ceil(math.log2(tiles)))
    return {"preprocess": P * (117 + S_sh) + V * 79, "scan": 8 * P, "emit": 12 * D,
            "sort(depth+tile)": D * (24 * math.ceil((32 + tb) / 8) + 8), "ranges": 8 * D + 8 * tiles,
            "blend_fwd": 48 * D_trav + 28 * N + 8 * tiles, "blend_bwd": 48 * D_trav + 20 * N + 36 * V,
            "preprocess_bwd": V * (309 + 36 + 24) + V * 236}


def binning_counts(dgr, ctx, P, dev):
    """(D, D_trav, V) of the forward whose autograd node is `ctx` (msgs_binning_stats)"""
    import ctypes as C
    geom, binning, image, D = ctx.state
    scratch = torch.empty(256, dtype=torch.uint8, device=dev)
    o = (C.c_int64 * 2)()
    dgr._C.check(dgr._C.lib.msgs_binning_stats(C.byref(ctx.call.view), P, C.c_void_p(ctx.radii.data_ptr()),
                                               C.c_void_p(binning.data_ptr()), binning.numel(),
                                               C.c_void_p(image.data_ptr()), image.numel(),
                                               C.c_void_p(scratch.data_ptr()), scratch.numel(), o,
                                               C.c_void_p(torch.cuda.current_stream().cuda_stream)), "stats")
    return int(D), int(o[0]), int(o[1])


def config_leg(name, scenes, dev, steps=20, warmup=3):
    """One of the other BASELINE configs through the same call surface (render() + backward of the fixed dL/dimage), for the
    driver's line: median period of `steps` steps, instance counts, per-kernel HIP-event times, whole-step algorithmic bytes
    against HBM, peak device memory.  C2 = BASELINE configs[1] (100 k Gaussians, 800x800, SH 3); C5 = configs[4] (5 M, 3840x2160,
    multi-scale levels, filters on)."""
    import diff_gaussian_rasterization as dgr
    from gaussian_renderer import PIPE, render
    from synthetic_model import SyntheticGaussians
    sc, cam, st = scenes.config(name)
    W, H, P = cam.image_width, cam.image_height, sc.P
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats(dev)
    pc = SyntheticGaussians(sc, dev, requires_grad=True)
    cam = cam.to(dev)
    bg = torch.zeros(3, device=dev)
    dL = scenes.grad_seed(W, H, {"C2": 1, "C5": 5}.get(name, 0)).to(dev)

    def step():
        for p_ in pc.parameters():
            p_.grad = None
        out = render(cam, pc, PIPE, bg, **st)
        out["render"].backward(dL)
        return out
    # depth-slab binning: the wrapper's adaptive policy engages it from the library's D / D_trav feedback — three
    # warm-up frames are what it needs; the same steps single-pass for comparison
    single = None
    if name == "C5":
        prev_pol, dgr.slab_policy = dgr.slab_policy, "never"
        try:
            single = period_median(step, steps, warmup, torch.cuda.synchronize)[0]
        finally:
            dgr.slab_policy = prev_pol
    med, ts = period_median(step, steps, max(warmup, 4), torch.cuda.synchronize)
    timers = [dgr._C.KernelTimer() for _ in range(3)]
    for t_ in timers:
        dgr._C.set_timer(t_)
        step()
    dgr._C.set_timer(None)
    torch.cuda.synchronize()
    acc = {}
    for t_ in timers:
        for k, v in t_.read_ms().items():
            if v >= 0:
                acc.setdefault(k, []).append(v)
    kernels = {k: round(float(np.mean(v)), 4) for k, v in acc.items()}
    out = step()
    torch.cuda.synchronize()
    D, D_trav, V = binning_counts(dgr, out["render"].grad_fn, P, dev)
    slab = None
    try:
        import ctypes as C_
        ctx_ = out["render"].grad_fn
        geom_ = dgr._resolve(ctx_.state)[0]
        o_ = (C_.c_int64 * 6)()
        dgr._C.check(dgr._C.lib.msgs_slab_stats(C_.c_void_p(geom_.data_ptr()), geom_.numel(), P, o_,
                                                C_.c_void_p(torch.cuda.current_stream().cuda_stream)), "msgs_slab_stats")
        if int(o_[0]):
            slab = {"fraction": round(float(ctx_.call.view.slab_fraction), 4), "ranks_in_slab_a": int(o_[1]),
                    "instances_slab_a": int(o_[2]), "tiles_left_open": int(o_[3]), "instances_slab_b": int(o_[4]),
                    "instances_binned": int(o_[2]) + int(o_[4]), "instances_single_pass": D,
                    "ms_per_step_single_pass": round(single, 4) if single is not None else None,
                    "how": "wrapper's adaptive policy (slab_policy = 'adaptive'): engaged from the library's D / D_trav feedback"}
    except Exception as e:                  # informational
        slab = {"error": repr(e)}
    alg = algorithmic_bytes(P, W, H, D, D_trav, V)
    total = float(sum(alg.values()))
    peak = int(torch.cuda.max_memory_allocated(dev))
    res = {"workload": {"C2": "BASELINE configs[1]: 100k Gaussians, 800x800, SH3, fwd+bwd",
                        "C5": "BASELINE configs[4]: 5M Gaussians, 3840x2160, multi-scale levels, filter_small+filter_large"}.get(name, name),
           "ms_per_step": round(med, 4), "min_ms": round(ts[0], 4), "p90_ms": round(ts[int(0.9 * (len(ts) - 1))], 4),
           "value": round(W * H / 1e6 / (med * 1e-3), 3), "unit": "Mpixels/s", "steps": steps,
           "how": "median period of the timed steps (HIP events in front of every step)",
           "gaussians": P, "width": W, "height": H, "instances": D, "D_trav": D_trav, "rendered": V, "kernel_ms": kernels,
           "sum_kernel_ms": round(float(sum(kernels.values())), 4),
           "whole_step": {"algorithmic_bytes": int(total), "GBps": round(total / (med * 1e-3) / 1e9, 1),
                          "frac": round(total / (med * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
           "peak_device_bytes": peak}
    if slab is not None:
        res["depth_slabs"] = slab
    elif single is not None:
        res["depth_slabs"] = {"engaged": False, "ms_per_step_single_pass": round(single, 4)}
    del pc, out, dL
    torch.cuda.empty_cache()
    return res


def reference_schedule_leg(scenes, dev, iters=300, cams_per_level=25, p_change_every=50):
    """The op under the reference's REAL call schedule (/root/reference/train.py:152-216,244-264), not a fixed view: the pyramid
    level — a new W x H — is drawn whenever the camera stack runs empty (75 % level 0, otherwise the least-trained level, :152-194;
    a stack of `cams_per_level` cameras here), and the model changes size every densification interval (clone / split / prune:
    +-3 % of random rows every `p_change_every` iterations here, :253-264).  Per variant (reference API render(), raw-parameter
    render_fused()): median, p99 and max period of an iteration (render + backward of a fixed dL/dimage; the iteration that
    follows a model rebuild is not counted — the rebuild is host work of the harness), and how many forwards could NOT take the
    speculative route (first visit of a view shape; afterwards the wrapper scales the last count of that shape by P)."""
    import random as _random
    import diff_gaussian_rasterization as dgr
    from gaussian_renderer import PIPE, render, render_fused
    from synthetic_model import SyntheticGaussians
    sc0, _, st = scenes.config("C3")
    levels = [(int(1920 / 2 ** k), int(1080 / 2 ** k)) for k in range(7)]
    cams = [scenes.front_camera(w, h).to(dev) for w, h in levels]
    dLs = [scenes.grad_seed(w, h, 40 + k).to(dev) for k, (w, h) in enumerate(levels)]
    bg = torch.zeros(3, device=dev)
    out = {"iterations": iters, "cameras_per_level_stack": cams_per_level, "model_size_change_every": p_change_every,
           "what": "train.py's schedule on the C3 model: level drawn per camera stack (75 % level 0, else the least trained), "
                   "P changed by +-3 % (random clone / prune) every 50 iterations; ms per iteration = render + backward"}
    for name, fn in (("reference_api", render), ("fused", render_fused)):
        rng = _random.Random(1234)
        g = torch.Generator().manual_seed(99)
        sc = sc0
        pc = SyntheticGaussians(sc, dev, requires_grad=True)
        trained = [0] * len(levels)
        stack, lvl = 0, 0
        before = dict(dgr.forward_stats)
        ev, skip, switches, changes, sizes = [], set(), 0, 0, [sc.P]
        settle_gc()
        with quiet_gc():
            for it in range(iters):
                if it and it % p_change_every == 0:                 # densify / prune: a new model of a slightly different size
                    P = sc.P
                    Pn = int(P * (1.0 + (0.03 if rng.random() < 0.5 else -0.03)))
                    idx = torch.randperm(P, generator=g)[:min(P, Pn)]
                    if Pn > P:
                        idx = torch.cat([idx, torch.randint(0, P, (Pn - P,), generator=g)])
                    sc = sc.subset(idx)
                    pc = SyntheticGaussians(sc, dev, requires_grad=True)
                    torch.cuda.synchronize()
                    skip.add(len(ev) - 1)                           # (the rebuild falls into the period that began with the last event)
                    changes += 1
                    sizes.append(sc.P)
                if stack == 0:                                      # camera stack empty: draw the level (train.py:152-194)
                    new = 0 if rng.random() < 0.75 else min(range(1, len(levels)), key=lambda k: (trained[k], k))
                    switches += new != lvl
                    lvl, stack = new, cams_per_level
                stack -= 1
                trained[lvl] += 1
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                ev.append(e)
                for p_ in pc.parameters():
                    p_.grad = None
                fn(cams[lvl], pc, PIPE, bg, **st)["render"].backward(dLs[lvl])
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            ev.append(e)
            torch.cuda.synchronize()
        raw = [(ev[i].elapsed_time(ev[i + 1]), i) for i in range(iters) if i not in skip]
        per = sorted(t for t, _ in raw)
        # the same without the FIRST iteration on a new model (and the very first): what the op itself adds at a size change
        # against what the first touch of new tensor sizes costs in torch's allocator and in the wrapper's one-off checks
        first_on_new = {0} | {i for i in range(iters) if i % p_change_every == 0}
        steady = sorted(t for t, i in raw if i not in first_on_new)
        slowest = sorted(raw, reverse=True)[:6]
        after = dgr.forward_stats
        out[name] = {"median_ms": round(per[len(per) // 2], 4), "p99_ms": round(per[int(0.99 * (len(per) - 1))], 4),
                     "max_ms": round(per[-1], 4), "slowest_iterations": [[i, round(t, 3)] for t, i in slowest],
                     "p99_ms_without_first_iteration_on_a_new_model": round(steady[int(0.99 * (len(steady) - 1))], 4),
                     "level_switches": switches, "model_size_changes": changes,
                     "model_sizes": [sizes[0], min(sizes), max(sizes)], "iterations_per_level": trained,
                     "forwards": after["forwards"] - before["forwards"],
                     "non_speculative_forwards": after["non_speculative"] - before["non_speculative"]}
        del pc
        torch.cuda.empty_cache()
    return out


def verification_mode_leg(scenes, dev, steps=5, warmup=2):
    """What msgs_set_deterministic(1) costs: the C3 step through the literal verification kernels (ms-gs_amd/csrc/literal.hip)."""
    import diff_gaussian_rasterization as dgr
    from gaussian_renderer import PIPE, render
    from synthetic_model import SyntheticGaussians
    sc, cam, st = scenes.config("C3")
    pc = SyntheticGaussians(sc, dev, requires_grad=True)
    cam = cam.to(dev)
    bg = torch.zeros(3, device=dev)
    dL = scenes.grad_seed(cam.image_width, cam.image_height, 2).to(dev)

    def step():
        for p_ in pc.parameters():
            p_.grad = None
        render(cam, pc, PIPE, bg, **st)["render"].backward(dL)
    prev = dgr.set_deterministic(True)
    try:
        med = period_median(step, steps, warmup, torch.cuda.synchronize)[0]
    finally:
        dgr.set_deterministic(prev)
    del pc
    torch.cuda.empty_cache()
    return {"ms_per_step": round(med, 3),
            "what": "C3 forward + backward in the verification mode (msgs_set_deterministic: the reference's blend loops restated "
                    "literally, exp in double, double sums in a fixed order); against the float32 oracle evaluated the same way the "
                    "forward is bit-identical and every gradient tensor within 3e-7 at C2 / C3 / C5 / C4 (tests/test_literal_gpu.py)"}


def train_iteration_timing(scenes, scene, cam, bg, settings, W, H, dev, steps, warmup, multi_view=True):
    import torch.nn.functional as F
    from gaussian_renderer import render
    from synthetic_model import SyntheticGaussians
    from train_epilogue import FusedAdam
    from train_step import fused_train_iteration
    from gaussian_renderer import PIPE
    gt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(3)).to(dev)
    out = {}

    def timed(fn):
        return period_median(fn, steps, warmup, torch.cuda.synchronize)[0]

    model = SyntheticGaussians(scene, dev)
    opt = FusedAdam(model.training_setup(7, scene.target_reso_lvl), lr=0.0, eps=1e-15)
    out["ms_per_iteration_optimizer_as_its_own_launch"] = round(timed(lambda: fused_train_iteration(model, opt, cam, gt, PIPE, bg, **settings)), 4)
    del model, opt
    # the same iteration with the Adam step taken INSIDE the per-Gaussian backward kernel (msgs_adam_in_backward_t): the
    # 236 B of gradient per Gaussian are neither written nor read back; parameters and moments bit-identical
    # (tests/test_train_step_gpu.py)
    model = SyntheticGaussians(scene, dev)
    opt = FusedAdam(model.training_setup(7, scene.target_reso_lvl), lr=0.0, eps=1e-15)
    out["ms_per_iteration"] = round(timed(lambda: fused_train_iteration(model, opt, cam, gt, PIPE, bg, step_in_backward=True,
                                                                        **settings)), 4)
    del model, opt

    model = SyntheticGaussians(scene, dev)
    opt = torch.optim.Adam(model.training_setup(7, scene.target_reso_lvl), lr=0.0, eps=1e-15)
    taps = torch.tensor([math.exp(-(k - 5) ** 2 / (2 * 1.5 ** 2)) for k in range(11)])       # loss_utils.py:23-30
    taps = (taps / taps.sum()).unsqueeze(1)
    w = taps.mm(taps.t()).float().to(dev).expand(3, 1, 11, 11).contiguous()
    conv = lambda t: F.conv2d(t, w, padding=5, groups=3)

    def torch_composition():
        pkg = render(cam, model, PIPE, bg, **settings)
        x = pkg["render"]
        m1, m2 = conv(x), conv(gt)
        m1s, m2s, m12 = m1.pow(2), m2.pow(2), m1 * m2
        s1, s2, s12 = conv(x * x) - m1s, conv(gt * gt) - m2s, conv(x * gt) - m12
        S = ((2 * m12 + 0.01 ** 2) * (2 * s12 + 0.03 ** 2)) / ((m1s + m2s + 0.01 ** 2) * (s1 + s2 + 0.03 ** 2))
        loss = 0.8 * torch.abs(x - gt).mean() + 0.2 * (1.0 - S.mean())
        loss.backward()
        with torch.no_grad():
            vis, radii, ps = pkg["visibility_filter"], pkg["radii"], pkg["pixel_sizes"]
            mask = vis & (model.target_reso_lvl == 0)
            mn = torch.clip(model.min_pixel_sizes[mask] * 1.05, -1)
            model.min_pixel_sizes[mask] = torch.where(ps[mask] > 0, torch.where(mn < 0, ps[mask], torch.min(mn, ps[mask])), mn)
            model.max_radii2D[vis] = torch.max(model.max_radii2D[vis], radii[vis])
            model.xyz_gradient_accum[:, 0][vis] += torch.norm(pkg["viewspace_points"].grad[vis, :2], dim=-1, keepdim=True)
            model.denom[:, 0][vis] += 1
            opt.step()
            opt.zero_grad(set_to_none=True)
    out["ms_per_iteration_torch_composition"] = round(timed(torch_composition), 4)
    # the same GPU pieces with ONE optimizer step over 8 views, two in flight (train_step.fused_train_iteration_views)
    try:
        if not multi_view:
            raise RuntimeError("skipped (--no-two-view)")
        from multi_view import ViewPipeline
        from train_step import fused_train_iteration_views
        model = SyntheticGaussians(scene, dev)
        opt = FusedAdam(model.training_setup(7, scene.target_reso_lvl), lr=0.0, eps=1e-15)
        vp_ = ViewPipeline(dev)
        cams8, gts8 = [cam] * 8, [gt] * 8
        t_ = timed(lambda: fused_train_iteration_views(model, opt, vp_, cams8, gts8, PIPE, bg, **settings))
        out["ms_per_view_8_views_per_optimizer_step"] = round(t_ / 8, 4)
        del model, opt
    except Exception as e:      # informational
        out["ms_per_view_8_views_per_optimizer_step"] = repr(e)
    out["note"] = ("C3 scene, fixed U(0,1) target, lambda_dssim 0.2, level 0, statistics on; informational; every figure is the "
                   "median period of the timed iterations (period_median); ms_per_iteration = train_step.fused_train_iteration("
                   "step_in_backward=True): render_fused + L1/SSIM loss + backward with the Adam step inside its per-Gaussian "
                   "kernel + statistics")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 30 untimed + 50 timed steps (0.1 s of GPU time).  Five warm-up steps are not enough after the host-side scene
    # build: the first ~20 steps run on a GPU whose clocks have dropped (measured on one box, same process otherwise:
    # --warmup 5 --steps 20 -> 1.083 ms per step, --warmup 50 --steps 20 -> 1.049, median_of_50 1.048 in both)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-pyramid", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the C2 / C5 legs (configs)")
    ap.add_argument("--no-two-view", action="store_true",
                    help="skip the informational two-views-in-flight block (tools/profile_round.sh: its co-resident kernels "
                         "would enter the per-kernel averages of the profile)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start it as `python3 bench.py --gpus N` (the script "
                         "starts its own ranks) or under torch.distributed.run with --nproc-per-node N")
    if os.environ.get("MSGS_BENCH_FAIL_RANK") == str(rank) and world > 1:
        raise SystemExit(3)        # rehearsal of a rank that dies before the rendezvous (tests/test_bench_launcher_cpu.py)
    if os.environ.get("MSGS_BENCH_LAUNCH_ONLY") == "1":
        # rehearsal of the rank start-up alone (tests/test_bench_launcher_cpu.py, no GPU): rendezvous over gloo, one
        # collective, rank 0 prints a JSON line the parent has to relay
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group(backend="gloo")
            t = torch.tensor([float(rank)])
            dist.all_reduce(t)
            dist.barrier()
            total = float(t.item())
            dist.destroy_process_group()
        else:
            total = 0.0
        if rank == 0:
            print(json.dumps({"launch_only": True, "n_gpus": world, "rank_sum": total}), flush=True)
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no fallback)")
    # one process per GPU; (local_rank % device_count only matters for the single-GPU rehearsal of the N > 1 path)
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" IS RCCL on ROCm.  MSGS_BENCH_BACKEND=gloo exists only to rehearse the multi-rank code path on a box
        # with fewer GPUs than ranks (RCCL refuses two ranks on one device).
        backend = os.environ.get("MSGS_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    import scenes
    import diff_gaussian_rasterization as dgr
    from gaussian_renderer import render
    from synthetic_model import SyntheticGaussians
    from view_parallel import PipelinedGradExchange
    from gaussian_renderer import PIPE

    W, H, P = args.width, args.height, args.gaussians
    if world == 1:      # C3
        settings = dict(filter_small=True, filter_large=True, fade_size=0.0)
        scene = scenes.frustum_scene(P, W, H, seed=2, sh_degree=3, multiscale=True)
        cams = [scenes.front_camera(W, H).to(dev)]
    else:               # C4: one shared world-space set, 8 ring cameras, view v -> rank v mod N
        scene, cams, settings = scenes.config_c4(n_views=8, P=P, width=W, height=H)
        cams = [c.to(dev) for c in cams]
    n_views = len(cams)
    cam = cams[0]
    pc = SyntheticGaussians(scene, dev, requires_grad=True)
    # N > 1: parameter .grad tensors are views into a flat fp32 bucket (two buckets, used alternately): the all-reduce
    #        of view k runs on RCCL's stream while view k+1 is rendered (PipelinedGradExchange); every exchange
    #        completes inside the timed region (drain before the closing synchronize).
    # N = 1: nothing to reduce -> grads are left to autograd (set-to-None each step, like optimizer.zero_grad).
    # (direct: the backward writes each view's gradients straight into the bucket — no zero-fill, no accumulation pass;
    #  available because the op chains the reference's getters itself)
    direct = bool(dgr.chain_reference_getters)
    from view_parallel import FactoredGradExchange
    exchange = FactoredGradExchange(pc, world) if world > 1 else None
    bg = torch.zeros(3, device=dev)
    dL = scenes.grad_seed(W, H, 2).to(dev)
    torch.cuda.synchronize()

    # HIP events recorded by the library on the stream it launches on, INSIDE the timed region: the two blend kernels (the
    # roofline's dominant kernel is one of them) on every 4th timed step.  An event record costs ~10 us of queue latency on this
    # runtime: all nine kernel classes on every 4th step slowed the timed region by 4 % (1.106 vs 1.066 ms per step, round 4),
    # so the other seven classes are timed on extra steps AFTER the timed region (per_kernel, below).
    TIMER_STRIDE = 4
    # at 1080p and above the backward blend is the dominant kernel by a wide margin (0.31 vs 0.18 ms at C3): it alone is timed
    # live there (two event records on every 4th step); smaller images time both blend kernels live
    LIVE = ("blend_bwd",) if W * H >= 1920 * 1080 else ("blend_fwd", "blend_bwd")
    timers = {} if args.no_kernel_timing else {k: dgr._C.KernelTimer(only=LIVE) for k in range(0, args.steps, TIMER_STRIDE)}

    step_no = [0]

    def next_cam():
        c = cams[(step_no[0] * world + rank) % n_views]
        step_no[0] += 1
        return c

    def step(timer=None):
        dgr._C.set_timer(timer)
        c = next_cam()
        if exchange is not None:
            exchange.begin_view(c.camera_center)
        else:
            for p_ in pc.parameters():
                p_.grad = None
        out = render(c, pc, PIPE, bg, **settings)
        out["render"].backward(dL)               # fixed dL/dimage (SURVEY §8(d) 'backward seed')
        if exchange is not None:                 # the exchange completes INSIDE the step: every .grad is final here
            exchange.end_view()
            exchange.finish()
        return out

    def timed_region(fn, n, drain=None):
        """barrier + synchronize on both sides, MAX over ranks (the driver's contract)"""
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        with quiet_gc():
            t = time.perf_counter()
            for k in range(n):
                fn(k)
            if drain is not None:
                drain()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
        if world > 1:
            tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt

    # N > 1: the factored exchange has only been rehearsed with gloo and with single-rank RCCL in this build's test
    # environment (one GPU per box).  If its first step fails on a real multi-GPU communicator, every rank falls back —
    # loudly, and the JSON line says so — to the dense flat all-reduce after each view, so that the scaling run still
    # yields a measurement of the C4 pattern.
    exchange_used = "factored" if exchange is not None else None
    if exchange is not None:
        # the ranks agree on the outcome over a SEPARATE CPU (gloo) group: a rank that failed between its two RCCL
        # collectives must not issue a third one on that communicator while its peers are still inside the second
        # (mismatched collectives hang); peers stuck in a collective a failed rank never joined are bounded by the
        # process-group timeout, not by this agreement
        agree = dist.new_group(backend="gloo")
        ok = torch.ones(1)
        try:
            step()
            torch.cuda.synchronize()
            if os.environ.get("MSGS_BENCH_FAIL_FACTORED") == str(rank):
                # rehearsal of the fallback (tests/test_bench_multirank_gpu.py): ONE rank fails its first factored step —
                # behind its collectives, so that its peer is not left inside one — and both ranks have to land on the
                # dense exchange through the agreement below
                raise RuntimeError("injected failure of the factored exchange (MSGS_BENCH_FAIL_FACTORED)")
        except Exception as e:           # noqa: BLE001 - any failure of the new path must not lose the measurement
            print(f"[bench rank {rank}] factored exchange failed ({e!r}); falling back to the dense all-reduce", file=sys.stderr)
            ok.zero_()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=agree)
        if ok.item() == 0:
            from view_parallel import FlatGradBucket
            dgr.set_grad_sinks(None)
            for p_ in pc.parameters():
                p_.grad = None
            fb_bucket = FlatGradBucket(pc.parameters())
            exchange, exchange_used = None, "dense_serial_allreduce (fallback)"

            def step(timer=None):        # noqa: F811 - replaces the factored step
                dgr._C.set_timer(timer)
                c = next_cam()
                if direct:
                    fb_bucket.detach_grads()
                    dgr.set_grad_sinks(fb_bucket.sinks())
                else:
                    fb_bucket.zero()
                out = render(c, pc, PIPE, bg, **settings)
                out["render"].backward(dL)
                fb_bucket.all_reduce(average_over=world)
                return out
    settle_gc()
    for _ in range(args.warmup):
        step()
    elapsed = timed_region(lambda k: step(timers.get(k)), args.steps)
    dgr._C.set_timer(None)

    ms_per_step = 1e3 * elapsed / args.steps
    value = world * (W * H / 1e6) / (elapsed / args.steps)
    # every kernel class on a few extra steps outside the timed region (roofline.per_kernel)
    all_timers = [] if args.no_kernel_timing else [dgr._C.KernelTimer() for _ in range(4)]
    for t_ in all_timers:
        step(t_)
    torch.cuda.synchronize()
    dgr._C.set_timer(None)

    # SURVEY 8(d) form of the same measurement (N = 1): median of >= 50 steps, each bracketed by HIP events on the compute
    # stream (the library launches on torch's current stream, so torch.cuda.Event sees its kernels)
    median50 = None
    if world == 1:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
        dgr._C.set_timer(None)
        for a_, b_ in evs:
            a_.record()
            step()
            b_.record()
        torch.cuda.synchronize()
        ts_ = sorted(a_.elapsed_time(b_) for a_, b_ in evs)
        med = 0.5 * (ts_[24] + ts_[25])
        median50 = {"ms_per_step": round(med, 4), "value": round(W * H / 1e6 / (med * 1e-3), 3), "unit": "Mpixels/s",
                    "min_ms": round(ts_[0], 4), "p90_ms": round(ts_[44], 4),
                    "how": "median of 50 steps, HIP events on the compute stream around each step"}

    # the same K steps with the host settings a drop-in train.py gets WITHOUT the two caller-side lines of INTEGRATION.md 1.1:
    # Python's collector on, autograd's backward on its device thread (the thread pools stay sized to the CPU quota: that cannot
    # be undone inside this process) — so that what the tuned settings are worth is visible next to `value`
    default_host = None
    if world == 1:
        try:
            if SINGLE_THREAD_BACKWARD:
                torch.autograd.set_multithreading_enabled(True)
            gc.enable()
            for _ in range(args.warmup):
                step()
            torch.cuda.synchronize()
            t_ = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            t_ = (time.perf_counter() - t_) / args.steps
            default_host = {"ms_per_step": round(1e3 * t_, 4), "value": round(W * H / 1e6 / t_, 3), "unit": "Mpixels/s",
                            "what": "the timed loop again with Python's GC enabled and torch's default multithreaded autograd "
                                    "(backward on the device thread)"}
        finally:
            if SINGLE_THREAD_BACKWARD:
                torch.autograd.set_multithreading_enabled(False)

    # N > 1, informational ("with and without the all-reduce"), the same K steps with
    #   (a) the dense exchange serialised after each view: ONE flat all-reduce of the 59 floats / Gaussian
    #   (b) that dense all-reduce overlapped with the rendering of the NEXT view (two buckets; an optimizer step then
    #       has to cover >= 2 views per GPU)
    #   (c) no exchange at all: pure view-sharded rendering fwd+bwd
    # `value` above INCLUDES the complete (factored) exchange in every step.
    extra = None
    if world > 1:
        from view_parallel import FlatGradBucket
        for p_ in pc.parameters():
            p_.grad = None
        bucket = FlatGradBucket(pc.parameters())

        def prepare(bk):
            if direct:
                bk.detach_grads()
                dgr.set_grad_sinks(bk.sinks())
            else:
                bk.zero()

        def step_serial(k):
            prepare(bucket)
            render(next_cam(), pc, PIPE, bg, **settings)["render"].backward(dL)
            bucket.all_reduce(average_over=world)

        def step_local(k):
            prepare(bucket)
            render(next_cam(), pc, PIPE, bg, **settings)["render"].backward(dL)
        step_serial(0)
        ts = timed_region(step_serial, args.steps) / args.steps
        step_local(0)
        tl = timed_region(step_local, args.steps) / args.steps
        dgr.set_grad_sinks(None)
        del bucket
        pipe = PipelinedGradExchange(pc.parameters(), world, direct=direct)

        def step_pipe(k):
            pipe.begin_view()
            render(next_cam(), pc, PIPE, bg, **settings)["render"].backward(dL)
            pipe.end_view()
        step_pipe(0)
        pipe.drain()
        tp = timed_region(step_pipe, args.steps, drain=pipe.drain) / args.steps
        dgr.set_grad_sinks(None)
        # an optimizer step that covers ALL 8 views of C4: every rank renders its 8 / N views with two in flight
        # (multi_view.ViewPipeline, gradients summed in the per-Gaussian backward straight into the flat bucket) and the dense
        # bucket crosses the ranks once per step (view_parallel.MultiViewStepExchange)
        two_views = None
        if n_views % world == 0 and n_views // world >= 2:
            from multi_view import ViewPipeline
            from view_parallel import MultiViewStepExchange
            for p_ in pc.parameters():
                p_.grad = None
            mine = [cams[v] for v in range(n_views) if v % world == rank]
            mv = MultiViewStepExchange(pc, n_views)
            vp_ = ViewPipeline(dev, n_streams=2)
            bwd_ = lambda i, pkg: pkg["render"].backward(dL)

            def step_multi(k):
                mv.step(vp_, mine, PIPE, bg, bwd_, **settings)
            # Informational: a failure here must not cost the scaling line — but mv.step ends in an all-reduce, and a failure
            # on ONE rank (out of memory, a HIP error on one GPU) must not leave its peers inside that collective.  So every
            # rank first runs its views WITHOUT the collective (the rank-local part: pipeline, accumulator, bucket), the ranks
            # agree over the CPU group that all of them got through, and only then the collective version runs — outside any
            # try: a failure there ends this rank, and the launcher (or torch.distributed.run) stops its peers.
            ok2 = torch.ones(1)
            err2 = None
            try:
                mv.bucket.detach_grads()
                vp_.train_views(mine, pc, PIPE, bg, bwd_, accumulator=mv.acc, **settings)
                torch.cuda.synchronize()
            except Exception as e:      # noqa: BLE001
                print(f"[bench rank {rank}] two_views_per_rank failed locally: {e!r}", file=sys.stderr)
                err2 = repr(e)
                ok2.zero_()
                dgr.set_grad_accumulator(None)
            dist.all_reduce(ok2, op=dist.ReduceOp.MIN, group=agree)
            if ok2.item() == 0:
                two_views = {"error": err2 or "another rank failed in its rank-local dry run; block skipped on every rank"}
            else:
                step_multi(0)
                tm_ = timed_region(step_multi, args.steps) / args.steps
                two_views = {"views_per_rank_per_step": len(mine), "ms_per_optimizer_step": round(1e3 * tm_, 4),
                             "fwd_bwd_ms_per_view": round(1e3 * tm_ / len(mine), 4),
                             "value": round(n_views * (W * H / 1e6) / tm_, 3), "unit": "Mpixels/s",
                             "what": "one optimizer step over all 8 C4 views: each rank renders its 8 / N views through the "
                                     "two-lane pipeline into one flat bucket, ONE dense all-reduce per step"}
            for p_ in pc.parameters():
                p_.grad = None
            del mv, vp_
        mp = world * (W * H / 1e6)
        dense_bytes = 4 * sum(p_.numel() for p_ in pc.parameters())
        extra = {"headline_exchange": exchange_used,
                 "dense_serial_allreduce": {"ms_per_step": round(1e3 * ts, 4), "value": round(mp / ts, 3)},
                 "dense_pipelined": {"ms_per_step": round(1e3 * tp, 4), "value": round(mp / tp, 3),
                                     "note": "all-reduce of view k overlapped with view k+1: needs >= 2 views per GPU "
                                             "per optimizer step"},
                 "without_exchange": {"ms_per_step": round(1e3 * tl, 4), "value": round(mp / tl, 3)},
                 "unit": "Mpixels/s", "dense_allreduce_bytes": dense_bytes,
                 "factored_bytes_received_per_gpu": exchange.bytes_per_step() if exchange is not None else None}

    result = {
        "metric": "Mpixels/s fwd+bwd @1080p, 1M Gaussians; fraction of HBM roofline",
        "value": round(value, 3), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "timing_note": "Python's cyclic GC is collected before the warm-up of and paused during every timed region (a full collection "
                       f"of this process takes 30-70 ms); OMP/MKL/torch intra-op pools = {HOST_THREADS} threads (the container's CPU "
                       f"quota is {usable_cpus()} of {os.cpu_count()} visible CPUs); autograd backward "
                       + ("on the calling thread (torch.autograd.set_multithreading_enabled(False))" if SINGLE_THREAD_BACKWARD
                          else "on autograd's device thread (torch default)"),
        "config": {"workload": (
            ("C3 (BASELINE configs[2]): 1M Gaussians, 1920x1080, SH3, multi-scale fields, filter_small+filter_large, "
             "fade 0; frozen seeded scene scenes.config('C3') [SCALE_K 0.004: D/P = 9.6 instances per Gaussian by the "
             "reference's 3-sigma-rect duplication (SURVEY 8(d) asks 8-12), 4.1 after this build's exact culling, "
             "V/P = 0.56 rendered]" if world == 1 else
             "C4 (BASELINE configs[3]): ONE shared 1M-Gaussian ball (scenes.config_c4, seed 4), 8 ring cameras, "
             "1920x1080, SH3; step k: rank r renders view (k*N + r) mod 8; per-Gaussian gradients exchanged over RCCL "
             "inside every step")
            if (P, W, H) == (1_000_000, 1920, 1080) else f"custom: {P} Gaussians {W}x{H}"),
                   "gaussians": P, "width": W, "height": H, "views_per_gpu_per_step": 1,
                   "api": "reference API: gaussian_renderer.render() -> GaussianRasterizer.forward(13 kwargs); the op "
                          "recognises the reference's getters in the autograd graph and chains their backward inside "
                          "msgs_backward (chain_reference_getters=" + str(bool(dgr.chain_reference_getters)) + ")",
                   "parallelism": f"view-parallel x{world}" + (
                       ": replicated parameters, one view per GPU per step, factored gradient exchange (all-gather of "
                       "the [P,3] dL/drgb factors + all-reduce of the 11 non-SH floats; SH gradient rows rebuilt on "
                       "every rank), complete inside the step" if world > 1 else "")},
    }

    if world == 1 and (P, W, H) == (1_000_000, 1920, 1080):
        # what an N-rank run of the same line WOULD exchange per GPU and step (formula of view_parallel.FactoredGradExchange.bytes_per_step
        # / the dense 59-float ring all-reduce), so that the driver's N = 1 record shows it; nothing is exchanged at N = 1
        def _factored(n):       # all-gather of {dL/drgb [P,3] | camera centre + pad (4)} + ring all-reduce of the 11 non-SH floats
            return 4 * ((n - 1) * (3 * P + 4) + 2 * (n - 1) * (11 * P) // n)

        def _dense(n):          # ring all-reduce of all 59 gradient floats per Gaussian
            return 4 * (2 * (n - 1) * (59 * P) // n)
        result["config"]["exchange_if_view_parallel"] = {
            "rccl_ranks": None, "backend": None,
            "bytes_received_per_gpu_per_step": {str(n): {"factored": _factored(n), "dense_allreduce": _dense(n)} for n in (2, 4, 8)},
            "what": "N = 1 runs no exchange.  At N ranks (python -m torch.distributed.run ... bench.py --gpus N) every step ends "
                    "with view_parallel.FactoredGradExchange over RCCL: all-gather of the [P,3] dL/drgb factors, all-reduce "
                    "(ncclAvg) of the 11 non-SH floats per Gaussian, SH rows rebuilt on every rank; the line then carries "
                    "config.exchange, config.backend and config.rccl_ranks"}
    if world > 1:
        result["config"]["exchange"] = exchange_used
        # which communicator carried the exchange: "nccl" IS RCCL on ROCm; rccl_ranks = its world size (None on the gloo rehearsal)
        result["config"]["backend"] = dist.get_backend()
        result["config"]["rccl_ranks"] = dist.get_world_size() if dist.get_backend() == "nccl" else None
    if rank == 0:
        # ---- roofline of the dominant kernel (per launch, averaged over the timed steps) ----
        stats = None
        try:
            import ctypes as C
            # rebuild the forward state once to read D_trav / V (outside the timed region)
            call_out = render(cam, pc, PIPE, bg, **settings)
            fn = call_out["render"].grad_fn
            ctx = fn
            geom, binning, image, D = ctx.state
            scratch = torch.empty(256, dtype=torch.uint8, device=dev)
            o = (C.c_int64 * 2)()
            lib = dgr._C.lib
            dgr._C.check(lib.msgs_binning_stats(C.byref(ctx.call.view), P, C.c_void_p(ctx.radii.data_ptr()),
                                                C.c_void_p(binning.data_ptr()), binning.numel(),
                                                C.c_void_p(image.data_ptr()), image.numel(),
                                                C.c_void_p(scratch.data_ptr()), scratch.numel(), o,
                                                C.c_void_p(torch.cuda.current_stream().cuda_stream)), "stats")
            stats = {"D": int(D), "D_trav": int(o[0]), "V": int(o[1])}
            # lane efficiency of the blend forward, measured by a counting replica of the kernel on this state
            o3 = (C.c_int64 * 7)()
            dgr._C.check(lib.msgs_blend_lane_stats(C.byref(ctx.call.view), C.c_void_p(geom.data_ptr()), geom.numel(), P, int(D),
                                                   C.c_void_p(binning.data_ptr()), binning.numel(),
                                                   C.c_void_p(image.data_ptr()), image.numel(),
                                                   C.c_void_p(scratch.data_ptr()), scratch.numel(), o3,
                                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)), "lane stats")
            stats["blend_fwd_lanes"] = {"wave_entry_evaluations": int(o3[0]), "evaluated_lanes": 64 * int(o3[0]),
                                        "lanes_still_blending": int(o3[1]), "lanes_blended": int(o3[2]),
                                        "lane_efficiency": round(int(o3[2]) / max(64 * int(o3[0]), 1), 4)}
            if o3[3] >= 0:      # counting replica of the one-wave-per-tile backward
                stats["blend_bwd_lanes"] = {"tile_entry_visits": int(o3[3]), "quadrant_entry_evaluations": int(o3[4]),
                                            "evaluated_lanes": 64 * int(o3[4]), "lanes_contributing": int(o3[5]),
                                            "visits_with_a_contribution": int(o3[6]),
                                            "lane_efficiency": round(int(o3[5]) / max(64 * int(o3[4]), 1), 4)}
        except Exception as e:  # statistics are informative; never fail the bench line on them
            stats = {"error": repr(e)}
        kernels = None
        if timers:
            acc = {}
            for t in all_timers:                          # all nine classes, steps after the timed region
                for k, v in t.read_ms().items():
                    if v >= 0:
                        acc.setdefault(k, []).append(v)
            live = {}
            for t in timers.values():                     # the blend pair, live inside the timed region: these win
                for k, v in t.read_ms().items():
                    if v >= 0:
                        live.setdefault(k, []).append(v)
            acc.update(live)
            kernels = {k: round(float(np.mean(v)), 4) for k, v in acc.items()}
        roof = None
        if kernels and stats and "D_trav" in stats:
            N = W * H
            tiles = ((W + 15) // 16) * ((H + 15) // 16)
            alg = {"blend_fwd": 48 * stats["D_trav"] + 28 * N + 8 * tiles,
                   "blend_bwd": 48 * stats["D_trav"] + 20 * N + 36 * stats["V"]}
            dom = max(alg, key=lambda k: kernels.get(k, 0.0))
            achieved = alg[dom] / (kernels[dom] * 1e-3) / 1e9
            both = (alg["blend_fwd"] + alg["blend_bwd"]) / ((kernels["blend_fwd"] + kernels["blend_bwd"]) * 1e-3) / 1e9
            # HBM bytes per launch from rocprofv3 PMC passes of this same command (FETCH_SIZE / WRITE_SIZE in separate
            # passes, corrected per MI355X_MICROARCH.md §HBM), collected offline by tools/profile_round.sh (counters cannot
            # be read from inside the timed process) and committed under profiles/ together with the sha256 of the kernel
            # sources they were measured on: reported only while that hash still matches this tree, null otherwise.
            name_of = {"blend_fwd": "blend_forward_kernel", "blend_bwd": "blend_backward_tile"}[dom]
            here = csrc_sha256()

            def committed(prefix):
                """the newest profiles/<prefix>_rNN.json whose kernel-source hash matches this tree"""
                import glob
                for fname in sorted(glob.glob(os.path.join(ROOT, "profiles", prefix + "_r*.json")), reverse=True):
                    try:
                        j_ = json.load(open(fname))
                        if j_.get("csrc_sha256") != here or (P, W, H) != (1_000_000, 1920, 1080) or world != 1:
                            continue
                        hits = [v for k, v in j_["kernels"].items() if k.startswith(name_of)]
                        if hits:
                            committed.source = os.path.relpath(fname, ROOT)
                            return max(hits, key=lambda v: v.get("launches", 1))
                    except Exception:
                        continue
                return None
            committed.source = None
            tr = committed("traffic")
            traffic = int(tr["hbm_bytes"]) if tr else None
            committed.traffic_file = committed.source if tr else None
            # what actually bounds the kernel: INSTRUCTION ISSUE.  One model for both blend kernels: per
            # launch, wave-instructions by class from the committed PMC summary (same hash rule) x the calibrated issue cost
            # of each class (tools/valu_calib.hip -> profiles/r2_valu_calibration.txt: a SIMD issues about one instruction of
            # ANY kind per ~2 cycles; plain fp32 VALU 2.2, v_cmp / v_cndmask / DPP 4.25, v_exp / v_rcp 8.1, scalar 2.0,
            # LDS 4.0 per instruction per SIMD at the kernels' access widths) / (1024 SIMDs x 2.4 GHz).  The VALU mix per
            # kernel is counted in the ISA of the inner loops (fractions of plain / half-rate / transcendental).
            valu = None
            sq = committed("sq")
            sq_source = committed.source
            # instruction mix of the hot loops as the compiler emitted them (tools/isa_mix.py at build time -> build/isa_mix.json;
            # the copy committed under profiles/ is the fallback when the build directory did not travel)
            isa, isa_source = None, None
            for cand_ in [os.path.join(ROOT, "ms-gs_amd", "build", "isa_mix.json")] + sorted(
                    __import__("glob").glob(os.path.join(ROOT, "profiles", "r*_isa_mix.json")), reverse=True):
                try:
                    isa = json.load(open(cand_))
                    isa_source = os.path.relpath(cand_, ROOT)
                    break
                except Exception:
                    continue
            cyc = (isa or {}).get("cycles_per_class", {"plain": 2.2, "half": 4.25, "trans": 8.1, "salu": 2.0, "lds": 4.0, "vmem": 2.0})
            SIMD_HZ = 1024 * 2.4e9

            def cyc_of(c):
                return sum(float(c.get(k, 0.0)) * cyc[k] for k in cyc)
            if sq and "SQ_INSTS_VALU" in sq:
                if isa and dom == "blend_fwd":
                    mix = tuple(isa["blend_fwd"]["valu_mix"])
                elif isa and dom == "blend_bwd" and stats and "blend_bwd_lanes" in stats and isa["blend_bwd"].get("per_reduction"):
                    # dynamic VALU mix of the backward: visits x entry overhead + quadrant evaluations x quadrant step +
                    # reductions x reduction, with the counting replica's numbers of this very scene
                    bl, ib = stats["blend_bwd_lanes"], isa["blend_bwd"]
                    tot = {k: bl["tile_entry_visits"] * ib["per_entry_visit"][k] + bl["quadrant_entry_evaluations"] * ib["per_quadrant_step"][k]
                           + bl["visits_with_a_contribution"] * ib["per_reduction"][k] for k in ("plain", "half", "trans")}
                    nv_ = sum(tot.values())
                    mix = (tot["plain"] / nv_, tot["half"] / nv_, tot["trans"] / nv_)
                else:
                    mix = {"blend_fwd": (18.5 / 23.5, 4.0 / 23.5, 1.0 / 23.5), "blend_bwd": (0.64, 0.31, 0.05)}[dom]
                nv, ns = float(sq["SQ_INSTS_VALU"]), float(sq.get("SQ_INSTS_SALU", 0.0))
                nl = float(sq.get("SQ_INSTS_LDS", 0.0))
                cyc_v = mix[0] * cyc["plain"] + mix[1] * cyc["half"] + mix[2] * cyc["trans"]
                cycles = nv * cyc_v + ns * cyc["salu"] + nl * cyc["lds"]
                floor_ms = cycles / SIMD_HZ * 1e3
                valu = {"valu_wave_instructions": nv, "salu_wave_instructions": ns, "lds_wave_instructions": nl,
                        "valu_mix_plain_half_trans": [round(m_, 4) for m_ in mix],
                        "cycles_per_valu_instruction_of_this_mix": round(cyc_v, 3),
                        "issue_floor_ms": round(floor_ms, 4), "frac_of_issue_floor": round(floor_ms / kernels[dom], 4),
                        "valu_only_floor_ms": round(nv * cyc_v / SIMD_HZ * 1e3, 4),
                        "model": "sum over classes of wave-instructions x calibrated cycles per instruction per SIMD "
                                 "(VALU mix, scalar 2.0, LDS 4.0) / (1024 SIMDs x 2.4 GHz); NOT an efficiency — it prices the "
                                 "kernel's OWN instruction count, so any issue-bound kernel scores ~1: see useful_issue",
                        "source": f"{sq_source} + profiles/r2_valu_calibration.txt + {isa_source or 'hand-counted mix (no isa_mix.json)'}"}
            # USEFUL issue: the time the vector units would need for the pairs that actually contribute, at the per-pair
            # instruction cost of this very ISA with every lane useful and no per-entry overhead — what the hardware could do
            # for this work — over the measured kernel time.  forward: pairs that blended x VALU of one (wave, entry) step / 64;
            # backward: pairs that contributed a gradient x VALU of one quadrant step / 64.
            useful = None
            if isa and stats and "blend_fwd_lanes" in stats and kernels.get("blend_fwd") and kernels.get("blend_bwd"):
                f_ms = stats["blend_fwd_lanes"]["lanes_blended"] / 64.0 * isa["blend_fwd"]["valu_cycles_per_wave_entry"] / SIMD_HZ * 1e3
                useful = {"blend_fwd": {"contributing_pairs": stats["blend_fwd_lanes"]["lanes_blended"],
                                        "valu_per_64_pairs": round(isa["blend_fwd"]["valu_per_wave_entry"], 2),
                                        "useful_ms": round(f_ms, 4), "kernel_ms": kernels["blend_fwd"],
                                        "frac": round(f_ms / kernels["blend_fwd"], 4)}}
                pair_ms = f_ms
                if "blend_bwd_lanes" in stats:
                    b_ms = stats["blend_bwd_lanes"]["lanes_contributing"] / 64.0 * isa["blend_bwd"]["valu_cycles_per_quadrant_step"] / SIMD_HZ * 1e3
                    useful["blend_bwd"] = {"contributing_pairs": stats["blend_bwd_lanes"]["lanes_contributing"],
                                           "valu_per_64_pairs": round(isa["blend_bwd"]["valu_per_quadrant_step"], 2),
                                           "useful_ms": round(b_ms, 4), "kernel_ms": kernels["blend_bwd"],
                                           "frac": round(b_ms / kernels["blend_bwd"], 4)}
                    pair_ms += b_ms
                    # the same ISA counts x the replica's counts reproduce the kernel: a check of the model, not a new number
                    bl, ib = stats["blend_bwd_lanes"], isa["blend_bwd"]
                    if ib.get("per_reduction"):
                        model_ms = (bl["tile_entry_visits"] * cyc_of(ib["per_entry_visit"]) + bl["quadrant_entry_evaluations"] * cyc_of(ib["per_quadrant_step"])
                                    + bl["visits_with_a_contribution"] * cyc_of(ib["per_reduction"])) / SIMD_HZ * 1e3
                        useful["blend_bwd"]["isa_x_replica_model_ms"] = round(model_ms, 4)
                    fl = stats["blend_fwd_lanes"]
                    useful["blend_fwd"]["isa_x_replica_model_ms"] = round(fl["wave_entry_evaluations"] * isa["blend_fwd"]["cycles_per_wave_entry"] / SIMD_HZ * 1e3, 4)
                useful["blend_fwd_plus_bwd"] = {"useful_ms": round(pair_ms, 4), "kernel_ms": round(kernels["blend_fwd"] + kernels["blend_bwd"], 4),
                                                "frac": round(pair_ms / (kernels["blend_fwd"] + kernels["blend_bwd"]), 4)}
                useful["what"] = ("contributing (pixel, Gaussian) pairs x VALU instructions of one per-pair step in the emitted ISA / 64 lanes "
                                  "x calibrated cycles / (1024 SIMDs x 2.4 GHz), over the HIP-event kernel time: the share of the kernel's "
                                  "time that is arithmetic on pairs that count; the rest is idle lanes, per-entry overhead (fetch, masks, "
                                  "loop control) and, in the backward, the cross-lane reduction and the atomics")
                useful["source"] = isa_source
            # every SURVEY 8(d) row against HBM, from the same HIP-event kernel times: the HBM-bound kernels read against
            # HBM, the issue-bound blend pair against both
            Dn, Vn = stats["D"], stats["V"]
            alg_all = algorithmic_bytes(P, W, H, Dn, stats["D_trav"], Vn)
            kms = dict(kernels)
            kms["sort(depth+tile)"] = kernels.get("depth_sort", 0.0) + kernels.get("tile_sort", 0.0)
            per_kernel = {k: {"algorithmic_bytes": int(b), "ms": round(kms[k], 4),
                              "GBps": round(b / (kms[k] * 1e-3) / 1e9, 1),
                              "frac": round(b / (kms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                          for k, b in alg_all.items() if kms.get(k, 0.0) > 0}
            # measured HBM bytes per step and class from the same committed PMC summary (mean per launch x launches per step;
            # every launch of that profile is a C3 launch: tools/profile_round.sh runs without the pyramid and two-view legs)
            try:
                tj = json.load(open(os.path.join(ROOT, committed.traffic_file))) if committed.traffic_file else None
            except Exception:
                tj = None
            if tj:
                classes = {"preprocess": ("preprocess_kernel",), "scan": ("scan_",), "emit": ("emit_kernel",),
                           "sort(depth+tile)": ("radix_", "group_scan_kernel"), "ranges": ("ranges_kernel",),
                           "blend_fwd": ("blend_forward_kernel",), "blend_bwd": ("blend_backward_tile",),
                           "preprocess_bwd": ("preprocess_backward_kernel",)}
                steps_ = max([v["launches"] for k, v in tj["kernels"].items() if k.startswith("blend_backward_tile")] or [0])
                for cls, prefixes in classes.items():
                    if cls not in per_kernel or not steps_:
                        continue
                    tot = sum(v["hbm_bytes"] * v["launches"] for k, v in tj["kernels"].items() if k.startswith(prefixes))
                    if tot > 0:
                        per_kernel[cls]["traffic"] = int(tot / steps_)
                        per_kernel[cls]["traffic_over_algorithmic"] = round(tot / steps_ / alg_all[cls], 2)
                        per_kernel[cls]["traffic_GBps"] = round(tot / steps_ / (kms[cls] * 1e-3) / 1e9, 1)
            total_alg = float(sum(alg_all.values()))
            whole = {"algorithmic_bytes": int(total_alg), "ms_per_step": round(ms_per_step, 4),
                     "GBps": round(total_alg / (ms_per_step * 1e-3) / 1e9, 1),
                     "frac": round(total_alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "instances": Dn,
                     "note": "sum of the SURVEY 8(d) per-kernel algorithmic bytes with THIS build's instance count D (exact "
                             "culling) / the driver-timed step / 8 TB/s; K4 priced as the reference's 64-bit-key sort"}
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "algorithmic_bytes": alg[dom], "avg_kernel_ms": kernels[dom], "valu_issue": valu, "useful_issue": useful,
                    "blend_fwd_plus_bwd": {"achieved": round(both, 2), "frac": round(both / HBM_PEAK_GBS, 5),
                                           "algorithmic_bytes": alg["blend_fwd"] + alg["blend_bwd"],
                                           "ms": round(kernels["blend_fwd"] + kernels["blend_bwd"], 4)},
                    "whole_step": whole, "per_kernel": per_kernel}
        result["roofline"] = roof
        result["median_of_50"] = median50
        result["default_host_settings"] = default_host
        result["kernel_ms"] = kernels
        result["kernel_timing"] = (None if not timers else
                                   f"HIP events recorded by the library: {' / '.join(LIVE)} live on {len(timers)} of the "
                                   f"{args.steps} timed steps, the other classes on {len(all_timers)} steps after the timed region")
        if extra is not None:
            result["exchange"] = extra
            result["two_views_per_rank"] = two_views
        # the reference's own measurement: render time per resolution scale (train.py:488-496,541; viewer.py:67-81)
        if world == 1 and (P, W, H) == (1_000_000, 1920, 1080) and not args.no_pyramid:
            try:
                result["pyramid_ms"], result["render_forward_ms"] = pyramid_timing(scenes, pc, settings, bg, dev)
            except Exception as e:
                result["pyramid_ms"] = result["render_forward_ms"] = {"error": repr(e)}
        # informational: the opt-in raw-parameter entry (render_fused: activations + SH concat inside K1/K9,
        # SURVEY §8(f) rank 1) on the same workload.  `value` above stays on the reference-API drop-in path.
        if world == 1:
            try:
                from gaussian_renderer import render_fused

                def fused_step():
                    for p_ in pc.parameters():
                        p_.grad = None
                    render_fused(cam, pc, PIPE, bg, **settings)["render"].backward(dL)
                tf = 1e-3 * period_median(fused_step, max(args.steps, 20), args.warmup, torch.cuda.synchronize)[0]
                result["fused_path"] = {"ms_per_step": round(1e3 * tf, 4), "value": round(W * H / 1e6 / tf, 3),
                                        "unit": "Mpixels/s", "entry": "GaussianRasterizer.forward_raw",
                                        "how": "median period of >= 20 steps"}
            except Exception as e:
                result["fused_path"] = {"error": repr(e)}
            # informational: the same workload with TWO views in flight (host/multi_view.py; review item 1).  Eight copies of the
            # C3 view per sweep, so a view is the same work as a headline step; `value` above stays one view per step.
            try:
                result["two_view_pipeline"] = None if args.no_two_view else two_view_timing(
                    pc, cam, bg, dL, settings, W, H, args.warmup, (result.get("roofline") or {}).get("whole_step"))
            except Exception as e:
                result["two_view_pipeline"] = {"error": repr(e)}
            # informational: one whole training iteration (train.py:202-218,239-250,416-418) on the same workload —
            # render_fused + fused L1/SSIM loss + backward + statistics + FusedAdam, vs the same iteration with the
            # reference's torch composition around this rasterizer (torch loss formulation, torch.optim.Adam, masked-
            # index statistics).  Separate copies of the model; neither is part of `value`.
            try:
                result["train_iteration"] = train_iteration_timing(scenes, scene, cam, bg, settings, W, H, dev,
                                                                   args.steps, args.warmup, multi_view=not args.no_two_view)
            except Exception as e:
                result["train_iteration"] = {"error": repr(e)}
        result["binning"] = stats
        # the other single-GPU BASELINE configs, driver-visible (review item 4): C2 and C5 through the same call surface
        if world == 1 and (P, W, H) == (1_000_000, 1920, 1080) and not args.no_configs:
            result["configs"] = {}
            pc = dL = call_out = None           # (the closures above keep the names alive: drop the tensors, not the names)
            gc.collect()
            torch.cuda.empty_cache()
            try:
                result["reference_schedule"] = reference_schedule_leg(scenes, dev)
            except Exception as e:          # informational
                result["reference_schedule"] = {"error": repr(e)}
            try:
                result["verification_mode"] = verification_mode_leg(scenes, dev)
            except Exception as e:          # informational
                result["verification_mode"] = {"error": repr(e)}
            for name in ("C2", "C5"):
                try:
                    # (C2 is host-bound and a step takes 0.4 ms: 100 steps for a steadier median, still 50 ms)
                    result["configs"][name] = config_leg(name, scenes, dev, steps=100 if name == "C2" else 20,
                                                         warmup=10 if name == "C2" else 3)
                except Exception as e:      # informational
                    result["configs"][name] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline(scenes, scene, settings, W, H)
                # the same whole-step figure with the REFERENCE's duplication count (3-sigma rects, counted by the oracle)
                d_ref = result["cpu_baseline"].get("instances_reference_duplication")
                ws = (result.get("roofline") or {}).get("whole_step")
                if d_ref and ws and stats and "D" in stats:
                    tiles_ = ((W + 15) // 16) * ((H + 15) // 16)
                    per_inst = 12 + (24 * math.ceil((32 + max(1, math.ceil(math.log2(tiles_)))) / 8) + 8) + 8
                    b_ref = ws["algorithmic_bytes"] + per_inst * (d_ref - stats["D"])
                    ws["with_reference_duplication"] = {
                        "instances": int(d_ref), "algorithmic_bytes": int(b_ref),
                        "frac": round(b_ref / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            except Exception as e:
                result["cpu_baseline"] = {"error": repr(e)}
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
