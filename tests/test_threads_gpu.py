"""The library from several host threads at once (include/msgs.h, threading note: re-entrant; msgs_forward keeps its status words per
calling thread, msgs_forward_launch per handle).  Four Python threads, each with its own HIP stream, model copy and camera, render
forward + backward 25 times concurrently — a trainer thread next to viewer / evaluation threads; torch's autograd runs each backward on
its device thread, so forwards and backwards of different views interleave freely.  Every result must equal the one the same view
gives alone: outputs bit for bit, gradients to the last bits of the float64 atomics (1e-6 of the tensor's max norm)."""
import threading

import pytest
import torch

import scenes
from parity_utils import PIPE

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]
LEAVES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
ST = dict(filter_small=False, filter_large=False, fade_size=1.0)


def test_four_threads_render_and_backpropagate_concurrently():
    from gaussian_renderer import render, render_fused
    from synthetic_model import SyntheticGaussians
    from multi_view import ViewPipeline
    n_threads, reps = 4, 25
    W, H = 320, 200
    sc = scenes.ball_scene(40000, seed=45, log_s=-3.0)
    dev = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
    cams = [scenes.ring_camera(v, n_threads, W, H).to(dev) for v in range(n_threads)]
    dLs = [scenes.grad_seed(W, H, 70 + v).to(dev) for v in range(n_threads)]
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    fns = [render, render_fused, render, render_fused]

    def one(v, model):
        for p in model.parameters():
            p.grad = None
        out = fns[v](cams[v], model, PIPE, bg, **ST)
        out["render"].backward(dLs[v])
        return out

    refs = []
    for v in range(n_threads):
        m = SyntheticGaussians(sc, dev)
        o = one(v, m)
        torch.cuda.synchronize()
        refs.append(({k: o[k].clone() for k in ("render", "depth", "radii")}, [p.grad.clone() for p in m.parameters()]))

    errors = []
    start = threading.Barrier(n_threads)

    def worker(v):
        try:
            torch.cuda.set_device(dev)
            stream = torch.cuda.Stream(dev)
            model = SyntheticGaussians(sc, dev)
            torch.cuda.synchronize()
            start.wait()
            with torch.cuda.stream(stream):
                for r in range(reps):
                    if v == 3 and r % 5 == 4:
                        # this thread also runs a two-lane sweep now and then (launch / finish handles next to the per-thread blocks)
                        with torch.no_grad():
                            ViewPipeline(dev).render_views([cams[v]] * 3, model, PIPE, bg, **ST)
                    o = one(v, model)
                    stream.synchronize()
                    want, want_g = refs[v]
                    for k in want:
                        if not torch.equal(o[k], want[k]):
                            raise AssertionError(f"thread {v} rep {r}: {k} differs")
                    for n, p, g in zip(LEAVES, model.parameters(), want_g):
                        d = (p.grad - g).abs().max().item()
                        if d > 1e-6 * g.abs().max().item():
                            raise AssertionError(f"thread {v} rep {r}: grad {n} off by {d:.3e}")
        except BaseException as e:                 # noqa: BLE001 - reported by the main thread
            errors.append(e)
            try:
                start.abort()
            except Exception:
                pass

    # the main thread plays a view-parallel trainer that has its SH-factor sink registered (view_parallel.FactoredGradExchange):
    # the registry is per thread, the workers' backwards must keep forming their own SH gradient rows
    import diff_gaussian_rasterization as dgr
    dgr.set_grad_sinks({}, sh_factor=torch.empty(sc.P, 3, device=dev))
    try:
        threads = [threading.Thread(target=worker, args=(v,)) for v in range(n_threads)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=300)
    finally:
        dgr.set_grad_sinks(None)
    assert not any(t.is_alive() for t in threads), "a worker thread hangs"
    assert not errors, errors[0]
