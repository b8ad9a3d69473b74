"""Run as a child process with MSGS_BLOCKING_SYNC=1 (the switch is latched on first use, so it needs its own process; started by
tests/conftest.py through tests/_rehearsals.py, collected by tests/test_blocking_sync_gpu.py).

With the blocking fallback msgs_forward_finish copies the instance count from DEVICE words that live in the launch's stage-1
scratch — at resolve time, behind whatever the stream was given since.  Several forwards launched on ONE stream before any of them
is resolved (ViewPipeline(n_streams=1), or any deferred_forward user) therefore need that scratch to outlive the launch: the
Python layer keeps it referenced until resolve().  Everything here must equal the serial loop bit for bit."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

assert os.environ.get("MSGS_BLOCKING_SYNC") == "1"

import torch  # noqa: E402

import scenes  # noqa: E402
from parity_utils import PIPE  # noqa: E402

LEAVES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
ST = dict(filter_small=False, filter_large=False, fade_size=1.0)
OUT_KEYS = ("render", "acc_pixel_size", "depth", "radii", "pixel_sizes")


def main():
    import diff_gaussian_rasterization as dgr
    from gaussian_renderer import render
    from multi_view import ViewPipeline
    from synthetic_model import SyntheticGaussians
    n_views, W, H = 6, 320, 200
    # views of very different instance counts (ring cameras + two zoom levels): a stale or overwritten count shows
    sc = scenes.ball_scene(60000, seed=44, log_s=-3.0)
    cams = [scenes.ring_camera(v, n_views, W if v % 2 == 0 else W // 2, H if v % 2 == 0 else H // 2).to("cuda")
            for v in range(n_views)]
    dLs = [scenes.grad_seed(c.image_width, c.image_height, 90 + v).cuda() for v, c in enumerate(cams)]
    bg = torch.tensor([0.1, 0.2, 0.3], device="cuda")

    ref_pc = SyntheticGaussians(sc, "cuda")
    ref, ref_D = [], []
    for cam, dL in zip(cams, dLs):
        o = render(cam, ref_pc, PIPE, bg, **ST)
        o["render"].backward(dL)
        ref.append(o)
        ref_D.append(o["render"].grad_fn.state[3])
    torch.cuda.synchronize()
    assert len(set(ref_D)) > 1, ref_D

    for round_ in range(3):            # first round: no guess for the half-size shape; later rounds: speculative stage 2
        # (a) several forwards on ONE stream before any is resolved
        pc = SyntheticGaussians(sc, "cuda")
        with torch.no_grad():
            with dgr.deferred_forward() as pending:
                outs = [render(c, pc, PIPE, bg, **ST) for c in cams]
                assert all(p_.state is None for p_ in pending)
                got_D = [p_.resolve()[3] for p_ in pending]
        torch.cuda.synchronize()
        assert got_D == ref_D, (round_, got_D, ref_D)
        for o, r in zip(outs, ref):
            for k in OUT_KEYS:
                assert torch.equal(o[k], r[k]), (round_, k)
        # (b) the one-lane pipeline, forward + backward
        pc = SyntheticGaussians(sc, "cuda")
        kept = []

        def bwd(i, pkg):
            pkg["render"].backward(dLs[i])
            kept.append(pkg)
        ViewPipeline("cuda", n_streams=1).train_views(cams, pc, PIPE, bg, bwd, **ST)
        torch.cuda.synchronize()
        for o, r, D in zip(kept, ref, ref_D):
            assert o["render"].grad_fn.state.resolve()[3] == D
            for k in OUT_KEYS:
                assert torch.equal(o[k], r[k]), (round_, k)
        for n in LEAVES:
            assert torch.equal(getattr(pc, n).grad, getattr(ref_pc, n).grad), (round_, n)
    print("BLOCKING_SYNC_OK", ref_D)


if __name__ == "__main__":
    main()
