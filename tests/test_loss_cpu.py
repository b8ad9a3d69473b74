"""oracle/loss_oracle.py against golden vectors produced by the reference's own utils/loss_utils.py
(tests/golden/make_loss_golden.py), plus host-side checks of the GPU loss wrappers that need no GPU."""
import ctypes as C
import glob
import os

import numpy as np
import pytest
import torch

from oracle import loss_oracle as lo

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "loss_*.npz")))


def test_golden_files_present():
    assert len(GOLDEN) == 6


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: os.path.basename(p)[:-4])
def test_loss_oracle_matches_reference_golden(path):
    z = np.load(path)
    lam = float(z["lambda_dssim"])
    r = lo.l1_ssim(z["img"], z["gt"], lam)                      # float64 restatement
    assert abs(r["l1"] - float(z["l1_f64"])) <= 1e-12
    assert abs(r["ssim"] - float(z["ssim_f64"])) <= 1e-12
    assert abs(r["loss"] - float(z["loss_f64"])) <= 1e-12
    g = z["grad_f64"]
    assert np.abs(r["grad"] - g).max() <= 1e-10 * np.abs(g).max()
    # separable evaluation (what the kernels do): the window differs by the float32 rounding of its 121 products
    q = lo.l1_ssim(z["img"], z["gt"], lam, separable=True)
    assert abs(q["loss"] - r["loss"]) <= 1e-6 and np.abs(q["grad"] - g).max() <= 1e-5 * np.abs(g).max()
    # and the float32 run of the reference is within float32 noise of the float64 oracle
    assert abs(r["loss"] - float(z["loss_f32"])) <= 1e-6
    # (case f, a render close to its target: sigma = E[x^2] - mu^2 cancels, and the reference's own float32 gradient
    # is 1.4e-4 away from its float64 gradient; noise images: 1e-5)
    assert np.abs(r["grad"] - z["grad_f32"]).max() <= (5e-4 if path.endswith("loss_f.npz") else 2e-5) * np.abs(g).max()


def test_window_taps_match_the_reference_and_the_library():
    z = np.load(GOLDEN[0])
    np.testing.assert_array_equal(lo.window_taps(), z["window"])
    from diff_gaussian_rasterization import _backend as _C
    w = (C.c_float * 11)()
    assert _C.lib.msgs_ssim_window(w) == 0
    np.testing.assert_array_equal(np.array(list(w), dtype=np.float32), z["window"])
    assert _C.lib.msgs_ssim_window(None) == -1


def test_loss_host_logic_without_gpu():
    from loss_utils import l1_ssim_loss, ssim
    from diff_gaussian_rasterization import _backend as _C
    with pytest.raises(RuntimeError, match="GPU-only"):
        l1_ssim_loss(torch.zeros(3, 8, 8), torch.zeros(3, 8, 8))
    with pytest.raises(NotImplementedError):
        ssim(torch.zeros(3, 8, 8), torch.zeros(3, 8, 8), window_size=7)
    assert _C.lib.msgs_loss_scratch_bytes(3, 1080, 1920) >= 3 * 3 * 1080 * 1920 * 4
    assert _C.lib.msgs_loss_scratch_bytes(0, 4, 4) == 0
    assert _C.lib.msgs_loss_forward(None, None, 3, 4, 4, 0.2, None, None, 0, 1, None) == -1
