"""The randomised three-way parity sweep in front of the driver (tests/fuzz_cases.py has the criterion and the classes): 320
seeded configurations over every forward variant, both backward generations, the fine kernels, fade > 0, the chained and the
plain getter path, and the precomputed-colour / precomputed-covariance entries, each against the float32 oracle, the float64
truth and the FMA-contracted float32 oracle.  Asserts ZERO unexplained exceedances and bounds every explained class by count."""
import json
import os

import pytest

pytestmark = pytest.mark.gpu

N, SEED = 320, 20251002
# Explained-exceedance budgets, in configurations of the 320 (measured on MI355X: profiles/r5_parity.md)
BUDGET = {"shared_borderline_pixel": 4, "oracle_f32_off_truth": 3, "float32_rounding_mode": 3, "k8_conditioning": 4,
          "cancelled_sum": 1}


def test_randomised_three_way_sweep_has_no_unexplained_exceedance():
    import fuzz_cases
    results, s = fuzz_cases.run_sweep(N, SEED)
    print("[fuzz] summary", json.dumps(s))
    out = os.environ.get("MSGS_FUZZ_SUMMARY")
    if out:
        with open(out, "w") as f:
            json.dump({"summary": s, "exceedances": [{"cfg": r["cfg"], "status": r["status"], "detail": r["detail"]}
                                                     for r in results if r["status"] != "pass"]}, f, indent=1)
    bad = [r for r in results if r["status"] == "unexplained"]
    assert not bad, "\n".join(f"{r['cfg']} :: {r['detail']}" for r in bad[:10])
    for c, n in BUDGET.items():
        assert s[c] <= n, (c, s[c], n)
    assert s["pass"] >= N - sum(BUDGET.values())
    # the exclusions of the strict checks stay rare over the sweep as a whole (single tiny scenes can exceed any fraction)
    assert s["borderline_pixel_fraction"] < 0.0045 and s["tier1_gaussian_fraction"] < 0.03, s
    # the multi-scale models rendered without their filters did give the occlusion cut-off something to cut
    mw = s["multiscale_without_filters"]
    assert mw["configurations"] >= 40 and mw["with_closed_blocks"] >= 8 and mw["with_every_block_closed"] >= 4, mw
    # every kernel variant was drawn
    cfgs = [r["cfg"] for r in results]
    assert {c["fwd_var"] for c in cfgs} == {0, 1, 3, 4, 5, 6} and {c["bwd_gen"] for c in cfgs} == {0, 1, 2}
    assert {c["gran"] for c in cfgs} == {0, 1, 2} and {c["entry"] for c in cfgs} == {"render", "precomp_col", "precomp_cov", "precomp_both"}
    assert any(c["fade"] > 0 and c["ms"] for c in cfgs) and {c["chain"] for c in cfgs} == {True, False}
