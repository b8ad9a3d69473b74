"""Randomised three-way parity sweep (long runs; the asserted 320-configuration version is tests/test_fuzz_parity_gpu.py).
usage: python tools/fuzz_parity.py [N=1000] [seed=0] [summary.json]
Prints one line per exceedance with its class (tests/fuzz_cases.py) and the summary; writes it as JSON when a path is given."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import fuzz_cases  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
results, s = fuzz_cases.run_sweep(N, seed)
print("fuzz summary:", json.dumps(s, indent=1))
if len(sys.argv) > 3:
    with open(sys.argv[3], "w") as f:
        json.dump({"n": N, "seed": seed, "summary": s,
                   "exceedances": [{"cfg": r["cfg"], "status": r["status"], "detail": r["detail"]}
                                   for r in results if r["status"] != "pass"]}, f, indent=1)
raise SystemExit(1 if s["unexplained"] else 0)
