"""Re-run the exceedances of a sweep summary (tools/fuzz_parity.py's JSON) under the current classifier.
usage: python tools/fuzz_rerun.py summary.json [status=unexplained|all] [out.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import fuzz_cases  # noqa: E402

src = json.load(open(sys.argv[1]))
want = sys.argv[2] if len(sys.argv) > 2 else "unexplained"
rows = []
for e in src["exceedances"]:
    if want != "all" and e["status"] != want:
        continue
    r = fuzz_cases.run_config(e["cfg"])
    print(f"[rerun] was {e['status']} -> {r['status']}: {e['cfg']} :: {r['detail']}")
    rows.append({"cfg": e["cfg"], "was": e["status"], "status": r["status"], "detail": r["detail"]})
if len(sys.argv) > 3:
    json.dump({"source": os.path.basename(sys.argv[1]), "reruns": rows}, open(sys.argv[3], "w"), indent=1)
