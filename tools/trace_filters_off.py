"""Kernel trace subject: 40 forward-only renders of the C3 model WITHOUT its filters at level 0 (the occlusion cut-off's view).
usage (GPU box): cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT &&
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_foff -- python3 tools/trace_filters_off.py
then tools/summarize_kernel_stats (below, `python3 tools/trace_filters_off.py --summary <dir> <out.csv>`): per-kernel calls / average."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--summary":
    src, dst = sys.argv[2], sys.argv[3]
    rows = []
    for f in glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    with open(dst, "w") as o:
        o.write("# rocprofv3 --kernel-trace --stats -- python3 tools/trace_filters_off.py (3 warm-up + 40 timed forward renders, C3 model,\n"
                "# 1920x1080, filters off, occlusion cut-off in its default state); name,calls,total_ns,avg_ns\n")
        for r in rows:
            o.write(f"\"{r['Name'][:140]}\",{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f}\n")
    raise SystemExit(0)

for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import scenes  # noqa: E402
from gaussian_renderer import PIPE, render  # noqa: E402
from synthetic_model import SyntheticGaussians  # noqa: E402

sc, _, _ = scenes.config("C3")
pc = SyntheticGaussians(sc, "cuda", requires_grad=False)
bg = torch.zeros(3, device="cuda")
cam = scenes.front_camera(1920, 1080).to("cuda")
with torch.no_grad():
    for _ in range(43):
        out = render(cam, pc, PIPE, bg, filter_small=False, filter_large=False, fade_size=1.0)
torch.cuda.synchronize()
print("done", float(out["render"].mean()))
