"""Condenses the rocprofv3 output of tools/profile_round.sh into the small CSV summaries kept under profiles/:
kernel_stats.csv (per-kernel calls / total / average), pmc_hbm.csv (FETCH_SIZE, WRITE_SIZE and the corrected HBM bytes per
launch: (2*FETCH + WRITE) * 1024, MI355X_MICROARCH.md HBM section), pmc_sq.csv / pmc_sq2.csv (mean SQ counters per launch),
traffic.json (bench.py's roofline.traffic source) and idle.txt (GPU idle gaps per step, tools/idle_gaps.py).
usage: python tools/summarize_rocprof.py <prof_dir> <out_dir>"""
import csv, glob, json, os, re, subprocess, sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)


def csrc_sha256():
    """hash of the kernel sources the profiled library was built from (same function in bench.py): bench.py reports the
    committed counters only while this still matches the tree it runs from"""
    import hashlib
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ms-gs_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(root)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(root, name), "rb").read())
    return h.hexdigest()


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("msgs::(anonymous namespace)::", "").replace("msgs::", "")
    return re.sub(r"\(.*$", "", name)


def one(pattern):
    f = sorted(glob.glob(os.path.join(src, pattern), recursive=True))
    return f[0] if f else None


f = one("stats/**/*kernel_stats.csv")
if f:
    with open(f) as fh, open(os.path.join(dst, "kernel_stats.csv"), "w") as out:
        out.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline\n")
        out.write("name,calls,total_ns,avg_ns,pct\n")
        for r in csv.DictReader(fh):
            out.write('"%s",%s,%s,%d,%s\n' % (r["Name"][:120], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"]))
f = one("stats/**/*kernel_trace.csv")
if f:
    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "idle_gaps.py")
    r = subprocess.run([sys.executable, tool, f], capture_output=True, text=True)
    open(os.path.join(dst, "idle.txt"), "w").write(r.stdout + r.stderr)


def counters(sub):
    f = one(sub + "/**/*counter_collection.csv")
    acc = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    if not f:
        return acc, disp
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
    return acc, disp


fa, fd = counters("pmc_FETCH_SIZE")
wa, wd = counters("pmc_WRITE_SIZE")
if fa and wa:
    OWN = ("blend", "preprocess", "radix", "scan", "emit", "ranges", "zero_kernel", "adam", "ssim", "loss", "densify", "knn",
           "det_", "collect", "tile_stats", "visible_count", "voxel", "morton", "bbox")
    rows, traffic, merged = [], {}, defaultdict(lambda: [0.0, 0, 0.0, 0])
    for k in fa:
        if k not in wa:
            continue
        nf, nw = len(fd[k]), len(wd[k])
        fk, wk = fa[k]["FETCH_SIZE"] / nf, wa[k]["WRITE_SIZE"] / nw
        hb = (2 * fk + wk) * 1024
        rows.append((hb, k, nf, fk, wk))
        if k.startswith(OWN):                      # traffic.json: this library's kernels, template variants merged
            m = merged[re.sub(r"<.*$", "", k)]
            m[0] += fa[k]["FETCH_SIZE"]; m[1] += nf; m[2] += wa[k]["WRITE_SIZE"]; m[3] += nw
    for k, (fs, nf, ws, nw) in merged.items():
        traffic[k] = {"fetch_kib": fs / nf, "write_kib": ws / nw, "hbm_bytes": (2 * fs / nf + ws / nw) * 1024, "launches": nf}
    rows.sort(reverse=True)
    with open(os.path.join(dst, "pmc_hbm.csv"), "w") as out:
        out.write("# rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (two separate passes) -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing\n")
        out.write("# mean per launch; KiB as reported; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE reports half of a wide coalesced read)\n")
        out.write("kernel,launches,FETCH_SIZE_KiB,WRITE_SIZE_KiB,hbm_bytes_corrected\n")
        for hb, k, n, fk, wk in rows:
            if k.startswith(OWN) or hb > 1e7:
                out.write("%s,%d,%.0f,%.0f,%.4g\n" % (k, n, fk, wk, hb))
    json.dump({"source": "tools/profile_round.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing; mean per launch; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction of MI355X_MICROARCH.md, HBM section)",
               "csrc_sha256": csrc_sha256(), "kernels": traffic}, open(os.path.join(dst, "traffic.json"), "w"), indent=1)

sq_json = None
for sub, name in (("pmc_sq", "pmc_sq.csv"), ("pmc_sq2", "pmc_sq2.csv")):
    a, d = counters(sub)
    if not a:
        continue
    names = sorted({c for k in a for c in a[k]})
    with open(os.path.join(dst, name), "w") as out:
        out.write("# rocprofv3 --pmc %s (one pass, no trace domains) -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing\n" % " ".join(names))
        out.write("# mean per launch\nkernel,launches," + ",".join(names) + "\n")
        order = sorted(a, key=lambda k: -a[k].get("SQ_WAVE_CYCLES", a[k].get(names[0], 0)) / len(d[k]))
        for k in order[:24]:
            n = len(d[k])
            out.write(k + ",%d," % n + ",".join("%.4g" % (a[k][c] / n) for c in names) + "\n")
    if sub == "pmc_sq2" and sq_json is not None:      # second counter pass (LDS instructions, waits) joins the same keys
        for k in a:
            kk = re.sub(r"<.*$", "", k)
            if kk in sq_json["kernels"] and len(d[k]) >= sq_json["_launches"].get(kk, 0):
                sq_json["kernels"][kk].update({c: a[k][c] / len(d[k]) for c in names})
        sq_json.pop("_launches")
        json.dump(sq_json, open(os.path.join(dst, "sq.json"), "w"), indent=1)
    if sub == "pmc_sq":       # machine-readable twin for bench.py's informational VALU field
        sq_json = {"_launches": {re.sub(r"<.*$", "", k): len(d[k]) for k in sorted(order[:24], key=lambda k: len(d[k]))}}
        sq_json.update({"source": "rocprofv3 --pmc " + " ".join(names) + " -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing; mean per launch",
                   "csrc_sha256": csrc_sha256(),
                   # template variants share a key: keep the one launched most often (the hot one)
                   "kernels": {kk: vv for kk, vv in reversed([(re.sub(r"<.*$", "", k), {c: a[k][c] / len(d[k]) for c in names})
                                                              for k in sorted(order[:24], key=lambda k: -len(d[k]))])}})
        json.dump({k: v for k, v in sq_json.items() if k != "_launches"}, open(os.path.join(dst, "sq.json"), "w"), indent=1)
