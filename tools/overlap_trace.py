"""Overlap summary of a rocprofv3 --kernel-trace CSV of tools/two_view_loop.py:
python tools/overlap_trace.py <kernel_trace.csv> [views_in_window]
Takes the last 60 % of the trace (steady state), reports per view: wall span, summed kernel time, time with >= 1 / >= 2 kernels
running, and the mean duration of every kernel class (to compare a serial and a pipelined run: slowdown under co-residency)."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
t_lo = ev[0][0] + 0.4 * (ev[-1][1] - ev[0][0])
seg = [e for e in ev if e[0] >= t_lo]
views = sum(1 for e in seg if 'blend_forward' in e[2] and 'order' not in e[2]) or 1
span = seg[-1][1] - seg[0][0]
pts = sorted([(e[0], 1) for e in seg] + [(e[1], -1) for e in seg])
busy1 = busy2 = 0
depth, last = 0, pts[0][0]
for t, d in pts:
    if depth >= 1: busy1 += t - last
    if depth >= 2: busy2 += t - last
    depth += d; last = t
total = sum(e[1] - e[0] for e in seg)
print("views in window %d | per view: span %.1f us, sum of kernel time %.1f us, >=1 kernel %.1f us, >=2 kernels %.1f us, idle %.1f us"
      % (views, span / views / 1e3, total / views / 1e3, busy1 / views / 1e3, busy2 / views / 1e3, (span - busy1) / views / 1e3))
by, cnt = collections.Counter(), collections.Counter()
for e in seg:
    k = e[2].replace('msgs::(anonymous namespace)::', '').replace('void at::native::', '')[:60]
    by[k] += e[1] - e[0]; cnt[k] += 1
print("-- per view and kernel: total us | launches per view | mean us per launch")
for k, v in by.most_common(24):
    print("%8.1f us | %5.2f | %8.1f us  %s" % (v / views / 1e3, cnt[k] / views, v / cnt[k] / 1e3, k))
