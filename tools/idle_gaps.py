"""GPU idle gaps per step from a rocprofv3 --kernel-trace CSV of bench.py: python tools/idle_gaps.py <kernel_trace.csv>"""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
idx = [i for i, e in enumerate(ev) if 'blend_forward' in e[2]]
n = min(10, len(idx) - 4)
first = min(8, len(idx) - n - 1)
seg = ev[idx[first]:idx[first + n] + 1]
span = seg[-1][0] - seg[0][0]
busy = sum(e[1] - e[0] for e in seg[:-1])
print("per step: span %.1f us, busy %.1f us, idle %.1f us" % (span / n / 1e3, busy / n / 1e3, (span - busy) / n / 1e3))
gaps, cnt = collections.Counter(), collections.Counter()
for a, b in zip(seg[:-1], seg[1:]):
    g = b[0] - a[1]
    if g > 1500:
        key = (a[2].replace('msgs::(anonymous namespace)::', '')[:34], b[2].replace('msgs::(anonymous namespace)::', '')[:34])
        gaps[key] += g; cnt[key] += 1
for k, v in gaps.most_common(14):
    print("%7.1f us/step x%-3d %s -> %s" % (v / n / 1e3, cnt[k], k[0], k[1]))
busy_by = collections.Counter()
for e in seg[:-1]:
    busy_by[e[2].replace('msgs::(anonymous namespace)::', '')[:50]] += e[1] - e[0]
print("-- busy time per step by kernel")
for k, v in busy_by.most_common(16):
    print("%7.1f us  %s" % (v / n / 1e3, k))
