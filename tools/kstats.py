"""Short table of a rocprofv3 *kernel_stats.csv: python tools/kstats.py <csv> [n] -> name (shortened), calls, average us, total ms"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
def short(s):
    s = re.sub(r"\(anonymous namespace\)::", "", s)
    s = re.sub(r"^void ", "", s)
    m = re.match(r"([\w:]+(<[^()]*?>)?)", s)
    return (m.group(1) if m else s)[:90]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:n]:
    print(f"{short(r['Name']):92s} {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:10.1f} us {float(r['TotalDurationNs']) / 1e6:10.2f} ms")
