"""Summarises the output of tools/diag_det_vs_atomic.py read from stdin (median / p90 / max per path and tensor)."""
import sys,re,collections
d=collections.defaultdict(list)
for l in sys.stdin:
    m=re.match(r"(\d+) (\w+) vs oracle: (.*)", l)
    if m:
        dd=eval(m.group(3))
        for k in ("scaling","rotation","means3D","means2D","opacity"):
            d[(m.group(2),k)].append(float(dd[k]))
import statistics
for k,v in sorted(d.items()):
    v=sorted(v); print(k, "median %.1e  p90 %.1e  max %.1e  n>1e-4: %d/%d"%(statistics.median(v), v[int(.9*len(v))], v[-1], sum(x>1e-4 for x in v), len(v)))
