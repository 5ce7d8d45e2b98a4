"""Per-step breakdown from a rocprofv3 kernel trace of bench.py: kernels between the last two SGD launches, grouped by name,
plus the GPU idle time inside the step (gaps between consecutive kernels).  usage: step_breakdown.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sgd = [i for i, r in enumerate(rows) if "sgd_momentum_flat_kernel" in r["Kernel_Name"]]
lo, hi = sgd[-2] + 1, sgd[-1] + 1
step = rows[lo:hi]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
by = defaultdict(lambda: [0, 0])
busy, gap, last_end = 0, 0, None
gaps, prev = [], None
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    k = k.split("(")[0][:80]
    by[k][0] += e - s
    by[k][1] += 1
    busy += e - s
    if last_end is not None and s > last_end:
        gap += s - last_end
        gaps.append((s - last_end, prev, k))
    prev = k
    last_end = max(last_end or e, e)
print("step wall %.3f ms, kernel busy %.3f ms, idle gaps %.3f ms, %d launches" % ((t1 - t0) / 1e6, busy / 1e6, gap / 1e6, len(step)))
for k, (ns, n) in sorted(by.items(), key=lambda kv: -kv[1][0]):
    print("%8.3f ms %4d  %s" % (ns / 1e6, n, k))
print("largest gaps (us: after -> before):")
for ns, a, b in sorted(gaps, reverse=True)[:14]:
    print("%8.1f  %s -> %s" % (ns / 1e3, a[:50], b[:50]))
print("gaps > 2 us: %d, sum %.3f ms" % (sum(1 for g in gaps if g[0] > 2000), sum(g[0] for g in gaps if g[0] > 2000) / 1e6))
