"""Timeline of ONE training step from a rocprofv3 kernel trace (csv) of bench.py: every kernel between the last two optimizer launches in
start order with its start / duration (us), the queue it ran on and how much of its lifetime another kernel was running beside it -
the view that shows what the second (weight-gradient) stream hides and where the chip idles.
usage: python tools/timeline.py <kernel_trace.csv> [min_us=0]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sgd = [i for i, r in enumerate(rows) if "sgd_momentum_flat_kernel" in r["Kernel_Name"]]
step = rows[sgd[-2] + 1:sgd[-1] + 1]
t0 = int(step[0]["Start_Timestamp"])
iv = [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0) for r in step]
qcol = "Queue_Id" if "Queue_Id" in step[0] else ("Stream_Id" if "Stream_Id" in step[0] else None)
# union busy time / idle
ev = sorted([(s, 1) for s, e in iv] + [(e, -1) for s, e in iv])
busy = both = 0
depth, last = 0, 0
for t, d in ev:
    if depth >= 1:
        busy += t - last
    if depth >= 2:
        both += t - last
    depth += d
    last = t
wall = max(e for _, e in iv)
print("step wall %.1f us | >=1 kernel running %.1f us | >=2 running %.1f us | idle %.1f us | sum of kernel durations %.1f us | %d launches"
      % (wall / 1e3, busy / 1e3, both / 1e3, (wall - busy) / 1e3, sum(e - s for s, e in iv) / 1e3, len(step)))
for i, (r, (s, e)) in enumerate(zip(step, iv)):
    ov = 0
    for j, (s2, e2) in enumerate(iv):
        if j != i:
            ov = max(ov, 0) + max(0, min(e, e2) - max(s, s2))
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    if (e - s) / 1e3 >= min_us:
        print("%9.1f +%7.1f us  q%-3s shared %5.0f%%  %s" % (s / 1e3, (e - s) / 1e3, (r[qcol] if qcol else "?"), 100.0 * min(ov, e - s) / max(1, e - s), k))
