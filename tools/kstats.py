"""Print a rocprofv3 `*kernel_stats.csv` as: calls, total ms, average us, short kernel name (kernel names contain commas)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for r in rows[:top]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print("%6d %10.3f ms %10.1f us  %s" % (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, name[:90]))
