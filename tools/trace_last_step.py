"""Per-kernel summary of the LAST training step in a rocprofv3 kernel trace (the whole-run --stats table also holds
MIOpen's one-off find-mode kernels from the warm-up steps).  usage: trace_last_step.py <trace_kernel_trace.csv> <step_ms> <out.csv>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
end = int(rows[-1]["End_Timestamp"])
win = [r for r in rows if int(r["Start_Timestamp"]) > end - int(float(sys.argv[2]) * 1e6)]
agg = collections.defaultdict(lambda: [0, 0])
for r in win:
    a = agg[r["Kernel_Name"]]
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
with open(sys.argv[3], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k[:160], v[0], v[1], round(v[1] / v[0], 1), round(100.0 * v[1] / tot, 3)])
    w.writerow(["TOTAL (busy) over a window of %s ms" % sys.argv[2], len(win), tot, "", 100.0])
print("busy ms", tot / 1e6, "kernels", len(win))
