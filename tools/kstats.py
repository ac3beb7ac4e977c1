"""Per-step view of a rocprofv3 kernel_stats.csv: python tools/kstats.py <csv> <steps> [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 24
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:top]:
    print(f'{r["Name"][:62]:62s} calls {int(r["Calls"]):5d} {float(r["TotalDurationNs"]) / steps / 1e3:9.1f} us/step  avg {float(r["AverageNs"]) / 1e3:8.1f} us')
print(f"sum of kernel durations: {tot / steps / 1e6:.3f} ms per step")
