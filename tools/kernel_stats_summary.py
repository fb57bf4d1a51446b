"""Print a rocprofv3 kernel_stats.csv as ms per benchmark step.  usage: kernel_stats_summary.py stats.csv n_steps_incl_warmup"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time per step: {tot / n / 1e6:.3f} ms")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    print(f"{name[:70]:70s} calls/step {int(r['Calls']) / n:7.1f}  ms/step {float(r['TotalDurationNs']) / n / 1e6:8.3f}  avg_us {float(r['AverageNs']) / 1e3:9.1f}")
