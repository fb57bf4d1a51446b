#!/bin/bash
# rocprofv3 kernel stats of the incremental AR mode at cfg2 (eager): per-kernel totals of 5 calls + GPU-busy of the last call
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ki
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ki -o k -- python $GRAFT_REPO_ROOT/tools/inc_profile.py ${1:-incremental} > $OUT/inc_profile.txt 2>&1
cp /tmp/ki/k_kernel_stats.csv $OUT/inc_kernel_stats.csv
python - <<PY >> $OUT/inc_profile.txt
import csv
rows = list(csv.DictReader(open("/tmp/ki/k_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // 5
last = rows[-n:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last)
span = int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])
print(f"last call: {n} launches, busy {busy/1e6:.2f} ms, span {span/1e6:.2f} ms, idle {100*(1-busy/span):.1f} %")
agg = {}
for r in last:
    k = r["Kernel_Name"][:100] + f"  grid={r['Grid_Size_X']}x{r['Grid_Size_Y']} wg={r['Workgroup_Size_X']}"
    a = agg.setdefault(k, [0, 0])
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
# one AR iteration in launch order (the 60 launches that end the second-to-last decoder iteration of the last call)
import re
seq = [r for r in last if "Cijk" not in r["Kernel_Name"]]
idx = [i for i, r in enumerate(seq) if "argmax" in r["Kernel_Name"]]
if len(idx) >= 3:
    a, b = idx[-3], idx[-2]
    t0 = int(seq[a]["End_Timestamp"])
    print(f"--- one AR iteration ({b - a} launches, {(int(seq[b]['End_Timestamp']) - t0) / 1e3:.1f} us):")
    for r in seq[a + 1:b + 1]:
        nm = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:70]
        print(f"   +{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f}  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} us  {nm}  grid={r['Grid_Size_X']}")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{t/1e3:9.1f} us  {c:4d} x {t/c/1e3:7.1f} us  {k}")
PY
cat $OUT/inc_profile.txt
python - <<PY
import csv
rows = list(csv.DictReader(open("/tmp/ki/k_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // 5
last = [r for r in rows[-n:] if "fewq" in r["Kernel_Name"] or ("attention_mfma_kernel" in r["Kernel_Name"] and int(r["Grid_Size_X"]) > 2000000)]
print("temporal attention launches of the last call (us):", " ".join(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.0f}" for r in last))
PY
