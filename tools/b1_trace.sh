#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kb1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kb1 -o k -- python $GRAFT_REPO_ROOT/tools/b1_profile.py ${1:-incremental} > $OUT/b1_profile.txt 2>&1
python - <<PY >> $OUT/b1_profile.txt
import csv, re
rows = list(csv.DictReader(open("/tmp/kb1/k_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // 6
last = rows[-n:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last)
span = int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])
print(f"last call: {n} launches, busy {busy/1e6:.2f} ms, span {span/1e6:.2f} ms")
agg = {}
for r in last:
    k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:90]
    a = agg.setdefault(k, [0, 0])
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{t/1e3:9.1f} us  {c:4d} x {t/c/1e3:7.1f} us  {k}")
PY
grep -v "^W2\|^E2" $OUT/b1_profile.txt
