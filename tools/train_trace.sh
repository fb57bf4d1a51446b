#!/bin/bash
# rocprofv3 kernel stats of the training step (tools/train_probe.py B L precision iters): per-kernel totals per step
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -o k -- python $GRAFT_REPO_ROOT/tools/train_probe.py ${1:-64} ${2:-16} ${3:-bf16} 6 > $OUT/train_trace.txt 2>&1
python - <<PY >> $OUT/train_trace.txt
import csv, re
rows = list(csv.DictReader(open("/tmp/kt2/k_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last two steps: find the adam kernels
idx = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
a, b = idx[-3], idx[-1]
seg = rows[a + 1:b + 1]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
span = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
print(f"last 2 steps: {len(seg)} launches, busy {busy/2e6:.2f} ms per step, span {span/2e6:.2f} ms per step")
agg = {}
for r in seg:
    k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:95]
    x = agg.setdefault(k, [0, 0])
    x[0] += 1
    x[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{t/2e6:8.3f} ms/step {c//2:4d} x {t/c/1e3:8.1f} us  {k}")
PY
tail -45 $OUT/train_trace.txt
