#!/bin/bash
# Matrix-core utilisation and the clock the chip actually holds, per kernel, from PMC counters of the bench command (one extra rocprofv3
# --pmc pass, kernel-trace only): SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES per kernel, and GRBM_GUI_ACTIVE / kernel duration = effective
# shader clock (MI355X_MICROARCH.md, DVFS).  Output: gpurun_out/pmc_mfma_<tag>.txt   usage: tools/pmc_mfma.sh r04
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_mfma
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d /tmp/pmc_mfma -o p -- \
  python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-mode --no-parity-mode --no-decode-roofline --no-train-step --no-latency-b1 --streams 1 > /tmp/pmc_mfma.log 2>&1
python - <<PY > $GRAFT_REPO_ROOT/gpurun_out/pmc_mfma_$TAG.txt
import csv, collections, re
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
did = {}
for r in csv.DictReader(open("/tmp/pmc_mfma/p_kernel_trace.csv")):
    name = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))
    did[r["Dispatch_Id"]] = (name, int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for r in csv.DictReader(open("/tmp/pmc_mfma/p_counter_collection.csv")):
    name, d = did.get(r["Dispatch_Id"], (None, None))
    if name is None:
        continue
    cnt[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        dur[name].append(d)
print("per kernel (averages over its launches in one profiled bench run; counters are chip-wide sums unless noted):")
print("GUI = GRBM_GUI_ACTIVE / 8 XCDs (cycles of the kernel's span on the GRBM clock); MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GUI x 1024 SIMDs);")
print("a profiled run serialises the launches (idle gaps between them): its clocks are HIGHER than the back-to-back bench run's (profiles/r04_clock_probe.txt)")
print(f"{'kernel':62s} {'n':>4s} {'us':>8s} {'GUI/us MHz':>10s} {'MFMA_BUSY':>12s} {'MfmaUtil':>9s} {'INSTS_MFMA':>11s} {'INSTS_VALU':>11s}")
rows = []
for k, c in cnt.items():
    n = len(c.get("GRBM_GUI_ACTIVE", []))
    if not n:
        continue
    avg = lambda key: sum(c.get(key, [0.0])) / max(1, len(c.get(key, [])))
    us = sum(dur[k]) / len(dur[k]) / 1e3
    gui = avg("GRBM_GUI_ACTIVE") / 8.0
    rows.append((us * n, k, n, us, gui / us if us else 0.0, avg("SQ_VALU_MFMA_BUSY_CYCLES"), avg("SQ_VALU_MFMA_BUSY_CYCLES") / max(1.0, gui * 1024.0),
                 avg("SQ_INSTS_MFMA"), avg("SQ_INSTS_VALU")))
for tot, k, n, us, mhz, mb, mu, im, iv in sorted(rows, reverse=True)[:14]:
    print(f"{k[:62]:62s} {n:4d} {us:8.1f} {mhz:10.0f} {mb:12.4g} {mu:9.3f} {im:11.4g} {iv:11.4g}")
PY
cat $GRAFT_REPO_ROOT/gpurun_out/pmc_mfma_$TAG.txt
