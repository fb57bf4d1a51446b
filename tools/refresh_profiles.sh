#!/bin/bash
# One gpurun call that regenerates everything under profiles/ for the committed code:
#   bench line (with cpu_baseline), bench line under rocprofv3 + kernel stats, PMC traffic.  usage: tools/refresh_profiles.sh r01
TAG=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
# the bench line first, on a rested chip (the PMC passes before it cost the line 3-6 % on some boxes); roofline.traffic comes from the
# committed PMC summary (profiles/<tag>_pmc_traffic.json), refreshed by the passes at the end of this script for the next commit
cd $GRAFT_REPO_ROOT
python bench.py 2> $OUT/bench_$TAG.err | tail -1 > $OUT/${TAG}_bench_cfg2_bf16.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-mode --no-parity-mode --no-decode-roofline --no-train-step --no-latency-b1 --no-cfg4 2> /dev/null | tail -1 > $OUT/${TAG}_bench_cfg2_bf16_under_rocprof.json
cp /tmp/kt/k_kernel_stats.csv $OUT/${TAG}_bench_cfg2_bf16_kernel_stats.csv
# GPU busy vs span of the timed step (last generate call): sum of kernel durations / (last end - first start) over the second half
python - <<PY > $OUT/${TAG}_gpu_busy.txt
import csv
rows = list(csv.DictReader(open("/tmp/kt/k_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
half = rows[len(rows) // 2:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in half)
span = int(half[-1]["End_Timestamp"]) - int(half[0]["Start_Timestamp"])
print(f"second half of the kernel trace (the timed generate call): {len(half)} launches, busy {busy/1e6:.2f} ms, span {span/1e6:.2f} ms, idle {100*(1-busy/span):.1f} %")
PY
# VQ-VAE decode on its own (roofline_decode of the bench line): per-kernel rows
rm -rf /tmp/kd
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kd -o k -- python $GRAFT_REPO_ROOT/tools/bench_vqvae.py > $OUT/${TAG}_decode_bench.txt 2>&1
cp /tmp/kd/k_kernel_stats.csv $OUT/${TAG}_decode_kernel_stats.csv
python $GRAFT_REPO_ROOT/tools/decode_timeline.py /tmp/kd > $OUT/${TAG}_decode_timeline.txt 2>&1      # one decode call, launch by launch, with the gaps
bash $GRAFT_REPO_ROOT/tools/pmc_bench.sh $TAG > $OUT/pmc_$TAG.txt 2>&1
cat $OUT/${TAG}_gpu_busy.txt; head -c 600 $OUT/${TAG}_bench_cfg2_bf16.json; echo; tail -9 $OUT/pmc_$TAG.txt
