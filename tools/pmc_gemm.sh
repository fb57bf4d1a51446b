#!/bin/bash
# usage: tools/pmc_gemm.sh [libpath]  -- PMC counters of the decoder-stack GEMM shapes (tuning only)
cd /tmp && export TMPDIR=/tmp
LIB=${1:-}
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  rm -rf /tmp/pmc; MAGE_HIP_LIB=$LIB rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc -o p -- python $GRAFT_REPO_ROOT/tools/bench_gemm.py --rounds 1 > /dev/null 2>&1
  python - <<PY
import csv, collections
rows = list(csv.DictReader(open("/tmp/pmc/p_counter_collection.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "gemm_kernel" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"][40:75], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k, {c: f"{sum(x)/len(x):.4g}" for c, x in v.items()})
PY
done
