#!/bin/bash
# L2 hit / miss and HBM traffic per kernel of any command: rocprofv3 --pmc passes (one counter set per pass, kernel-trace only).
# usage: tools/pmc_l2.sh <out.txt> <command ...>   (run on the GPU box via gpurun)
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  t=$(echo $c | tr ' ' '_')
  rm -rf /tmp/pl_$t
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pl_$t -o p -- "$@" > /tmp/pl_$t.log 2>&1
done
python - <<PY > $OUT
import csv, collections, re, glob
out = collections.defaultdict(dict)
for f in glob.glob("/tmp/pl_*/p_counter_collection.csv"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))
        agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        out[k][c] = sum(v) / len(v)
        out[k]["n"] = len(v)
print(f"{'kernel':70s} {'n':>4s} {'fetch(x2) MB':>13s} {'write MB':>10s} {'L2 hit':>12s} {'L2 miss':>12s} {'hit rate':>8s} {'L2 req':>12s} {'EA rdreq':>12s}")
for k, v in sorted(out.items(), key=lambda kv: -(kv[1].get("TCC_REQ_sum", 0) * kv[1]["n"]))[:30]:
    h, m = v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0)
    print(f"{k[:70]:70s} {v['n']:4d} {2 * v.get('FETCH_SIZE', 0) * 1024 / 1e6:13.1f} {v.get('WRITE_SIZE', 0) * 1024 / 1e6:10.1f} {h:12.0f} {m:12.0f} {h / max(h + m, 1):8.3f} "
          f"{v.get('TCC_REQ_sum', 0):12.0f} {v.get('TCC_EA0_RDREQ_sum', 0):12.0f}")
PY
cat $OUT
