#!/bin/bash
# HBM traffic per kernel of any command from PMC counters: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), kernel-trace only, corrected as
# MI355X_MICROARCH.md prescribes (gfx950: FETCH_SIZE x2).  usage: tools/pmc_any.sh <out.txt> <command ...>   (run on the GPU box via gpurun)
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pa_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pa_$c -o p -- "$@" > /tmp/pa_$c.log 2>&1
done
python - <<PY > $OUT
import csv, collections, re
out = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f"/tmp/pa_{c}/p_counter_collection.csv")):
        if r["Counter_Name"] == c:
            name = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))
            agg[name].append(float(r["Counter_Value"]) * 1024)
    for k, v in agg.items():
        out[k][c] = sum(v) / len(v)
        out[k]["n"] = len(v)
rows = [(k, v["n"], 2 * v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)) for k, v in out.items()]
print(f"{'kernel':72s} {'launches':>8s} {'fetch(x2) MB':>13s} {'write MB':>10s} {'total GB (all launches)':>24s}")
for k, n, f, w in sorted(rows, key=lambda r: -(r[2] + r[3]) * r[1])[:30]:
    print(f"{k[:72]:72s} {n:8d} {f / 1e6:13.1f} {w / 1e6:10.1f} {(f + w) * n / 1e9:24.2f}")
PY
cat $OUT
