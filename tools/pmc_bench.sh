#!/bin/bash
# HBM traffic of the bench kernels from PMC counters: two separate rocprofv3 --pmc passes (FETCH_SIZE needs 3 TCC
# slots, WRITE_SIZE 2: they do not fit one pass), kernel-trace only.  Output: gpurun_out/pmc_<tag>.json
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-mode --no-parity-mode --no-decode-roofline --no-train-step --no-latency-b1 --no-cfg4 --streams 1 > /tmp/pmc_$c.log 2>&1
done
python - <<PY
import csv, collections, json, re
out = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f"/tmp/pmc_{c}/p_counter_collection.csv")):
        if r["Counter_Name"] == c:
            name = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))
            agg[name].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out[k][c + "_KB_avg"] = sum(v) / len(v)
        out[k]["launches"] = len(v)
        # a symbol that serves two shapes (out_proj K = 512 and c_proj K = 2048 share the x + Linear(.) producer): the two clusters of its launches
        m = sum(v) / len(v)
        lo, hi = [x for x in v if x < m], [x for x in v if x >= m]
        if lo and hi and min(hi) > 1.3 * max(lo):
            out[k][c + "_KB_clusters"] = [[len(lo), sum(lo) / len(lo)], [len(hi), sum(hi) / len(hi)]]
res = {}
for k, v in out.items():
    f, w = v.get("FETCH_SIZE_KB_avg", 0.0), v.get("WRITE_SIZE_KB_avg", 0.0)
    # MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads
    # (16 B/lane global_load and buffer_load..lds alike) -> doubled; WRITE_SIZE taken as is (KB -> bytes x1024)
    res[k] = {"launches": v["launches"], "fetch_bytes_raw": f * 1024, "fetch_bytes_corrected": 2 * f * 1024, "write_bytes": w * 1024,
              "hbm_bytes_per_launch": (2 * f + w) * 1024}
    if "FETCH_SIZE_KB_clusters" in v:
        res[k]["fetch_bytes_corrected_clusters"] = [[n, 2 * x * 1024] for n, x in v["FETCH_SIZE_KB_clusters"]]
    if "WRITE_SIZE_KB_clusters" in v:
        res[k]["write_bytes_clusters"] = [[n, x * 1024] for n, x in v["WRITE_SIZE_KB_clusters"]]
json.dump(res, open("$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG.json", "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:8]:
    print(f"{k[:60]:60s} n={v['launches']:4d} fetch(x2) {v['fetch_bytes_corrected']/1e6:9.1f} MB write {v['write_bytes']/1e6:9.1f} MB")
PY
