#!/bin/bash
# HBM traffic of ONE VectorQuantizedVAE.decode call (960 frames, bf16) from PMC counters: two separate rocprofv3 --pmc passes over
# tools/bench_vqvae.py (FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel-trace only; corrected as MI355X_MICROARCH.md prescribes
# (gfx950: FETCH_SIZE reports half of wide streaming reads).  Output: gpurun_out/pmc_decode_<tag>.json (copy to profiles/<tag>_pmc_decode.json:
# bench.py reads it for roofline_decode.traffic).  usage: tools/pmc_decode.sh r04
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcd_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcd_$c -o p -- python $GRAFT_REPO_ROOT/tools/bench_vqvae.py frames=960 > /tmp/pmcd_$c.log 2>&1
done
python - <<PY
import csv, collections, json, re
# launches of one decode call (mage_amd/modules/vqvae_model.py: _decode_chunk, bf16, 16x16 latents, dim 256)
per_call = {"resblock_table_kernel": 1, "gemm8_kernel<1, 0, false, true, 0, 0, false, false>": 1, "resblock_rows_kernel": 1,
            "gemm8_kernel<1, 0, false, true, 5, 0, false, false>": 1, "convt_fold_tanh_img_kernel": 1}
out = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f"/tmp/pmcd_{c}/p_counter_collection.csv")):
        if r["Counter_Name"] == c:
            name = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))
            agg[name].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if k in per_call:
            out[k][c] = sum(v) / len(v) * 1024
            out[k]["launches_measured"] = len(v)
res, tot = {}, 0.0
for k, n in per_call.items():
    f, w = out[k].get("FETCH_SIZE", 0.0), out[k].get("WRITE_SIZE", 0.0)
    res[k] = {"launches_per_call": n, "launches_measured": out[k].get("launches_measured", 0), "fetch_bytes_corrected": 2 * f, "write_bytes": w,
              "hbm_bytes_per_launch": 2 * f + w}
    tot += n * (2 * f + w)
    print(f"{k[:56]:56s} x{n}  fetch(x2) {2 * f / 1e6:8.1f} MB  write {w / 1e6:8.1f} MB")
res["hbm_bytes_per_call"] = tot
res["frames"] = 960
print(f"one decode call of 960 frames: {tot / 1e6:.1f} MB of HBM traffic ({tot / 960 / 1e6:.3f} MB per frame)")
json.dump(res, open("$GRAFT_REPO_ROOT/gpurun_out/pmc_decode_$TAG.json", "w"), indent=1)
PY
