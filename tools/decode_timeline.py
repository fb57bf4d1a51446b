"""One VectorQuantizedVAE.decode call launch by launch from a rocprofv3 --kernel-trace directory: start offset, gap to the previous kernel, duration.
usage: python tools/decode_timeline.py <rocprof output dir>   (tools/refresh_profiles.sh runs it on tools/bench_vqvae.py's trace)"""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the last occurrence of the decode sequence: from the last resblock_table kernel
idx = [i for i, r in enumerate(rows) if "resblock_table" in r["Kernel_Name"]]
i0 = idx[-2] if len(idx) > 1 else idx[-1]
t0 = int(rows[i0]["Start_Timestamp"]); prev_end = t0
for j, r in enumerate(rows[i0:i0 + 14]):
    if j and "resblock_table" in r["Kernel_Name"]:
        break                                          # the next call
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:7.1f}  {r['Kernel_Name'][:90]}")
    prev_end = e
print(f"one call: {(prev_end - t0) / 1e3:.1f} us from the first kernel's start to the last one's end")
