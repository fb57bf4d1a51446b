#!/bin/bash
# Regenerate EVERY fixture under tests/golden/ from the reference (/root/reference, build container only) into a temp dir and compare the
# values with the committed files: the pin of the oracle is "outputs of the reference itself", so the committed fixtures must be exactly what
# the committed generators produce.  Exit code 0 = identical (same keys, max |delta| = 0 for every array, identical JSON).
set -e
cd "$(dirname "$0")/.."
T=$(mktemp -d)
trap 'rm -rf "$T"' EXIT
MAGE_GOLDEN_OUT="$T" python tools/gen_golden.py > "$T/gen.log" 2>&1 || { tail -20 "$T/gen.log"; exit 1; }
MAGE_GOLDEN_OUT="$T" python tools/gen_golden_glue.py >> "$T/gen.log" 2>&1 || { tail -20 "$T/gen.log"; exit 1; }
python - "$T" <<'PY'
import glob, json, os, sys
import numpy as np
new, old = sys.argv[1], os.path.join("tests", "golden")
bad = 0
names = sorted(set(os.path.basename(p) for p in glob.glob(os.path.join(old, "*.npz")) + glob.glob(os.path.join(new, "*.npz"))))
for n in names:
    a, b = os.path.join(old, n), os.path.join(new, n)
    if not (os.path.exists(a) and os.path.exists(b)):
        print(f"{n}: only in {'committed' if os.path.exists(a) else 'regenerated'} set"); bad += 1; continue
    A, B = np.load(a, allow_pickle=True), np.load(b, allow_pickle=True)
    if set(A.files) != set(B.files):
        print(f"{n}: keys differ: {sorted(set(A.files) ^ set(B.files))}"); bad += 1; continue
    worst = 0.0
    for k in A.files:
        x, y = A[k], B[k]
        if x.shape != y.shape or x.dtype != y.dtype:
            print(f"{n}[{k}]: {x.dtype}{x.shape} vs {y.dtype}{y.shape}"); bad += 1; continue
        if x.dtype.kind in "fiub":
            d = float(np.abs(x.astype(np.float64) - y.astype(np.float64)).max()) if x.size else 0.0
            worst = max(worst, d)
        elif not np.array_equal(x, y):
            print(f"{n}[{k}]: differs"); bad += 1
    print(f"{n}: {len(A.files)} arrays, max |delta| = {worst:g}")
    bad += worst != 0.0
for n in ("state_dict_layout.json", "glue_dataload.json"):
    same = json.load(open(os.path.join(old, n))) == json.load(open(os.path.join(new, n)))
    print(f"{n}: {'identical' if same else 'DIFFERS'}")
    bad += not same
print("fixtures identical to the regenerated set" if not bad else f"{bad} fixture(s) differ")
sys.exit(1 if bad else 0)
PY
