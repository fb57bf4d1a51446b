#!/bin/bash
# Run a command against a debug library (mage_amd/csrc/Makefile: `make debug`, `make asan`).
#   tools/debug_run.sh python -m pytest tests/test_gpu_ops.py -q                 UBSan (trapping) host code + device-side MAGE_DASSERT invariants
#   tools/debug_run.sh --asan python -m pytest tests/test_cpu_boundary.py -q      AddressSanitizer host code (host-only checks: no GPU allocation)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = "--asan" ]; then
    shift
    LIB="$ROOT/mage_amd/lib/libmage_hip_asan.so"
    [ -f "$LIB" ] || { echo "build it first: make -C mage_amd/csrc -j8 asan"; exit 2; }
    ASANRT=$(/opt/rocm/bin/hipcc -print-file-name=libclang_rt.asan-x86_64.so)
    # detect_leaks=0: python keeps allocations for the life of the process
    LD_PRELOAD="$ASANRT" ASAN_OPTIONS=detect_leaks=0:detect_odr_violation=0 MAGE_HIP_LIB="$LIB" "$@"
else
    LIB="$ROOT/mage_amd/lib/libmage_hip_debug.so"
    [ -f "$LIB" ] || { echo "build it first: make -C mage_amd/csrc -j8 debug"; exit 2; }
    MAGE_HIP_LIB="$LIB" "$@"
fi
