#!/bin/bash
# The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only: GPU sanitizers are not available on the pool).
# Builds oracle/*.c into a scratch library with -fsanitize=address,undefined and runs the CPU tests that exercise the oracle
# against it (golden vectors, known-answer files, the LM and BA oracles).  Development aid.
#     tools/oracle_sanitize.sh
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
TMP="$(mktemp -d)"
gcc -fsanitize=address,undefined -fno-omit-frame-pointer -O1 -g -shared -fPIC -fopenmp -mpopcnt -std=c11 "$ROOT"/oracle/*.c -o "$TMP/liboracle_san.so" -lm
cat > "$TMP/run.py" <<PY
import sys
sys.path.insert(0, "$ROOT")
import oracle.oracle as o
o._LIB_PATH = "$TMP/liboracle_san.so"
import pytest
sys.exit(pytest.main(["$ROOT/tests", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
                      "-k", "oracle or golden or kat or reproj or lm or ba or filter"]))
PY
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
    UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 OMP_NUM_THREADS=4 python3 "$TMP/run.py"
rc=$?
rm -rf "$TMP"
exit $rc
