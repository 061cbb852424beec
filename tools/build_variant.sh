#!/bin/bash
# Links an experiment build of libslamhip.so in which ONE source file is replaced by a variant (development aid):
#     tools/build_variant.sh bf_hamming /path/to/variant.hip tools/exp/libslamhip_NAME.so [extra hipcc flags]
# The other objects are the shipped ones (slam-experiments_amd/lib/obj).  Nothing in the product or the tests loads the result.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$ROOT/slam-experiments_amd/csrc"
WHICH="$1"; VARIANT="$2"; OUT="$3"; shift 3
mkdir -p "$(dirname "$OUT")"
make -C "$SRC" -j8 all >/dev/null
TMP="$(mktemp -d)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$ROOT/include" -I/opt/rocm/include -I"$SRC" -fvisibility=hidden -DSLAM_BUILD "$@" \
    -x hip -c "$VARIANT" -o "$TMP/variant.o"
OBJS=$(ls "$ROOT"/slam-experiments_amd/lib/obj/*.o | grep -v "/$WHICH.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$TMP/variant.o" $OBJS -ldl
rm -rf "$TMP"
echo "built $OUT"
