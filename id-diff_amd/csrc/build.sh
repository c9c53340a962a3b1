#!/bin/bash
# Build libidiff_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
SRC="runtime upfirdn2d fused_bias_act igemm conv_narrow norm_act rng spectrum sbr winograd"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function"
# stamp of the sources this library is built from; _lib.lib() refuses a library whose stamp differs from the tree's
STAMP=$(cat $(ls *.hip *.h | LC_ALL=C sort) ../../include/idiff_hip.h | sha256sum | cut -c1-16)
FLAGS="$FLAGS -DIDIFF_SOURCE_STAMP=\"$STAMP\""
OBJ=$(mktemp -d)
trap 'rm -rf "$OBJ"' EXIT
pids=()
for s in $SRC; do
  extra=""
  # packed-f32 VALU (v_pk_add/fma_f32) costs extra issue cycles beside MFMAs: keep the Winograd transforms scalar
  [ "$s" = winograd ] && extra="-Xclang -target-feature -Xclang -packed-fp32-ops"
  # (the host half of the compilation does not know that feature and says so: filtered)
  "$HIPCC" $FLAGS $extra -c "$s.hip" -o "$OBJ/$s.o" 2> >(grep -v "not a recognized feature for this target" >&2) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -fPIC -shared -fvisibility=hidden -o libidiff_hip.so.tmp $(for s in $SRC; do echo "$OBJ/$s.o"; done)
mv -f libidiff_hip.so.tmp libidiff_hip.so
echo "built $(pwd)/libidiff_hip.so"
