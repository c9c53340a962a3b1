#!/bin/bash
# Build libidiff_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
SRC="runtime upfirdn2d fused_bias_act igemm conv_narrow norm_act rng spectrum sbr winograd winograd43 winograd43h wino1d attention"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function"
# stamp of the sources this library is built from; _lib.lib() refuses a library whose stamp differs from the tree's
STAMP=$(cat $(ls *.hip *.h | LC_ALL=C sort) ../../include/idiff_hip.h | sha256sum | cut -c1-16)
FLAGS="$FLAGS -DIDIFF_SOURCE_STAMP=\"$STAMP\""
# VARIANT builds (diagnostic / A-B kernels of scripts/*.py, often wrong by construction): IDIFF_VARIANT=<name> with
# IDIFF_VARIANT_FLAGS="-D..." goes to libidiff_hip.<name>.so and NEVER to libidiff_hip.so; the flags are compiled into the library
# (idiff_variant_flags) and _lib.lib() refuses a library that reports any at the production path.
VARIANT=${IDIFF_VARIANT:-}
VFLAGS=${IDIFF_VARIANT_FLAGS:-}
OUT=libidiff_hip.so
if [ -n "$VFLAGS" ] && [ -z "$VARIANT" ]; then
  echo "build.sh: IDIFF_VARIANT_FLAGS without IDIFF_VARIANT -- a variant build must name its own output file" >&2; exit 4
fi
if [ -n "$VARIANT" ]; then
  case "$VARIANT" in *[!A-Za-z0-9_]*) echo "build.sh: IDIFF_VARIANT must be alphanumeric" >&2; exit 4;; esac
  OUT="libidiff_hip.$VARIANT.so"
  # (the flags as ONE shell word inside the string literal: $FLAGS is expanded unquoted below)
  FLAGS="$FLAGS $VFLAGS -DIDIFF_VARIANT_FLAGS=\"$(printf '%s' "${VFLAGS:-(none)}" | tr -d '\\"' | tr ' ' ',')\""
fi
OBJ=$(mktemp -d)
trap 'rm -rf "$OBJ"' EXIT
pids=()
for s in $SRC; do
  extra=""
  # packed-f32 VALU (v_pk_add/fma_f32) costs extra issue cycles beside MFMAs: keep the Winograd transforms scalar
  { [ "$s" = winograd ] || [ "$s" = winograd43 ] || [ "$s" = wino1d ]; } && extra="-Xclang -target-feature -Xclang -packed-fp32-ops"
  # (the host half of the compilation does not know that feature and says so: filtered)
  "$HIPCC" $FLAGS $extra -c "$s.hip" -o "$OBJ/$s.o" 2> >(grep -v "not a recognized feature for this target" >&2) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -fPIC -shared -fvisibility=hidden -o "$OUT.tmp" $(for s in $SRC; do echo "$OBJ/$s.o"; done)
# No-scratch rule: both GPU memory faults of this project were inside kernels whose register allocation had
# spilled hundreds of bytes per lane; a library with such a kernel is not produced at all.  Read from the code objects'
# metadata: .private_segment_fixed_size (bytes of scratch per lane) of every kernel.
SCRATCH_LIMIT=${IDIFF_SCRATCH_LIMIT:-64}
LLVM=${LLVM_BIN:-/opt/rocm/lib/llvm/bin}
if [ -x "$LLVM/llvm-objdump" ] && [ -x "$LLVM/llvm-readelf" ]; then
  cp "$OUT.tmp" "$OBJ/lib.so"
  (cd "$OBJ" && "$LLVM/llvm-objdump" --offloading lib.so > /dev/null)
  nk=0
  for co in "$OBJ"/lib.so*gfx950*; do
    [ -e "$co" ] || continue
    bad=$("$LLVM/llvm-readelf" --notes "$co" | awk -v lim="$SCRATCH_LIMIT" '
      /\.name:/ {name=$2}
      /\.private_segment_fixed_size:/ {n++; if ($2+0 > lim) print name ": " $2 " bytes of scratch per lane (limit " lim ")"}
      /\.uses_dynamic_stack:/ {if ($2 == "true") print name ": uses a dynamic stack (scratch sized at launch)"}
      END {print "kernels " n > "/dev/stderr"}' 2> "$OBJ/count")
    nk=$((nk + $(awk '{print $2}' "$OBJ/count")))
    if [ -n "$bad" ]; then
      echo "build.sh: kernels with a scratch segment -- refusing to produce libidiff_hip.so:" >&2
      echo "$bad" >&2
      rm -f "$OUT.tmp"
      exit 3
    fi
  done
  # every __global__ function of the objects must have been seen by the check (kernel descriptors: one .kd symbol each)
  want=$("$LLVM/llvm-readelf" --dyn-syms --wide "$OBJ"/lib.so*gfx950* | grep -c '\.kd$' || true)
  [ "$nk" -gt 0 ] && [ "$nk" -eq "$want" ] || { echo "build.sh: scratch check saw $nk kernels, the code objects hold $want" >&2; rm -f "$OUT.tmp"; exit 3; }
  echo "scratch check: $nk kernels, none above $SCRATCH_LIMIT bytes per lane, none with a dynamic stack"
elif [ "${IDIFF_SKIP_SCRATCH_CHECK:-0}" = 1 ]; then
  echo "build.sh: LLVM binutils not found under $LLVM -- scratch check SKIPPED (IDIFF_SKIP_SCRATCH_CHECK=1)" >&2
else
  echo "build.sh: LLVM binutils not found under $LLVM: the no-scratch rule cannot be checked (IDIFF_SKIP_SCRATCH_CHECK=1 to build anyway)" >&2
  rm -f "$OUT.tmp"; exit 3
fi
mv -f "$OUT.tmp" "$OUT"
echo "built $(pwd)/$OUT"
