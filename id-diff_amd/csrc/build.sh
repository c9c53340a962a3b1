#!/bin/bash
# Build libidiff_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
SRC="runtime.hip upfirdn2d.hip fused_bias_act.hip igemm.hip norm_act.hip rng.hip spectrum.hip winograd.hip"
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden \
  -Wall -Wno-unused-function -o libidiff_hip.so.tmp $SRC
mv -f libidiff_hip.so.tmp libidiff_hip.so
echo "built $(pwd)/libidiff_hip.so"
