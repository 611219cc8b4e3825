#!/bin/bash
# Build an experimental variant of libtorchpq_amd.so: one translation unit recompiled with extra
# flags, everything else taken from csrc/build/.  The variant is loaded with TPQ_AMD_LIB=<path>.
#   tools/build_variant.sh <name> <source file in csrc> "<extra flags>" [object name]
# -> torchpq_amd/variants/libtorchpq_amd_<name>.so
set -euo pipefail
NAME="$1"; SRC="$2"; EXTRA="${3:-}"; OBJNAME="${4:-${SRC%.*}}"
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
CS="${ROOT}/torchpq_amd/csrc"
OUT="${ROOT}/torchpq_amd/variants"
mkdir -p "$OUT/obj_${NAME}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -DNDEBUG)
"$HIPCC" "${FLAGS[@]}" $EXTRA -x hip -c "${CS}/${SRC}" -o "$OUT/obj_${NAME}/${OBJNAME}.o"
OBJS=()
for o in "${CS}"/build/*.o; do
  [[ "$(basename "$o")" == "${OBJNAME}.o" ]] || OBJS+=("$o")
done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT/libtorchpq_amd_${NAME}.so" "${OBJS[@]}" "$OUT/obj_${NAME}/${OBJNAME}.o"
echo "built $OUT/libtorchpq_amd_${NAME}.so"
