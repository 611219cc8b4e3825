#!/bin/bash
# Build an experimental variant of libtorchpq_amd.so: the named translation units recompiled with extra
# flags (and -DTPQ_AB_SWITCHES: the environment A/B switches exist only in variants, never in the product
# library), everything else taken from csrc/build/.  The variant is loaded with TPQ_AMD_LIB=<path>.
#   tools/build_variant.sh <name> "<extra flags>" <unit>...
#     unit = object basename: scan, lloyd, select, kmeans, assign_fast, ..., or scan_packed_<M>
#   e.g. tools/build_variant.sh prof "-DTPQ_SCAN_PROFILE" scan scan_packed_64 scan_packed_32
# -> torchpq_amd/variants/libtorchpq_amd_<name>.so
set -euo pipefail
NAME="$1"; EXTRA="${2:-}"; shift 2
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
CS="${ROOT}/torchpq_amd/csrc"
SRC_DIR="${SRC_DIR:-$CS}"   # (sources of the recompiled units; another directory = an older copy of the kernels for A/B)
OUT="${ROOT}/torchpq_amd/variants"
mkdir -p "$OUT/obj_${NAME}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function
       -DNDEBUG -DTPQ_AB_SWITCHES)
pids=()
for unit in "$@"; do
  if [[ "$unit" == scan_packed_* ]]; then
    src="scan_packed.hip"; def="-DTPQ_PACKED_M=${unit#scan_packed_}"
  elif [[ -f "${SRC_DIR}/${unit}.hip" ]]; then
    src="${unit}.hip"; def=""
  else
    src="${unit}.cpp"; def=""
  fi
  ( "$HIPCC" "${FLAGS[@]}" $EXTRA $def -x hip -c "${SRC_DIR}/${src}" -o "$OUT/obj_${NAME}/${unit}.o" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p" || { echo "compile failed" >&2; exit 1; }; done
OBJS=()
for o in "${CS}"/build/*.o; do
  [[ -f "$OUT/obj_${NAME}/$(basename "$o")" ]] || OBJS+=("$o")
done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT/libtorchpq_amd_${NAME}.so" "${OBJS[@]}" "$OUT/obj_${NAME}"/*.o
echo "built $OUT/libtorchpq_amd_${NAME}.so"
