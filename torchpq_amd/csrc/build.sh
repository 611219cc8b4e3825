#!/bin/bash
# Builds libtorchpq_amd.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${HERE}/../libtorchpq_amd.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math
       -Wall -Wno-unused-function -DNDEBUG)
mkdir -p "${HERE}/build"
pids=()
stale() {  # object older than its source or any shared header?
  local obj="$1" src="$2"
  [[ ! -f "$obj" || "$obj" -ot "$src" || "$obj" -ot "${HERE}/common.h" \
     || "$obj" -ot "${HERE}/wave_topk.h" || "$obj" -ot "${HERE}/scan_layout.h" \
     || "$obj" -ot "${HERE}/scan_device.h" || "$obj" -ot "${HERE}/probe_fast.h" \
     || "$obj" -ot "${HERE}/../../include/torchpq_amd.h" || "${FORCE:-0}" == "1" ]]
}
# the scan-layout kernels: one translation unit per sub-quantizer count
for m in 4 8 12 16 20 24 28 32 40 48 56 64 96 120 128; do  # = TPQ_PACKED_M_LIST (scan_device.h)
  obj="${HERE}/build/scan_packed_${m}.o"
  if stale "$obj" "${HERE}/scan_packed.hip"; then
    ( "$HIPCC" "${FLAGS[@]}" -DTPQ_PACKED_M=${m} -x hip -c "${HERE}/scan_packed.hip" -o "$obj" ${EXTRA_FLAGS:-} ) &
    pids+=($!)
  fi
done
for src in api.cpp scan.hip pack.hip select.hip lut.hip kmeans.hip kmeans_split.hip assign_fast.hip lloyd.hip container.hip ubench.hip; do
  obj="${HERE}/build/${src%.*}.o"
  if stale "$obj" "${HERE}/${src}"; then
    ( "$HIPCC" "${FLAGS[@]}" -x hip -c "${HERE}/${src}" -o "$obj" ${EXTRA_FLAGS:-} ) &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [[ -n "$p" ]] && { wait "$p" || { echo "compile failed" >&2; exit 1; }; }; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -Wl,-z,defs -o "$OUT" "${HERE}"/build/*.o
echo "built $OUT"
