#!/bin/bash
# Builds libtorchpq_amd.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
# Incremental: every object carries the compiler's own dependency list (build/<obj>.d, -MD), so a change to
# scan_device.h rebuilds the scan units only -- not the k-means / Lloyd units, which take the longest.
# FORCE=1 rebuilds everything; JOBS=n bounds the parallel compiles (default: the host's cores).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${HERE}/../libtorchpq_amd.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math
       -Wall -Wno-unused-function -Wno-unused-variable -DNDEBUG)
JOBS="${JOBS:-$(nproc)}"
mkdir -p "${HERE}/build"
cd "${HERE}"   # (dependency files written by hand-run compiles may hold paths relative to this directory)
# a change of flags rebuilds everything (the flags of the last build are kept next to the objects)
FLAGLINE="${FLAGS[*]} ${EXTRA_FLAGS:-}"
if [[ ! -f "${HERE}/build/flags.txt" || "$(cat "${HERE}/build/flags.txt")" != "$FLAGLINE" ]]; then FORCE=1; fi
stale() {  # object missing, or older than any file its last compile read?
  local obj="$1" src="$2" dep="${1%.o}.d" f
  [[ "${FORCE:-0}" == "1" || ! -f "$obj" || ! -f "$dep" || "$obj" -ot "$src" ]] && return 0
  # the .d file: "obj: dep dep \" lines; every existing dependency must be older than the object
  for f in $(sed -e 's/^[^:]*://' -e 's/\\$//' "$dep"); do
    [[ -f "$f" && "$obj" -ot "$f" ]] && return 0
  done
  return 1
}
cmds=()
# the scan-layout kernels: one translation unit per sub-quantizer count
for m in 64 120 128 96 56 48 40 32 28 24 20 16 12 8 4; do  # = TPQ_PACKED_M_LIST (scan_device.h), longest first
  obj="${HERE}/build/scan_packed_${m}.o"
  if stale "$obj" "${HERE}/scan_packed.hip"; then
    cmds+=("'$HIPCC' ${FLAGS[*]} -DTPQ_PACKED_M=${m} -MD -MF '${obj%.o}.d' -x hip -c '${HERE}/scan_packed.hip' -o '$obj' ${EXTRA_FLAGS:-}")
  fi
done
for src in lloyd.hip kmeans.hip assign_fast.hip select.hip scan.hip kmeans_split.hip container.hip lut.hip pack.hip ubench.hip api.cpp; do
  obj="${HERE}/build/${src%.*}.o"
  if stale "$obj" "${HERE}/${src}"; then
    cmds+=("'$HIPCC' ${FLAGS[*]} -MD -MF '${obj%.o}.d' -x hip -c '${HERE}/${src}' -o '$obj' ${EXTRA_FLAGS:-}")
  fi
done
if (( ${#cmds[@]} )); then
  echo "compiling ${#cmds[@]} unit(s), ${JOBS} at a time"
  printf '%s\n' "${cmds[@]}" | xargs -d '\n' -n 1 -P "${JOBS}" bash -c || { echo "compile failed" >&2; exit 1; }
fi
echo "$FLAGLINE" > "${HERE}/build/flags.txt"
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -Wl,-z,defs -o "$OUT" "${HERE}"/build/*.o
echo "built $OUT"
