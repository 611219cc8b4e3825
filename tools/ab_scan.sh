#!/bin/bash
# Same-box A/B of the list scan: the product library against one or more variants (tools/build_variant.sh), on the
# synthetic uniform indexes of tools/scan_microbench.py at the shapes of the reference's grid and of BASELINE.json.
#   bash tools/ab_scan.sh oldwalk [more variants...]
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
SHAPES=(
  "--m 64 --n-cells 1024 --cell 977 --n-probe 32 --k 100"
  "--m 64 --n-cells 4096 --cell 244 --n-probe 128 --k 1"
  "--m 64 --n-cells 4096 --cell 244 --n-probe 16 --k 100"
  "--m 64 --n-cells 16384 --cell 61 --n-probe 128 --k 1"
  "--m 64 --n-cells 16384 --cell 61 --n-probe 16 --k 100"
  "--m 32 --n-cells 1024 --cell 977 --n-probe 32 --k 100"
  "--m 32 --n-cells 4096 --cell 244 --n-probe 128 --k 1"
  "--m 32 --n-cells 4096 --cell 244 --n-probe 32 --k 100"
  "--m 32 --n-cells 16384 --cell 61 --n-probe 128 --k 1"
  "--m 16 --n-cells 1024 --cell 977 --n-probe 32 --k 100"
  "--m 16 --n-cells 4096 --cell 244 --n-probe 128 --k 1"
  "--m 16 --n-cells 16384 --cell 61 --n-probe 128 --k 100"
  "--m 8 --n-cells 1024 --cell 977 --n-probe 32 --k 100"
  "--m 8 --n-cells 4096 --cell 244 --n-probe 128 --k 1"
  "--m 8 --n-cells 16384 --cell 61 --n-probe 128 --k 100"
)
for shape in "${SHAPES[@]}"; do
  echo "== $shape"
  echo -n "  product: "; python "$ROOT/tools/scan_microbench.py" $shape --layouts packed --iters 20 2>/dev/null
  for v in "$@"; do
    echo -n "  $v: "; TPQ_AMD_LIB="$ROOT/torchpq_amd/variants/libtorchpq_amd_$v.so" python "$ROOT/tools/scan_microbench.py" $shape --layouts packed --iters 20 2>/dev/null
  done
done
