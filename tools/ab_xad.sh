#!/bin/bash
# Same-box A/B of the short-code look-up address (round 6: two VALU per look-up in the blocks below 64): the product
# library against a variant, caller-supplied tables, every shape checked against the reference-layout kernel.
#   bash tools/ab_xad.sh xad > gpurun_out/ab_xad.txt
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
V="${1:-xad}"
SHAPES=(
  "--m 4" "--m 8" "--m 12" "--m 16" "--m 20" "--m 24" "--m 28" "--m 32" "--m 40" "--m 48" "--m 56" "--m 64" "--m 96" "--m 120" "--m 128"
  "--m 64 --n-cells 4096 --cell 244 --n-probe 32 --k 100"
  "--m 64 --n-cells 16384 --cell 61 --n-probe 32 --k 100"
  "--m 8 --k 1" "--m 16 --k 1" "--m 32 --k 1"
  "--m 32 --n-cells 4096 --cell 244 --n-probe 32 --k 100"
  "--m 32 --n-cells 4096 --cell 244 --n-probe 128 --k 1"
  "--m 32 --n-cells 16384 --cell 61 --n-probe 128 --k 100"
  "--m 16 --n-cells 4096 --cell 244 --n-probe 32 --k 100"
  "--m 8 --n-cells 4096 --cell 244 --n-probe 128 --k 10"
  "--m 8 --k 300" "--m 32 --k 300" "--m 32 --k 1000"
)
for shape in "${SHAPES[@]}"; do
  echo "== $shape"
  echo -n "  product: "; python "$ROOT/tools/scan_microbench.py" $shape --layouts packed --iters 20 2>/dev/null
  echo -n "  $V: "; TPQ_AMD_LIB="$ROOT/torchpq_amd/variants/libtorchpq_amd_$V.so" python "$ROOT/tools/scan_microbench.py" $shape --layouts ref,packed --iters 20 --check 2>&1 | tail -1
done
