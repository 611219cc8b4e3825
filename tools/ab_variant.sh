#!/bin/bash
# Same-box A/B of the list scan, product library against a variant, each shape checked against the reference-layout kernel:
#   bash tools/ab_variant.sh <variant> "<m values>" [<baseline variant> instead of the product]
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
V="$1"
B="${3:-}"
BL=""; BN="product"
if [ -n "$B" ]; then BL="$ROOT/torchpq_amd/variants/libtorchpq_amd_$B.so"; BN="$B"; fi
for m in $2; do
  for extra in "" "--n-cells 4096 --cell 244 --n-probe 32" "--k 1" "--n-cells 16384 --cell 61 --n-probe 32"; do
    echo "== --m $m $extra"
    for rep in 1 2; do
      echo -n "  $BN: "; TPQ_AMD_LIB="$BL" python "$ROOT/tools/scan_microbench.py" --m $m $extra --layouts packed --iters 20 2>/dev/null
      echo -n "  $V: "; TPQ_AMD_LIB="$ROOT/torchpq_amd/variants/libtorchpq_amd_$V.so" python "$ROOT/tools/scan_microbench.py" --m $m $extra --layouts ref,packed --iters 20 --check 2>&1 | tail -1
    done
  done
done
