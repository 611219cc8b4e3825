#!/bin/bash
# Same-box A/B of the slots per lane and tile (scan_device.h packed_slots): product against a -DTPQ_SLOTS_LOG2 variant.
#   bash tools/ab_slots.sh s2 "12 16 20 24 28 40 48 56"
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
V="$1"
for m in $2; do
  for extra in "" "--n-cells 4096 --cell 244 --n-probe 32" "--k 1"; do
    echo "== --m $m $extra"
    echo -n "  product: "; python "$ROOT/tools/scan_microbench.py" --m $m $extra --layouts packed --iters 20 2>/dev/null
    echo -n "  $V: "; TPQ_AMD_LIB="$ROOT/torchpq_amd/variants/libtorchpq_amd_$V.so" python "$ROOT/tools/scan_microbench.py" --m $m $extra --layouts ref,packed --iters 20 --check 2>&1 | tail -1
  done
done
