#!/bin/bash
# Same-box A/B of search() end to end (fused table: the routes the bench times) -- product against variants:
#   bash tools/ab_fused.sh "old two" > gpurun_out/ab_fused.txt
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
for rep in 1 2 3; do
for preset in c2 c4; do
  for nq in 10000 1250; do
    echo "== $preset nq=$nq"
    echo -n "  product: "; python "$ROOT/tools/search_breakdown.py" --preset $preset --nq $nq 2>/dev/null | tail -1
    for v in $1; do
      echo -n "  $v: "; TPQ_AMD_LIB="$ROOT/torchpq_amd/variants/libtorchpq_amd_$v.so" python "$ROOT/tools/search_breakdown.py" --preset $preset --nq $nq 2>/dev/null | tail -1
    done
  done
done
done
