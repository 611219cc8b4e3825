#!/bin/bash
# same-box A/B of variant libraries on the C5 record (bench.py --secondary-only c5):
#   bash tools/ab_c5.sh updpf0 updpf2 updpf4     (names of torchpq_amd/variants/libtorchpq_amd_<name>.so; "main" = the product)
cd "$(dirname "${BASH_SOURCE[0]}")/.."
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" == "main" ]; then unset TPQ_AMD_LIB; else export TPQ_AMD_LIB="$PWD/torchpq_amd/variants/libtorchpq_amd_$v.so"; fi
  python bench.py --secondary-only c5 --no-traffic-pass 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['secondary']['c5']
print('$v', 'rep$rep', {k:j.get(k) for k in ('iter_ms','assign_ms','update_ms_derived','assign_labels_equal_to_fp32_kernel','new_centroids_max_rel_diff_vs_tpq_compute_centroids','error')})"
done
done
