#!/bin/bash
# Same-box A/B of the large-batch route of the m = 64 scan (dump mode + finish kernel) against the one-launch finish
# (variant library with the A/B switches: tools/build_variant.sh ab "" scan), on tools/dump_route_check.py's shapes;
# then the per-kernel times of the first four shapes.
cd "$(dirname "${BASH_SOURCE[0]}")/.."
ROOT="$(pwd)"
echo "== product (auto route)"; python tools/dump_route_check.py 2>&1 | grep "fused\": true" | cut -c1-250
echo "== variant TPQ_SCAN_DUMP=0"; TPQ_AMD_LIB=$ROOT/torchpq_amd/variants/libtorchpq_amd_ab.so TPQ_SCAN_DUMP=0 python tools/dump_route_check.py --no-check 2>&1 | grep "fused\": true" | cut -c1-250
bash tools/kstats.sh ks4 python $ROOT/tools/dump_route_check.py --quick --no-check | head -6
