#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for k in 400 500; do
echo "== k=$k product"; bash tools/kstats.sh ks_$k python $PWD/tools/dump_route_check.py --one 64,2,1024,977,32,$k,10000 | head -6
echo "== k=$k lists"; TPQ_AMD_LIB=$PWD/torchpq_amd/variants/libtorchpq_amd_ab.so TPQ_SCAN_DUMP=0 bash tools/kstats.sh ksl_$k python $PWD/tools/dump_route_check.py --one 64,2,1024,977,32,$k,10000 | head -6
done
