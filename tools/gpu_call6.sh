#!/bin/bash
cd "$GRAFT_REPO_ROOT"
echo "== product"; python tools/dump_route_check.py --large-k-sweep 2>&1 | grep ms | python -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print(j['k'], j['cell'], j['n_probe'], j['ms'])"
echo "== lists"; TPQ_AMD_LIB=$PWD/torchpq_amd/variants/libtorchpq_amd_ab.so TPQ_SCAN_DUMP=0 python tools/dump_route_check.py --large-k-sweep 2>&1 | grep ms | python -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print(j['k'], j['cell'], j['n_probe'], j['ms'])"
