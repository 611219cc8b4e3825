#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_dump_route.py -x -q 2>&1 | tail -3
fmt='
import sys,json
for l in sys.stdin:
    if not l.startswith("{"): continue
    j=json.loads(l); print(j["k"], j["cell"], j["n_probe"], j["nq"], j["ms"], j.get("equal"))'
echo "== product"; python tools/dump_route_check.py --large-k 2>&1 | python -c "$fmt"
python tools/dump_route_check.py --large-k-sweep 2>&1 | python -c "$fmt"
for k in 110 120 128 240 248 420 440 460; do python tools/dump_route_check.py --one 64,2,1024,977,32,$k,10000 2>&1 | python -c "$fmt"; done
echo "== lists / one-launch finish"
for k in 110 120 128 240 248 420 440 460; do TPQ_AMD_LIB=$PWD/torchpq_amd/variants/libtorchpq_amd_ab.so TPQ_SCAN_DUMP=0 python tools/dump_route_check.py --one 64,2,1024,977,32,$k,10000 2>&1 | python -c "$fmt"; done
bash tools/kstats.sh ks_400 python $PWD/tools/dump_route_check.py --one 64,2,1024,977,32,400,10000 | head -4
