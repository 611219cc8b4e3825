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
echo "== lists"; TPQ_AMD_LIB=$PWD/torchpq_amd/variants/libtorchpq_amd_ab.so TPQ_SCAN_DUMP=0 python tools/dump_route_check.py --large-k --no-check 2>&1 | python -c "$fmt"
