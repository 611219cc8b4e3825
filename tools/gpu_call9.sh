#!/bin/bash
cd "$GRAFT_REPO_ROOT"
fmt='
import sys,json
for l in sys.stdin:
    if not l.startswith("{"): continue
    j=json.loads(l); print(j["k"], j["cell"], j["n_probe"], j["nq"], j["ms"], j.get("equal"))'
for k in 420 448 470 490 504; do
for shape in 4096,244,32 1024,977,16; do
echo -n "product "; python tools/dump_route_check.py --one 64,2,$shape,$k,10000 2>&1 | python -c "$fmt"
echo -n "lists   "; TPQ_AMD_LIB=$PWD/torchpq_amd/variants/libtorchpq_amd_ab.so TPQ_SCAN_DUMP=0 python tools/dump_route_check.py --one 64,2,$shape,$k,10000 2>&1 | python -c "$fmt"
done
done
