#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TPQ_AMD_LIB=$PWD/torchpq_amd/variants/libtorchpq_amd_ab.so
fmt='import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j["config"]["nq"], j["config"]["n_cells"], j["config"]["n_probe"], "scan", j["scan_ms"], "total", j["total_ms"])'
for shape in "--preset c2" "--preset c2 --n-cells 4096 --cell 244 --n-probe 16" "--preset c2 --n-cells 16384 --cell 61 --n-probe 32"; do
for nq in 128 256 384 512 768 1000; do
  echo -n "lists/fused  "; python tools/search_breakdown.py $shape --nq $nq --iters 30 2>/dev/null | python -c "$fmt"
  echo -n "dump minq=64 "; TPQ_SCAN_DUMP_MINQ=64 python tools/search_breakdown.py $shape --nq $nq --iters 30 2>/dev/null | python -c "$fmt"
done
done
