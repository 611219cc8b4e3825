#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dump_route.py -x -q 2>&1 | tail -3
echo "== product (route auto)"; python tools/dump_route_check.py --large-k 2>&1 | cut -c1-260 | tee $O/r05_large_k_product.log
echo "== variant TPQ_SCAN_DUMP=0 (the sorted lists)"; TPQ_AMD_LIB=$PWD/torchpq_amd/variants/libtorchpq_amd_ab.so TPQ_SCAN_DUMP=0 python tools/dump_route_check.py --large-k --no-check 2>&1 | cut -c1-260 | tee $O/r05_large_k_lists.log
