mkdir -p gpurun_out
V=$PWD/torchpq_amd/variants
(timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_dump_route.py tests/test_gpu_scan_soak.py tests/test_gpu_round4.py -x -q 2>&1 | tail -3)
echo "== large k at the C2 shape (product)"; python tools/dump_route_check.py --large-k 2>/dev/null | grep '^{' | cut -c1-230
echo "== variant, TPQ_SCAN_DUMP=1 (four waves or the lists: round 5's rule for k in (440, 504])"
TPQ_AMD_LIB=$V/libtorchpq_amd_ab.so TPQ_SCAN_DUMP=1 python tools/dump_route_check.py --one 64,2,1024,977,32,500,10000 2>/dev/null | grep '^{' | cut -c1-230
for sh in 32,4,1024,977,32,300,10000 32,4,1024,977,32,500,10000 32,4,4096,244,32,300,10000 32,4,4096,244,32,500,10000 16,2,4096,244,32,400,10000; do
  echo "== $sh pools / dump_f32"
  python tools/dump_route_check.py --one $sh 2>/dev/null | grep '^{' | cut -c1-200
  TPQ_AMD_LIB=$V/libtorchpq_amd_ab.so TPQ_SCAN_DUMP_SHORT_K=504 python tools/dump_route_check.py --one $sh 2>/dev/null | grep '^{' | cut -c1-200
done
