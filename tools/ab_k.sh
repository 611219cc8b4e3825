# Same-box A/B of the large-k sorted-list kernels at m = 64 (tools/scan_microbench.py): bash tools/ab_k.sh <variant>
ROOT=$PWD
for k in 100 200 300 500; do
  echo "== --m 64 --k $k"
  for rep in 1 2; do
  echo -n "  product: "; python tools/scan_microbench.py --m 64 --k $k --layouts packed --iters 20 2>/dev/null
  echo -n "  oa: "; TPQ_AMD_LIB=$ROOT/torchpq_amd/variants/libtorchpq_amd_oa.so python tools/scan_microbench.py --m 64 --k $k --layouts packed --iters 20 2>/dev/null
  done
done
