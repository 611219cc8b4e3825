ROOT=$PWD
V=predpp
for shape in "--m 8" "--m 16" "--m 32" "--m 64" "--m 16 --n-cells 4096 --cell 244 --n-probe 32" "--m 32 --n-cells 4096 --cell 244 --n-probe 32" "--m 64 --n-cells 4096 --cell 244 --n-probe 32" "--m 64 --k 300" "--m 64 --k 1000" "--m 32 --k 300" "--m 8 --n-cells 4096 --cell 244 --n-probe 16"; do
  echo "== $shape"
  for rep in 1 2; do
  echo -n "  $V: "; TPQ_AMD_LIB=$ROOT/torchpq_amd/variants/libtorchpq_amd_$V.so python tools/scan_microbench.py $shape --layouts packed --iters 20 2>/dev/null
  echo -n "  product: "; python tools/scan_microbench.py $shape --layouts ref,packed --iters 20 --check 2>&1 | tail -1
  done
done
