ROOT=/root/repo
for shape in "--m 64 --n-cells 1024 --cell 977 --n-probe 32 --k 100" "--m 64 --n-cells 4096 --cell 244 --n-probe 16 --k 100" "--m 64 --n-cells 4096 --cell 244 --n-probe 8 --k 100" "--m 64 --n-cells 4096 --cell 244 --n-probe 64 --k 100" "--m 64 --n-cells 16384 --cell 61 --n-probe 32 --k 100" "--m 64 --n-cells 16384 --cell 61 --n-probe 128 --k 100" "--m 64 --n-cells 16384 --cell 6103 --n-probe 64 --k 100 --nq 4000"; do
  echo "== $shape"
  echo -n "  product: "; python $ROOT/tools/scan_microbench.py $shape --layouts packed --iters 20 2>/dev/null
  for v in "$@"; do
    echo -n "  $v: "; TPQ_AMD_LIB=$ROOT/torchpq_amd/variants/libtorchpq_amd_$v.so python $ROOT/tools/scan_microbench.py $shape --layouts packed --iters 20 2>/dev/null
  done
done
