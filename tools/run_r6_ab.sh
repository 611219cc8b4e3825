mkdir -p gpurun_out
V=$PWD/torchpq_amd/variants
for lib in product d3 nt d3nt; do
  if [ $lib = product ]; then timeout 300 python tools/ab_stream.py >> gpurun_out/ab_stream.jsonl 2>gpurun_out/ab_stream.err;
  else TPQ_AMD_LIB=$V/libtorchpq_amd_$lib.so timeout 300 python tools/ab_stream.py >> gpurun_out/ab_stream.jsonl 2>>gpurun_out/ab_stream.err; fi
done
timeout 900 python tools/reference_grid.py --m 32,16,8 --repeats 1 --out gpurun_out/grid_new.json > gpurun_out/grid_new.log 2>&1
TPQ_AMD_LIB=$V/libtorchpq_amd_ab.so TPQ_SCAN_DUMP_SHORT_K=0 timeout 900 python tools/reference_grid.py --m 32,16,8 --repeats 1 --no-check --out gpurun_out/grid_old.json > gpurun_out/grid_old.log 2>&1
tail -n 3 gpurun_out/grid_new.log gpurun_out/grid_old.log; cat gpurun_out/ab_stream.jsonl; tail -n 5 gpurun_out/ab_stream.err
