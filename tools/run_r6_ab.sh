V=$PWD/torchpq_amd/variants
for rep in 1 2; do
echo "== product"; python tools/ab_stream.py 2>/dev/null | grep '^{'
echo "== d3"; TPQ_AMD_LIB=$V/libtorchpq_amd_d3.so python tools/ab_stream.py 2>/dev/null | grep '^{'
done
for sh in 64,2,16384,61,32,100,10000 64,2,4096,244,32,100,10000 64,2,4096,244,128,100,10000; do
  a=$(python tools/dump_route_check.py --one $sh 2>/dev/null | grep '^{' | python -c "import json,sys; r=json.load(sys.stdin); print(r['ms'], r['GBps'])")
  b=$(TPQ_AMD_LIB=$V/libtorchpq_amd_d3.so python tools/dump_route_check.py --one $sh 2>/dev/null | grep '^{' | python -c "import json,sys; r=json.load(sys.stdin); print(r['ms'], r['GBps'])")
  echo "$sh  depth2: $a   depth3: $b"
done
