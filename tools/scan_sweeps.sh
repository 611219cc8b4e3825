#!/bin/bash
# The m, k and n_probe sweeps DESIGN 4.0 quotes, as one JSON file (run through gpurun from the repo root):
#   bash tools/scan_sweeps.sh > gpurun_out/scan_sweeps.json
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
echo "{"
echo "\"what\": \"tools/scan_microbench.py, 10 000 queries x 32 probes of 977 slots (C2 shape), scan + merge, 20 iterations; --check: packed == reference-layout kernel bit for bit\","
echo "\"m_sweep\": {"
first=1
for m in 4 8 12 16 20 24 28 32 40 48 56 64 96 120 128; do
  [ $first -eq 1 ] || echo ","
  first=0
  echo -n "\"$m\": $(python "$ROOT/tools/scan_microbench.py" --m $m --layouts packed --iters 20)"
done
echo "},"
echo "\"k_sweep_m64\": {"
first=1
for k in 10 100 200 300 500 1000; do
  [ $first -eq 1 ] || echo ","
  first=0
  echo -n "\"$k\": $(python "$ROOT/tools/scan_microbench.py" --m 64 --k $k --layouts packed,ref --iters 10 --check)"
done
echo "},"
echo "\"n_probe_sweep_m64\": {"
first=1
for np in 4 8 16 32 64 128; do
  [ $first -eq 1 ] || echo ","
  first=0
  echo -n "\"$np\": $(python "$ROOT/tools/scan_microbench.py" --m 64 --n-probe $np --layouts packed --iters 20)"
done
echo "}"
echo "}"
