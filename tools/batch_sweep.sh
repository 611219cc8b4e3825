#!/bin/bash
# search() end to end against the batch size, at the C2 and C4 shapes (tools/search_breakdown.py, synthetic uniform
# indexes): what a 10 000-query batch split over 2 / 4 / 8 GPUs (strong scaling) will do on each of them.
#   bash tools/batch_sweep.sh > gpurun_out/r04_batch_sweep.json
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
echo "{"
echo "\"what\": \"tools/search_breakdown.py: search() over nq queries, stream time per call (10 calls); 1 250 = a 10 000-query batch over 8 GPUs\","
for preset in c2 c4; do
  echo "\"$preset\": {"
  first=1
  for nq in 256 512 1250 2500 5000 10000; do
    [ $first -eq 1 ] || echo ","
    first=0
    echo -n "\"$nq\": $(python "$ROOT/tools/search_breakdown.py" --preset $preset --nq $nq 2>/dev/null | tail -1)"
  done
  if [ "$preset" == "c2" ]; then echo "},"; else echo "}"; fi
done
echo "}"
