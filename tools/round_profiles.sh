#!/bin/bash
# Everything DESIGN section 4.0 quotes for a round, in one gpurun call (writes under gpurun_out/<tag>_*; copy into profiles/
# with tools/collect_profiles.py <tag> + the cp lines at the end of this script's output):
#   bash tools/round_profiles.sh r04
TAG="${1:-r04}"
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$ROOT/gpurun_out"
cd "$ROOT"
bash tools/profile_bench.sh "$TAG" c2 c3 c4 c5 wide > "$OUT/${TAG}_profile_bench.log" 2>&1
bash tools/scan_sweeps.sh > "$OUT/${TAG}_scan_sweeps.json" 2> /dev/null
bash tools/batch_sweep.sh > "$OUT/${TAG}_batch_sweep.json" 2> /dev/null
python tools/reference_grid.py --out "$OUT/${TAG}_reference_grid.json" > "$OUT/${TAG}_reference_grid.log" 2>&1
python tools/probe_bench.py --n-cells 1024,4096,16384 --n-probe 1,16,32,64,128 > "$OUT/${TAG}_probe_routes.jsonl" 2> /dev/null
{
  echo "{\"what\": \"tools/search_breakdown.py --preset c2 --nq N --iters 50: stream time of search() on the C2-shaped index (ms)\", \"rows\": ["
  first=1
  for nq in 1 16 256; do
    [ $first -eq 1 ] || echo ","
    first=0
    echo -n "$(python tools/search_breakdown.py --preset c2 --nq $nq --iters 50 2>/dev/null | tail -1)"
  done
  echo "]}"
} > "$OUT/${TAG}_small_batches.json"
python tools/selection_soak.py --mode probe --cases 240 --seed 11 --big > "$OUT/${TAG}_soak_probe.json" 2>/dev/null
python tools/selection_soak.py --mode cascade --cases 120 --seed 12 > "$OUT/${TAG}_soak_cascade.json" 2>/dev/null
python tools/build_100m.py > "$OUT/${TAG}_build_100m.json" 2> /dev/null
tail -3 "$OUT/${TAG}_profile_bench.log"
ls -la "$OUT" | grep "${TAG}_"
