#!/bin/bash
# Round profile of bench.py on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_bench.sh r01
# 1. plain bench line                         -> gpurun_out/<tag>/bench.json
# 2. rocprofv3 --kernel-trace --stats         -> gpurun_out/<tag>/stats/
# 3. three separate --pmc passes (HBM-side fetch, TCC, SQ/LDS) -> gpurun_out/<tag>/pmc_*/
# 4. tools/summarize_rocprof.py               -> gpurun_out/<tag>/scan_packed.json
# Copy bench.json, the kernel stats CSV and scan_packed.json into profiles/ afterwards.
set -uo pipefail
TAG="${1:-r01}"
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="${ROOT}/gpurun_out/${TAG}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python ${ROOT}/bench.py --steps 20 --warmup 3"
$BENCH > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -1 "$OUT/bench.json"
ALGO=$(python -c "import json;print(json.load(open('$OUT/bench.json'))['roofline']['algorithmic_bytes_per_launch'])")
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- $BENCH --no-cpu-baseline > "$OUT/run_stats.log" 2>&1
i=0
for ctrs in "FETCH_SIZE TCC_EA0_RDREQ_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i + 1))
  rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/pmc_$i" -o run -- $BENCH --steps 5 --no-cpu-baseline > "$OUT/run_pmc$i.log" 2>&1
done
# flatten rocprofv3's <dir>/<host>/ level
for d in "$OUT"/stats "$OUT"/pmc_*; do
  find "$d" -mindepth 2 -name "*.csv" -exec mv {} "$d"/ \; 2>/dev/null
done
python "${ROOT}/tools/summarize_rocprof.py" "$OUT" scan_packed "$ALGO" > "$OUT/scan_packed.json"
head -c 1500 "$OUT/scan_packed.json"
ls "$OUT" "$OUT/stats" | head -30
