#!/bin/bash
# Round profile of bench.py on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_bench.sh r03 [c2 c3 c4 c5 wide]
# per workload W (c2 = the headline bench line; c3/c4/c5/wide = `bench.py --secondary-only W`):
#   1. plain run                                 -> gpurun_out/<tag>/<W>/bench.json
#   2. rocprofv3 --kernel-trace --stats          -> gpurun_out/<tag>/<W>/stats/
#   3. separate --pmc passes (HBM-side fetch, TCC, SQ/LDS) -> gpurun_out/<tag>/<W>/pmc_*/
#   4. tools/summarize_rocprof.py                -> gpurun_out/<tag>/<W>/summary.json
# Copy bench.json, the kernel-stats CSV and summary.json into profiles/ afterwards
# (tools/collect_profiles.py <tag> does it).
set -uo pipefail
TAG="${1:-r03}"
shift || true
WORKLOADS="${*:-c2 c3 c4 c5}"
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
for W in $WORKLOADS; do
  OUT="${ROOT}/gpurun_out/${TAG}/${W}"
  mkdir -p "$OUT"
  case "$W" in
    c2) BENCH="python ${ROOT}/bench.py --steps 20 --warmup 3 --no-secondary --no-traffic-pass"; KERNEL="scan_packed_kernel";;
    c3) BENCH="python ${ROOT}/bench.py --secondary-only c3 --no-traffic-pass"; KERNEL="scan_packed_kernel";;
    c4) BENCH="python ${ROOT}/bench.py --secondary-only c4 --c4-variant warm --no-traffic-pass"; KERNEL="scan_packed_kernel";;
    c4cold) BENCH="python ${ROOT}/bench.py --secondary-only c4 --c4-variant cold --no-traffic-pass"; KERNEL="scan_packed_kernel";;
    c5) BENCH="python ${ROOT}/bench.py --secondary-only c5 --no-traffic-pass"; KERNEL="coarse_kernel";;
    wide) BENCH="python ${ROOT}/bench.py --secondary-only wide --no-traffic-pass"; KERNEL="gemm_kernel";;
  esac
  $BENCH > "$OUT/bench.json" 2> "$OUT/bench.err"
  tail -c 600 "$OUT/bench.json"; echo
  ALGO=$(python - "$OUT/bench.json" "$W" <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
w = sys.argv[2]
r = j["roofline"] if w == "c2" else (j["secondary"]["c4"]["cold"]["roofline"] if w == "c4cold" else j["secondary"][w]["roofline"])
print(r.get("algorithmic_bytes_per_launch", r.get("algorithmic_flops_per_launch", 0)))
PY
)
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- $BENCH --no-cpu-baseline > "$OUT/run_stats.log" 2>&1
  i=0
  PASSES=("FETCH_SIZE TCC_EA0_RDREQ_sum")   # c3/c4/c5: HBM-side bytes only (each pass re-builds the workload)
  [[ "$W" == "c2" ]] && PASSES+=("TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE")
  # the k-means kernels (VERDICT r2 #1/#7a): matrix-pipe / VALU / wait shares and the effective clock, per kernel
  [[ "$W" == "c5" || "$W" == "wide" ]] && PASSES+=("GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
                                  "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
                                  "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM")
  for ctrs in "${PASSES[@]}"; do
    i=$((i + 1))
    STEPS=""; [[ "$W" == "c2" ]] && STEPS="--steps 5"
    rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/pmc_$i" -o run -- $BENCH $STEPS --no-cpu-baseline > "$OUT/run_pmc$i.log" 2>&1
  done
  # flatten rocprofv3's <dir>/<host>/ level
  for d in "$OUT"/stats "$OUT"/pmc_*; do
    find "$d" -mindepth 2 -name "*.csv" -exec mv {} "$d"/ \; 2>/dev/null
  done
  python "${ROOT}/tools/summarize_rocprof.py" "$OUT" "$KERNEL" "$ALGO" > "$OUT/summary.json"
  head -c 1200 "$OUT/summary.json"; echo
done
