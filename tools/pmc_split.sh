#!/bin/bash
# PMC passes over tools/assign_split_time.py (run through gpurun from the repo root)
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="${ROOT}/gpurun_out/pmc_split"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/p$i" -o run -- python "${ROOT}/tools/assign_split_time.py" > "$OUT/log$i.txt" 2>&1
  find "$OUT/p$i" -mindepth 2 -name "*.csv" -exec mv {} "$OUT/p$i"/ \; 2>/dev/null
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/p*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "max_sim_split" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k, "launches", len(v), "mean", sum(v) / len(v))
PY
