#!/bin/bash
# PMC passes over the k-means update kernel (C5 shape)
cd /tmp && export TMPDIR=/tmp
ROOT=/root/repo
OUT=$ROOT/gpurun_out/um_prof
mkdir -p $OUT
i=0
for ctrs in "FETCH_SIZE TCC_EA0_RDREQ_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d $OUT/p$i -o run -- python $ROOT/tools/kmeans_microbench.py --iters 2 > $OUT/log$i.txt 2>&1
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "centroid_accum_mfma" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, len(v), sum(v) / len(v))
PY
  [ -z "$f" ] && tail -3 $OUT/log$i.txt
done
