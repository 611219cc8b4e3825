#!/bin/bash
# rocprofv3 --pmc passes (one per counter group, counters only: no tracing) over one command; prints
# the per-launch mean of every counter for kernels whose name contains <kernel substring>.
#   bash tools/pmc.sh <out dir under gpurun_out> <kernel substring> "<ctr ctr ...;ctr ctr ...>" <command ...>
set -uo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="${ROOT}/gpurun_out/$1"; KERNEL="$2"; GROUPS_="$3"; shift 3
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
IFS=';' read -ra GR <<< "$GROUPS_"
for ctrs in "${GR[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/p$i" -o run -- "$@" > "$OUT/log$i.txt" 2>&1
done
python - "$OUT" "$KERNEL" <<'PY'
import csv, glob, sys, collections, json, os
out, kern = sys.argv[1], sys.argv[2]
res = {}
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kern in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[k] = {"launches": len(v), "mean": sum(v) / len(v)}
        print(f"{k:32s} launches {len(v):3d} mean {sum(v)/len(v):.6g}")
json.dump({"kernel": kern, "pmc": res}, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
PY
