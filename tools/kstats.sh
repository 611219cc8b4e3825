#!/bin/bash
# rocprofv3 --kernel-trace --stats of one command; prints the top kernels.
#   bash tools/kstats.sh <out dir under gpurun_out> <command ...>
set -uo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="${ROOT}/gpurun_out/$1"; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- "$@" > "$OUT/run.log" 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys, os
for f in glob.glob(os.path.join(sys.argv[1], "stats", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print(f"{float(r['AverageNs'])/1e3:10.1f} us x{int(r['Calls']):4d} {float(r['Percentage']):6.2f}%  {r['Name'][:100]}")
PY
tail -3 "$OUT/run.log"
