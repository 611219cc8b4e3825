#!/bin/bash
# round 5, call 2: dump-route tests, the round-4 tree on the driver's command (C3 bisect), sweeps, the reference grid
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dump_route.py tests/test_gpu_fullsize.py -x -q > $O/r05_pytest_dump.log 2>&1
echo "pytest rc $?" | tee -a $O/r05_pytest_dump.log
tail -4 $O/r05_pytest_dump.log
if [ -d .r04tree ]; then
  for i in 1 2; do
    (cd .r04tree && timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > ../$O/r04tree_bench_full_$i.json 2> ../$O/r04tree_bench_full_$i.err)
    python - $O/r04tree_bench_full_$i.json <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("r04tree headline", j["value"], j["roofline"]["kernel_ms"], j["roofline"]["frac"])
    for k, v in j.get("secondary", {}).items():
        rf = v.get("roofline", {})
        print("r04tree", k, v.get("value"), rf.get("kernel_ms"), rf.get("frac"), v.get("error"))
except Exception as e:
    print("parse failed", e)
PY
  done
fi
timeout 600 bash tools/batch_sweep.sh > $O/r05_batch_sweep.json 2> /dev/null
python -c "
import json; j=json.load(open('$O/r05_batch_sweep.json'))
for p in ('c2','c4'):
    print(p, {k:(v.get('queries_per_s') or v) for k,v in j[p].items()})
" 2>&1 | cut -c1-600
timeout 900 python tools/reference_grid.py --out $O/r05_reference_grid.json > $O/r05_reference_grid.log 2>&1
echo "grid rc $?"
python -c "
import json; j=json.load(open('$O/r05_reference_grid.json')); print(json.dumps(j['summary']))"
timeout 600 bash tools/scan_sweeps.sh > $O/r05_scan_sweeps.json 2> /dev/null
echo "sweeps rc $?"
