#!/bin/bash
# round 5, call 1: the GPU suite, then the driver's own bench command three times
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05_pytest_gpu.log 2>&1
echo "pytest rc $?" >> gpurun_out/r05_pytest_gpu.log
tail -5 gpurun_out/r05_pytest_gpu.log
for i in 1 2 3; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_full_$i.json 2> gpurun_out/r05_bench_full_$i.err
  echo "bench $i rc $?"
  python - gpurun_out/r05_bench_full_$i.json <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = j["roofline"]
    print("headline", j["value"], j["ms_per_step"], r["kernel_ms"], r["frac"], r.get("queries_redone_exactly"), r.get("profile_mismatch"))
    for k, v in j.get("secondary", {}).items():
        rf = v.get("roofline", {})
        print(k, v.get("value"), v.get("ms_per_step"), rf.get("kernel_ms"), rf.get("frac"), rf.get("queries_redone_exactly"), rf.get("n_split"), v.get("error"), v.get("skipped"), v.get("iter_ms"), v.get("ms"))
except Exception as e:
    print("parse failed", e)
PY
done
