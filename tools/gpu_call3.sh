#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O
bash tools/ab_c5.sh updpf0 updpf2 updpf4 2>&1 | tee $O/r05_ab_c5_updpf.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_pmc.json 2> $O/r05_bench_pmc.err
echo "bench rc $?"
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r05_bench_pmc.json") if l.startswith("{")][-1])
r = j["roofline"]
print("headline", j["value"], r["kernel_ms"], r["frac"], r.get("traffic"), r.get("traffic_measured_in_this_run"), r.get("traffic_source"))
for k in ("c3", "c4"):
    r = j["secondary"][k]["roofline"]
    print(k, r["kernel_ms"], r["frac"], r.get("traffic"), r.get("traffic_measured_in_this_run"), r.get("traffic_over_algorithmic"), j["secondary"][k].get("wall_s"))
PY
