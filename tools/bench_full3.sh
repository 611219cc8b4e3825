#!/bin/bash
# the driver's command three times -> gpurun_out/<tag>_bench_full_{1,2,3}.json (+ profiles_<tag>/ copies)
TAG="${1:-r05}"
cd "$(dirname "${BASH_SOURCE[0]}")/.."
mkdir -p gpurun_out/profiles_${TAG}
for i in 1 2 3; do
  python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_full_$i.json 2> gpurun_out/${TAG}_bench_full_$i.err
  echo "run $i rc $?"
  cp gpurun_out/${TAG}_bench_full_$i.json gpurun_out/profiles_${TAG}/${TAG}_bench_full_run$i.json
done
cp gpurun_out/${TAG}_bench_full_1.json gpurun_out/profiles_${TAG}/${TAG}_bench_full.json
python - <<PY
import json
for i in (1, 2, 3):
    j = json.loads([l for l in open("gpurun_out/${TAG}_bench_full_%d.json" % i) if l.startswith("{")][-1])
    r = j["roofline"]
    print(i, j["value"], r["kernel_ms"], r["frac"], r.get("profile_mismatch"), {k: v.get("roofline", {}).get("kernel_ms") for k, v in j["secondary"].items() if isinstance(v, dict)})
PY
