#!/bin/bash
# Everything DESIGN section 4 quotes for a round, in one gpurun call (writes under gpurun_out/<tag>_* and copies the judged
# artefacts into profiles/ on the box -- they come back under gpurun_out/profiles_<tag>/; cp them into profiles/):
#   bash tools/round_profiles.sh r06
TAG="${1:-r06}"
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
cd "$ROOT"
# 1. per workload: plain run, rocprofv3 --kernel-trace --stats, separate --pmc passes, summary
bash tools/profile_bench.sh "$TAG" c2 c3 c4 c4cold c5 wide > "$OUT/${TAG}_profile_bench.log" 2>&1
python tools/collect_profiles.py "$TAG" >> "$OUT/${TAG}_profile_bench.log" 2>&1
# 2. the DRIVER's command, three times, with those summaries in place (its lines carry kernel_ms_profile /
#    profile_mismatch against them, and the traffic of their own counter pass)
for i in 1 2 3; do
  python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/${TAG}_bench_full_$i.json" 2> "$OUT/${TAG}_bench_full_$i.err"
done
cp "$OUT/${TAG}_bench_full_1.json" "profiles/${TAG}_bench_full.json"
# 3. sweeps, the reference's grid, probe routes, small batches, large k, soaks, the 100 M build
bash tools/scan_sweeps.sh > "$OUT/${TAG}_scan_sweeps.json" 2> /dev/null
bash tools/batch_sweep.sh > "$OUT/${TAG}_batch_sweep.json" 2> /dev/null
python tools/reference_grid.py --out "$OUT/${TAG}_reference_grid.json" > "$OUT/${TAG}_reference_grid.log" 2>&1
python tools/probe_bench.py --n-cells 1024,4096,16384 --n-probe 1,16,32,64,128 > "$OUT/${TAG}_probe_routes.jsonl" 2> /dev/null
{
  echo "{\"what\": \"tools/search_breakdown.py --preset c2 --nq N --iters 50: stream time of search() on the C2-shaped index (ms)\", \"rows\": ["
  first=1
  for nq in 1 16 256; do
    [ $first -eq 1 ] || echo ","
    first=0
    echo -n "$(python tools/search_breakdown.py --preset c2 --nq $nq --iters 50 2>/dev/null | tail -1)"
  done
  echo "]}"
} > "$OUT/${TAG}_small_batches.json"
{ python tools/dump_route_check.py; python tools/dump_route_check.py --large-k; python tools/dump_route_check.py --large-k-sweep; } \
  2> /dev/null | grep '^{' > "$OUT/${TAG}_dump_route.jsonl"
python tools/selection_soak.py --mode probe --cases 240 --seed 11 --big > "$OUT/${TAG}_soak_probe.json" 2>/dev/null
python tools/selection_soak.py --mode cascade --cases 120 --seed 12 > "$OUT/${TAG}_soak_cascade.json" 2>/dev/null
python tools/build_100m.py > "$OUT/${TAG}_build_100m.json" 2> /dev/null
# round 6: what bounds the scan (counters), the prologue's phases (a -DTPQ_SCAN_PROFILE variant, when one was built:
# tools/build_variant.sh prof "-DTPQ_SCAN_PROFILE" scan scan_packed_64 scan_packed_32), the batch-size breakdown
bash tools/scan_counters.sh > "$OUT/${TAG}_scan_counters.jsonl" 2> /dev/null
if [ -f torchpq_amd/variants/libtorchpq_amd_prof.so ]; then
  {
    export TPQ_AMD_LIB=$PWD/torchpq_amd/variants/libtorchpq_amd_prof.so
    P="python tools/scan_phase_profile.py --fused"
    $P --m 64 --ds 2 --n-cells 16384 --cell 61 --n-probe 32
    $P --m 64 --ds 2 --n-cells 16384 --cell 61 --n-probe 8
    $P --m 64 --ds 2 --n-cells 4096 --cell 244 --n-probe 32
    $P --m 64 --ds 2 --n-cells 1024 --cell 977 --n-probe 32
    $P --m 32 --ds 4 --n-cells 4096 --cell 244 --n-probe 32
    unset TPQ_AMD_LIB
  } 2>&1 | grep -v amdgpu.ids > "$OUT/${TAG}_scan_phases.txt"
fi
python tools/batch_breakdown.py 2>/dev/null | grep '^{' > "$OUT/${TAG}_batch_breakdown.jsonl"
python tools/ab_stream.py 2>/dev/null | grep '^{' > "$OUT/${TAG}_stream_regimes.json"
mkdir -p "$OUT/profiles_${TAG}"
cp profiles/${TAG}_* "$OUT/profiles_${TAG}/" 2>/dev/null
for f in scan_sweeps.json batch_sweep.json reference_grid.json probe_routes.jsonl small_batches.json dump_route.jsonl \
         soak_probe.json soak_cascade.json build_100m.json scan_counters.jsonl scan_phases.txt batch_breakdown.jsonl \
         stream_regimes.json; do
  cp "$OUT/${TAG}_$f" "$OUT/profiles_${TAG}/${TAG}_$f" 2>/dev/null
done
for i in 1 2 3; do cp "$OUT/${TAG}_bench_full_$i.json" "$OUT/profiles_${TAG}/${TAG}_bench_full_run$i.json"; done
tail -3 "$OUT/${TAG}_profile_bench.log"
ls -la "$OUT/profiles_${TAG}"
