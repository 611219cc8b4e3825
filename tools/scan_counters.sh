#!/bin/bash
# What bounds the list scan (round 6): rocprofv3 --pmc passes (counters only) + a --kernel-trace --stats pass over the scan of
# three shapes of the reference grid / BASELINE.json; prints one JSON line per shape with the per-launch instruction counts and
# the VALU issue share: SQ_INSTS_VALU x 4 cycles (a wave64 instruction occupies its SIMD16 for four cycles:
# tools/ubench/valu_rate.hip, 1.9 ns) over 1 024 SIMDs x the kernel's cycles (GRBM_GUI_ACTIVE / 8 XCDs).
#   bash tools/scan_counters.sh > gpurun_out/r06_scan_counters.jsonl
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
G="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE;SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for shape in 64,2,16384,61,32,100,10000 64,2,4096,244,32,100,10000 64,2,1024,977,32,100,10000 32,4,4096,244,32,100,10000 16,2,4096,244,32,100,10000 8,2,4096,244,32,100,10000; do
  tag=ctr_$(echo $shape | tr ',' '_')
  bash "$ROOT/tools/pmc.sh" $tag "scan_packed_kernel" "$G" python "$ROOT/tools/dump_route_check.py" --one $shape --iters 3 > /dev/null 2>&1
  bash "$ROOT/tools/kstats.sh" ${tag}_ks python "$ROOT/tools/dump_route_check.py" --one $shape --iters 3 > /dev/null 2>&1
  python - "$ROOT/gpurun_out/$tag/pmc_summary.json" "$ROOT/gpurun_out/${tag}_ks/stats" "$shape" <<'PY'
import csv, glob, json, os, sys
pmc = json.load(open(sys.argv[1]))["pmc"]
m, ds, cells, cell, n_probe, k, nq = (int(x) for x in sys.argv[3].split(","))
ks = {}
for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "scan_packed_kernel" in r["Name"] or "scan_finish_exact_kernel" in r["Name"]:
            ks[r["Name"].split("(")[0].replace("void tpq::", "")] = round(float(r["AverageNs"]) / 1e3, 1)
g = lambda n: pmc.get(n, {}).get("mean")
cyc = g("GRBM_GUI_ACTIVE") / 8.0
out = {"m": m, "ds": ds, "n_cells": cells, "cell_slots": cell, "n_probe": n_probe, "k": k, "n_query": nq,
       "kernels_us": ks, "cycles_per_launch_under_the_counter_pass": round(cyc),
       "valu_instructions_per_query": round(g("SQ_INSTS_VALU") / nq), "lds_instructions_per_query": round(g("SQ_INSTS_LDS") / nq),
       "salu_instructions_per_query": round(g("SQ_INSTS_SALU") / nq), "vmem_reads_per_query": round(g("SQ_INSTS_VMEM_RD") / nq, 1),
       "valu_issue_share": round(g("SQ_INSTS_VALU") * 4.0 / (1024.0 * cyc), 3),
       "valu_instructions_per_code_byte": round(g("SQ_INSTS_VALU") / (nq * n_probe * cell * m), 3),
       "wait_any_share_of_wave_cycles": round(g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 3),
       "lds_bank_conflict_over_active": round(g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"), 3)}
print(json.dumps(out))
PY
done
