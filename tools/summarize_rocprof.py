#!/usr/bin/env python
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into one JSON summary for profiles/.

    python tools/summarize_rocprof.py <dir with stats/ pmc_*/> <kernel substring> <algorithmic bytes>
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    root, kernel, algo = sys.argv[1], sys.argv[2], float(sys.argv[3])
    out = {"kernel_filter": kernel, "algorithmic_bytes_per_launch": algo, "kernels": [], "pmc": {}}
    try:  # the kernel sources this profile was taken from (bench.py attaches it only on a match)
        import importlib.util
        here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(here, "bench.py"))
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        out["source_fingerprint"] = bench.source_fingerprint()
    except Exception as e:
        out["source_fingerprint"] = None
        out["source_fingerprint_error"] = repr(e)
    for f in glob.glob(os.path.join(root, "stats", "*kernel_stats.csv")):
        for r in list(csv.DictReader(open(f)))[:12]:
            out["kernels"].append({"name": r["Name"][:110], "calls": int(r["Calls"]),
                                   "avg_us": round(float(r["AverageNs"]) / 1e3, 2),
                                   "pct": float(r["Percentage"])})
    for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
        for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if kernel in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    out["vgpr"] = int(r["VGPR_Count"])
                    out["lds_block_bytes"] = int(r["LDS_Block_Size"])
                    out["grid"] = int(r["Grid_Size"])
            for k, v in agg.items():
                out["pmc"][k] = {"launches": len(v), "mean": sum(v) / len(v)}
            # every kernel of the k-means step, per kernel (c5: coarse / refine / exact re-check / update)
            per = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(f)):
                for tag in ("coarse_kernel", "refine_kernel", "update_kernel", "max_sim_kernel",
                            "select_resident_kernel", "centroid_accum_mfma_kernel", "max_sim_codebook_kernel",
                            "gemm_kernel<false", "gemm_kernel<true", "pair_exact_kernel", "gsplit_points_kernel"):
                    if tag in r["Kernel_Name"]:
                        per[tag][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for tag, ctrs in per.items():
                dst = out.setdefault("pmc_by_kernel", {}).setdefault(tag, {})
                for k, v in ctrs.items():
                    dst[k] = {"launches": len(v), "mean": sum(v) / len(v)}
    for tag, c in out.get("pmc_by_kernel", {}).items():  # derived shares (SQ_* in quad-cycles, MI355X_MICROARCH.md)
        d = {}
        if "GRBM_GUI_ACTIVE" in c:
            d["cycles_per_launch"] = c["GRBM_GUI_ACTIVE"]["mean"] / 8.0  # summed over the 8 XCDs
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                d["mfma_busy_share"] = c["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / 1024.0 / d["cycles_per_launch"]
        if "SQ_WAVE_CYCLES" in c:
            for k in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"):
                if k in c:
                    d[k.lower() + "_share_of_wave_cycles"] = c[k]["mean"] / c["SQ_WAVE_CYCLES"]["mean"]
        if d:
            c["derived"] = d
    p = out["pmc"]
    if "FETCH_SIZE" in p:
        # FETCH_SIZE is in KiB and, on gfx950, counts 64 B per 128-B request: x2 (MI355X_MICROARCH HBM)
        fetched = p["FETCH_SIZE"]["mean"] * 1024 * 2
        out["hbm_side_read_bytes_corrected"] = fetched
        out["traffic_over_algorithmic"] = fetched / algo
    if "TCC_EA0_RDREQ_sum" in p:
        out["tcc_ea_rdreq_x128B"] = p["TCC_EA0_RDREQ_sum"]["mean"] * 128
    if "TCC_HIT_sum" in p:
        out["l2_hit_rate"] = p["TCC_HIT_sum"]["mean"] / (p["TCC_HIT_sum"]["mean"] + p["TCC_MISS_sum"]["mean"])
    if "SQ_LDS_BANK_CONFLICT" in p:
        out["lds_conflict_over_active"] = p["SQ_LDS_BANK_CONFLICT"]["mean"] / p["SQ_LDS_IDX_ACTIVE"]["mean"]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
