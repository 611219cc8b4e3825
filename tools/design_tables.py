#!/usr/bin/env python
"""The tables of DESIGN section 4, generated from the tracked records (profiles/<tag>_*): nothing in that section is
typed by hand from another run.

    python tools/design_tables.py r05 > /tmp/tables.md
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def line(path):
    return json.loads([l for l in open(path) if l.startswith("{")][-1])


def rng(vals, fmt="{:.2f}"):
    vals = [v for v in vals if v is not None]
    if not vals:
        return "—"
    lo, hi = min(vals), max(vals)
    return fmt.format(lo) if fmt.format(lo) == fmt.format(hi) else f"{fmt.format(lo)}–{fmt.format(hi)}"


def driver_table(tag):
    runs = [line(os.path.join(P, f"{tag}_bench_full_run{i}.json")) for i in (1, 2, 3)
            if os.path.exists(os.path.join(P, f"{tag}_bench_full_run{i}.json"))]
    out = [f"Driver command `python bench.py --gpus 1 --steps 20 --warmup 5`, {len(runs)} runs in one gpurun call "
           f"(`profiles/{tag}_bench_full_run*.json`; run 1 = `{tag}_bench_full.json`):", "",
           "| record | value | scan / dominant kernel ms (min–max over runs; per-step min / median / max of run 1) | "
           "achieved | frac of 8 TB/s (or of the dense matrix peak) | fed by | traffic measured in the run ÷ algorithmic | "
           "queries redone exactly | profile cross-check |", "|---|---|---|---|---|---|---|---|---|"]

    def scan_row(name, get, unit="q/s"):
        rs = [get(r) for r in runs]
        rf = [r["roofline"] for r in rs]
        r0 = rf[0]
        val = rng([r["value"] / 1e6 for r in rs], "{:.3f}") + " M " + unit
        ms = rng([r["kernel_ms"] for r in rf], "{:.3f}")
        spread = f"{r0.get('kernel_ms_min')} / {r0.get('kernel_ms_median')} / {r0.get('kernel_ms_max')}"
        tr = rng([r.get("traffic_over_algorithmic") for r in rf if r.get("traffic_measured_in_this_run")], "{:.2f}")
        chk = (f"{r0.get('kernel_ms_profile')} ms, mismatch {r0.get('profile_mismatch')}"
               if r0.get("kernel_ms_profile") is not None else "—")
        out.append(f"| {name} | {val} | {ms} ({spread}) | {rng([r['achieved'] / 1e3 for r in rf])} TB/s | "
                   f"{rng([r['frac'] for r in rf], '{:.3f}')} | {r0.get('fed_by', '—')} | {tr} | "
                   f"{rng([r.get('queries_redone_exactly') for r in rf], '{:d}')} | {chk} |")

    scan_row("**C2** headline (configs[1])", lambda r: r)
    for k, label in (("c3", "C3 GIST shape (configs[2])"), ("c4", "C4 100 M slots (configs[3], one GPU), the timed batch")):
        if all(k in r.get("secondary", {}) and "roofline" in r["secondary"][k] for r in runs):
            scan_row(label, lambda r, k=k: r["secondary"][k])
    if all("roofline" in r.get("secondary", {}).get("c4", {}).get("cold", {}) for r in runs):
        scan_row("C4 COLD: every cell probed once per launch (the DRAM figure)", lambda r: r["secondary"]["c4"]["cold"])
    if all("roofline" in r.get("secondary", {}).get("residual", {}) for r in runs):
        scan_row("residual PQ at the C2 shape (SURVEY 8f-3; (m + 4) B per slot)", lambda r: r["secondary"]["residual"])
    c5 = [r["secondary"]["c5"] for r in runs if "c5" in r.get("secondary", {}) and "iter_ms" in r["secondary"]["c5"]]
    if c5:
        out.append(f"| C5 Lloyd iteration (configs[4]) | iteration {rng([c['iter_ms'] for c in c5])} ms = assign "
                   f"{rng([c['assign_ms'] for c in c5])} + update {rng([c['update_ms_derived'] for c in c5])} | "
                   f"assign {rng([c['roofline']['kernel_ms'] for c in c5])} | "
                   f"{rng([c['roofline']['achieved'] for c in c5], '{:.0f}')} TF/s fp32-equivalent | "
                   f"{rng([c['roofline']['frac'] for c in c5], '{:.3f}')} (issued flops over 2.5 PF) | — | — | — | labels == fp32 "
                   f"kernel: {rng([c['assign_labels_equal_to_fp32_kernel'] for c in c5], '{:.6f}')} |")
    wd = [r["secondary"]["wide"] for r in runs if "wide" in r.get("secondary", {}) and "ms" in r["secondary"]["wide"]]
    if wd:
        out.append(f"| wide coarse assign 1 M × 16 384 × 960 | {rng([w['ms'] for w in wd])} ms | — | "
                   f"{rng([w['roofline']['achieved'] for w in wd], '{:.0f}')} TF/s | "
                   f"{rng([w['roofline']['frac'] for w in wd], '{:.3f}')} | — | — | — | labels == fp32 kernel: "
                   f"{rng([w['labels_equal_to_fp32_kernel'] for w in wd], '{:.4f}')} |")
    c1 = [r["secondary"]["c1"] for r in runs if "c1" in r.get("secondary", {}) and "cpu" in r["secondary"]["c1"]]
    if c1:
        c = c1[0]
        out.append(f"| C1 (configs[0]) | CPU (oracle, {c['cpu']['cores']} threads): train {c['cpu']['train_s']} s, add "
                   f"{c['cpu']['add_s']} s, search {c['cpu']['search_queries_per_s']:.0f} q/s; the same index on the GPU: "
                   f"{rng([x['gpu']['search_queries_per_s'] / 1e6 for x in c1])} M q/s | — | — | — | — | — | — | ids equal "
                   f"{c['ids_equal_to_oracle']}, values within {c['values_max_rel_diff_vs_oracle']:.1e} |")
    fl = [r["secondary"]["flat"] for r in runs if "value" in r.get("secondary", {}).get("flat", {})]
    if fl:
        out.append(f"| FlatIndex exact search, 1 M x 128, 1 000 queries (SURVEY 8f-4) | {rng([f['value'] / 1e3 for f in fl], '{:.1f}')} k q/s | "
                   f"{rng([f['roofline']['kernel_ms'] for f in fl])} (whole search()) | "
                   f"{rng([f['roofline']['achieved'] / 1e3 for f in fl])} TB/s of the sims matrix written + read | "
                   f"{rng([f['roofline']['frac'] for f in fl], '{:.3f}')} | — | — | — | top-k overlap with float64 brute force "
                   f"{rng([f['oracle_check']['top_k_overlap_with_float64_brute_force'] for f in fl], '{:.4f}')} |")
    r0 = runs[0]

    def checks(r):
        rows = [("C2", r.get("oracle_check"))]
        sec = r.get("secondary", {})
        rows += [(k, sec.get(k, {}).get("oracle_check")) for k in ("c3", "c4", "residual")]
        rows.append(("c4 cold", sec.get("c4", {}).get("cold", {}).get("oracle_check")))
        return [(k, c) for k, c in rows if c]
    out += ["", "Oracle check ON the record (rows of the timed search() call vs the C oracle on the GPU's probed cells; run 1): " +
            "; ".join(f"{k}: {c['queries_checked']} rows, ids equal {c['ids_equal_to_oracle']}, values bit-equal "
                      f"{c['values_bit_equal']}" + (f", addresses beyond 2^24: {c['addresses_beyond_2p24']}"
                                                     if c.get("addresses_beyond_2p24") else "")
                      for k, c in checks(r0)) + "."]
    sp = r0["secondary"]["stream_peak"]
    if "settings_GBps" in sp:
        out += ["", "Stream-read settings of run 1 (8 GiB read once per launch, TB/s): " +
                "; ".join(f"{k}: {v / 1e3:.2f}" for k, v in sp["settings_GBps"].items()) + "."]
    out += ["", f"Stream peak of the box (8 GiB, dwordx4): {rng([r['secondary']['stream_peak']['value'] / 1e3 for r in runs])} "
            f"TB/s. CPU baseline (oracle, {r0['cpu_baseline']['cores']} threads, {r0['cpu_baseline']['sample'].split(',')[0]}): "
            f"{rng([r['cpu_baseline']['value'] for r in runs], '{:.0f}')} q/s, split {r0['cpu_baseline']['split_s']}. "
            f"recall_gt@100 {r0.get('recall_gt@100')}, recall_vs_ref@100 {r0.get('recall_vs_ref@100')}. "
            f"Strong-scaling ceiling from one GPU (rate of a 10 000/N batch ÷ rate of the full batch): "
            f"{r0.get('strong_scaling_prediction', {}).get('efficiency_at_n_gpus')}."]
    return "\n".join(out)


def grid_table(tag, prev="r05"):
    g = json.load(open(os.path.join(P, f"{tag}_reference_grid.json")))
    t = {(p["m"], p["n_cells"], p["n_probe"]): p for p in g["points"] if p["k"] == 100}
    t4 = {}
    if os.path.exists(os.path.join(P, f"{prev}_reference_grid.json")):
        g4 = json.load(open(os.path.join(P, f"{prev}_reference_grid.json")))
        t4 = {(p["m"], p["n_cells"], p["n_probe"]): p for p in g4["points"] if p["k"] == 100}
    nps = (1, 8, 16, 32, 64, 128)
    out = [f"`profiles/{tag}_reference_grid.json` summary: `{json.dumps(g['summary'])}`", "",
           "k = 100: M q/s (× T4) / scan fraction of 8 TB/s" + (f" [{prev}: fraction]" if t4 else ""), "",
           "| m, cells | " + " | ".join(f"n_probe {n}" for n in nps) + " |", "|---|" + "---|" * len(nps)]
    for m in (64, 32, 16, 8):
        for nc in (4096, 16384):
            if (m, nc, 1) not in t:
                continue
            cells = []
            for n in nps:
                p = t[(m, nc, n)]
                c = f"{p['qps'] / 1e6:.1f} ({p['x_t4']:.0f}×) / {p['frac']:.2f}"
                if (m, nc, n) in t4:
                    c += f" [{t4[(m, nc, n)]['frac']:.2f}]"
                cells.append(c)
            out.append(f"| {m}, IVF{nc} | " + " | ".join(cells) + " |")
    return "\n".join(out)


def sweeps(tag):
    out = []
    f = os.path.join(P, f"{tag}_scan_sweeps.json")
    if os.path.exists(f):
        j = json.load(open(f))
        out.append("k sweep at the C2 shape, caller-supplied LUT (`tools/scan_microbench.py`; ms per 10 000 queries, TB/s): " +
                   ", ".join(f"k={k}: {v['packed']['ms']:.2f} / {v['packed']['GBps'] / 1e3:.2f}" for k, v in j["k_sweep_m64"].items()))
        out.append("m sweep (k = 100, TB/s): " + ", ".join(f"{m}: {v['packed']['GBps'] / 1e3:.2f}" for m, v in j["m_sweep"].items()))
    f = os.path.join(P, f"{tag}_batch_sweep.json")
    if os.path.exists(f):
        j = json.load(open(f))
        for p in ("c2", "c4"):
            base = j[p]["10000"]["qps"]
            out.append(f"batch sweep {p} (search() end to end, M q/s and share of the 10 000-query rate): " +
                       ", ".join(f"{k}: {v['qps'] / 1e6:.2f} ({v['qps'] / base:.2f})" for k, v in j[p].items()))
    f = os.path.join(P, f"{tag}_dump_route.jsonl")
    if os.path.exists(f):
        rows = [json.loads(l) for l in open(f) if l.startswith("{")]
        big = [r for r in rows if r.get("fused") and r["nq"] == 10000 and r["cell"] == 977 and r["n_probe"] == 32 and r["cells"] == 1024]
        seen = {}
        for r in big:
            seen.setdefault(r["k"], r)
        out.append("fused search-path scan at the C2 shape against k (`tools/dump_route_check.py`; ms per 10 000 queries / TB/s): " +
                   ", ".join(f"k={k}: {r['ms']:.2f} / {r['GBps'] / 1e3:.2f}" for k, r in sorted(seen.items())))
        out.append(f"all {sum(1 for r in rows if 'equal' in r)} checked shapes bit-equal to the reference-layout kernel: "
                   f"{all(r['equal'] for r in rows if 'equal' in r)}")
    return "\n\n".join(out)


def counters(tag):
    f = os.path.join(P, f"{tag}_scan_counters.jsonl")
    if not os.path.exists(f):
        return ""
    rows = [json.loads(l) for l in open(f) if l.startswith("{")]
    out = ["What bounds the scan (`tools/scan_counters.sh`: rocprofv3 --pmc, counters only; 10 000 queries, 32 probes, k = 100):", "",
           "| m, cells x slots | kernels (us) | VALU instructions per query | per code byte | VALU issue share | LDS instructions "
           "per query | wave cycles waiting | LDS conflict / active |", "|---|---|---|---|---|---|---|---|"]
    for r in rows:
        ks = ", ".join(f"{k.split('<')[0].replace('scan_', '')} {v}" for k, v in r["kernels_us"].items())
        out.append(f"| {r['m']}, {r['n_cells']} x {r['cell_slots']} | {ks} | {r['valu_instructions_per_query']} | "
                   f"{r['valu_instructions_per_code_byte']} | **{r['valu_issue_share']:.2f}** | {r['lds_instructions_per_query']} | "
                   f"{r['wait_any_share_of_wave_cycles']:.2f} | {r['lds_bank_conflict_over_active']:.2f} |")
    return "\n".join(out)


if __name__ == "__main__":
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    print(driver_table(tag))
    print()
    print(grid_table(tag))
    print()
    print(sweeps(tag))
    print()
    print(counters(tag))
