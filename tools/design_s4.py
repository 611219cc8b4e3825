#!/usr/bin/env python
"""DESIGN.md section 4 from tools/design_s4_template.md + the tracked records of a round (profiles/<tag>_*): the tables are
tools/design_tables.py's, the inline figures are read from the same files.  Rewrites the section in place.
    python tools/design_s4.py r06 [--tests 769]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import design_tables as dt  # noqa: E402

P = os.path.join(ROOT, "profiles")


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    ntests = sys.argv[sys.argv.index("--tests") + 1] if "--tests" in sys.argv else "?"
    t = open(os.path.join(ROOT, "tools", "design_s4_template.md")).read()
    runs = [dt.line(os.path.join(P, f"{tag}_bench_full_run{i}.json")) for i in (1, 2, 3)]
    cold = [r["secondary"]["c4"]["cold"]["roofline"] for r in runs]
    warm = [r["secondary"]["c4"]["roofline"] for r in runs]
    grid = json.load(open(os.path.join(P, f"{tag}_reference_grid.json")))
    g = {(p["m"], p["n_cells"], p["n_probe"], p["k"]): p for p in grid["points"]}
    sweeps = json.load(open(os.path.join(P, f"{tag}_scan_sweeps.json")))
    phases = open(os.path.join(P, f"{tag}_scan_phases.txt")).read().strip()
    ctr = [json.loads(l) for l in open(os.path.join(P, f"{tag}_scan_counters.jsonl")) if l.startswith("{")]
    c0 = [c for c in ctr if c["m"] == 64 and c["n_cells"] == 16384][0]
    rep = {
        "VALU_RANGE": f"{100 * min(c['valu_issue_share'] for c in ctr):.0f}–{100 * max(c['valu_issue_share'] for c in ctr):.0f} %",
        "VALU_IVF16384": f"{100 * c0['valu_issue_share']:.0f} %", "INSTR_IVF16384": f"{c0['valu_instructions_per_query']:,}".replace(",", " "),
        "NTESTS": ntests,
        "DRIVER": dt.driver_table(tag), "GRID": dt.grid_table(tag), "SWEEPS": dt.sweeps(tag), "COUNTERS": dt.counters(tag),
        "PHASES": "```\n" + phases + "\n```",
        "COLD_TBPS": dt.rng([c["achieved"] / 1e3 for c in cold]), "COLD_FRAC": dt.rng([c["frac"] for c in cold], "{:.3f}"),
        "WARM_OVER_COLD": dt.rng([w["rate_over_cold_rate"] for w in warm]),
        "BELOW": str(grid["summary"]["m32_m64_nprobe_ge16_below_0.60"]),
        "P64": f"{g[(64, 16384, 32, 100)]['frac']:.2f}", "P32": f"{g[(32, 4096, 32, 100)]['frac']:.2f}",
        "MINX": f"{grid['summary']['min_x_t4']:.1f}",
        "K1000": f"{sweeps['k_sweep_m64']['1000']['packed']['GBps'] / 1e3:.2f}",
        "SS8": dt.rng([r["strong_scaling_prediction"]["efficiency_at_n_gpus"]["8"] for r in runs]),
    }
    for k, v in rep.items():
        t = t.replace("{{" + k + "}}", v)
    left = re.findall(r"\{\{[A-Z0-9_]+\}\}", t)
    assert not left, left
    path = os.path.join(ROOT, "DESIGN.md")
    d = open(path).read()
    i0 = d.index("## 4. Measurement")
    i1 = d.index("## 5. Oracle and parity")
    open(path, "w").write(d[:i0] + t.rstrip("\n") + "\n\n" + d[i1:])
    print("DESIGN.md section 4 rewritten from", tag)


if __name__ == "__main__":
    main()
