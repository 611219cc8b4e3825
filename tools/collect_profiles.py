#!/usr/bin/env python
"""Copy the judged artefacts of tools/profile_bench.sh from gpurun_out/<tag>/<W>/ (scratch) into
profiles/ (tracked):  <tag>_<name>_bench.json, <tag>_<name>_kernel_stats.csv, <tag>_<name>.json
with <name> = bench_scan_packed for the headline (c2) and c3 / c4 / c5 / wide otherwise."""
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    for w in ("c2", "c3", "c4", "c4cold", "c5", "wide"):
        src = os.path.join(ROOT, "gpurun_out", tag, w)
        if not os.path.isdir(src):
            continue
        name = "bench_scan_packed" if w == "c2" else w
        dst = os.path.join(ROOT, "profiles")
        lines = [l for l in open(os.path.join(src, "bench.json")) if l.startswith("{")]
        if lines:
            open(os.path.join(dst, f"{tag}_{name}_bench.json"), "w").write(lines[-1])
        for f in glob.glob(os.path.join(src, "stats", "*kernel_stats.csv")):
            shutil.copy(f, os.path.join(dst, f"{tag}_{name}_kernel_stats.csv"))
        if os.path.exists(os.path.join(src, "summary.json")):
            shutil.copy(os.path.join(src, "summary.json"), os.path.join(dst, f"{tag}_{name}.json"))
        print("collected", w)


if __name__ == "__main__":
    main()
