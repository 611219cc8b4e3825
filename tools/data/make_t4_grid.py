#!/usr/bin/env python
"""Transcribes the reference's published SIFT1M grid (1x Tesla T4) into tools/data/t4_sift1m_grid.json.
Runs in the build container only (reads /root/reference); the JSON it writes is data: 64 records of
published q/s and recall numbers, used by tools/reference_grid.py as the comparison column."""
import json
import os

SRC = "/root/reference/benchmark/turing/sift1m/json/ivf[8, 16, 32, 64]_pq[4096, 16384]_sift1m.json"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    recs = json.load(open(SRC))
    out = {
        "source": "benchmark/turing/sift1m/json/ivf[8, 16, 32, 64]_pq[4096, 16384]_sift1m.json:1 "
                  "(DeMoriarty/TorchPQ, 1x Tesla T4, SIFT1M, 10 000 queries, mean of 30 runs; 64 records)",
        "note": "published benchmark numbers (data), transcribed by tools/data/make_t4_grid.py; file-name labels "
                "are swapped upstream: first list = n_subvectors, second = n_cq_clusters",
        "records": [{"m": r["n_subvectors"], "n_cells": r["n_cq_clusters"], "n_probe": r["n_probe"],
                     "qps": {k: round(r[f"q/s@{k}"], 1) for k in ("1", "10", "100")},
                     "recall": {k: round(r[f"recall@{k}"], 4) for k in ("1", "10", "100")},
                     "train_s": round(r["train_time"], 3), "add_s": round(r["add_time"], 3)} for r in recs]}
    json.dump(out, open(os.path.join(HERE, "t4_sift1m_grid.json"), "w"), indent=0)
    print(len(out["records"]), "records")


if __name__ == "__main__":
    main()
