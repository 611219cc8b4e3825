#!/usr/bin/env python
"""Summary of tools/ab_variant.sh / ab_xad.sh output: GB/s of product and variant per shape (means over repeats)."""
import json
import sys
from collections import OrderedDict

rows = OrderedDict()
shape = None
for line in open(sys.argv[1]):
    line = line.rstrip()
    if line.startswith("=="):
        shape = line[3:]
        rows.setdefault(shape, {})
    elif ": {" in line and shape is not None:
        who, js = line.strip().split(": ", 1)
        try:
            d = json.loads(js)["packed"]
        except Exception:
            print(shape, who, "ERR", js[:80])
            continue
        r = rows[shape].setdefault(who, {"g": [], "eq": []})
        r["g"].append(d["GBps"])
        if "equal_to_first" in d:
            r["eq"].append(d["equal_to_first"])
for shape, r in rows.items():
    names = list(r)
    if len(names) < 2:
        continue
    a, b = r[names[0]], r[names[1]]
    ma, mb = sum(a["g"]) / len(a["g"]), sum(b["g"]) / len(b["g"])
    print(f"{shape:52s} {names[0]} {ma:7.0f}  {names[1]} {mb:7.0f}  ratio {mb / ma:.3f}  eq={all(b['eq'])}  "
          f"spread {min(a['g']):.0f}-{max(a['g']):.0f} / {min(b['g']):.0f}-{max(b['g']):.0f}")
