#!/usr/bin/env python
"""Look for the store-data overwrite pattern in the compiled kernels.

Measured on gfx950 (round 4, probe_sims_kernel): a 16-byte buffer store whose first data register a VALU
instruction overwrote three instructions later -- hipcc's own schedule -- stored the NEW value in 9 % of the rows
of a 10 000 x 4 096 matrix; with the overwrite further away (or an s_nop after the store) the rows were right every
time.  (In isolation -- tools/experiments/store_hazard -- only distances of one and two instructions reproduce
it; the spill stores and assign_prep_kernel's stores this tool lists at distance three are clean in their soaks.)  This tool
compiles each translation unit to assembly and lists every store of more than 8 bytes whose data registers a
vector instruction writes within --window instructions (default 3) of the store, in straight-line code.

    python tools/check_store_hazard.py [--window 3] [unit.hip ...]

Exit status 1 when a hit is found: a list to look at, not a verdict (probe_sims_kernel carries an explicit pad,
TPQ_STORE_PAD).
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "torchpq_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-DNDEBUG",
         "-x", "hip", "-S", "--cuda-device-only"]

STORE = re.compile(r"^\s*(buffer|global|flat|scratch)_store_(dwordx3|dwordx4|b96|b128)\s+(.*)$")
VREG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def regs(tok):
    m = VREG.fullmatch(tok.strip())
    if not m:
        return set()
    if m.group(3) is not None:
        return {int(m.group(3))}
    return set(range(int(m.group(1)), int(m.group(2)) + 1))


def store_data(line):
    m = STORE.match(line)
    if not m:
        return None
    ops = [o.strip() for o in m.group(3).split(",")]
    # buffer_store: vdata first; global/flat/scratch_store: vaddr, vdata
    tok = ops[0] if m.group(1) == "buffer" else (ops[1] if len(ops) > 1 else "")
    return regs(tok)


def writes(line):
    """destination vector registers of a VALU / MFMA / permlane instruction (first operand; swaps: two)"""
    t = line.strip()
    if not t.startswith("v_") or t.startswith("v_cmp") or t.startswith("v_nop"):
        return set()
    ops = t.split(None, 1)
    if len(ops) < 2:
        return set()
    parts = [o.strip() for o in ops[1].split(",")]
    w = regs(parts[0])
    if t.startswith("v_permlane32_swap") or t.startswith("v_permlane16_swap") or t.startswith("v_swap"):
        w |= regs(parts[1]) if len(parts) > 1 else set()
    return w


def check(path, window, extra):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + extra + [path, "-o", out],
                              stderr=subprocess.DEVNULL)
        lines = open(out).read().splitlines()
    return scan_lines(lines, window)


def scan_lines(lines, window):
    """the hits in one assembly listing: (kernel, line number, store, distance, overwriting instruction)"""
    hits = []
    kernel = "?"
    pending = []  # (data regs, instructions seen since, store text, line number)
    for n, raw in enumerate(lines, 1):
        line = raw.split(";")[0].rstrip()
        if not line.strip():
            continue
        if re.match(r"^[A-Za-z_.$][\w.$]*:", line):
            if line.startswith("_Z") or line.startswith("tpq"):
                kernel = line.split(":")[0]
            pending = []  # (straight-line code only)
            continue
        if line.lstrip().startswith("."):
            continue
        w = writes(line)
        nxt = []
        for data, seen, text, ln in pending:
            if w & data:
                hits.append((kernel, ln, text.strip(), seen + 1, line.strip()))
                continue
            if line.strip().startswith("s_nop"):
                m = re.search(r"s_nop\s+(\d+)", line)
                seen += int(m.group(1)) if m else 0
            if seen + 1 < window and not line.strip().startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc")):
                nxt.append((data, seen + 1, text, ln))
        pending = nxt
        d = store_data(line)
        if d:
            pending.append((d, 0, line, n))
    return hits


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--window", type=int, default=3)
    ap.add_argument("units", nargs="*")
    args = ap.parse_args()
    units = args.units or sorted(f for f in os.listdir(CS) if f.endswith(".hip"))
    bad = 0
    for u in units:
        path = u if os.path.isabs(u) else os.path.join(CS, u)
        variants = [[]]
        if os.path.basename(path) == "scan_packed.hip":
            variants = [[f"-DTPQ_PACKED_M={m}"] for m in (8, 16, 32, 64)]
        for extra in variants:
            hits = check(path, args.window, extra)
            tag = os.path.basename(path) + (" " + " ".join(extra) if extra else "")
            print(f"{tag}: {len(hits)} store(s) with data overwritten within {args.window} instructions")
            for k, ln, st, dist, wr in hits[:12]:
                print(f"    {k[:70]} line {ln}: {st}  <- +{dist}: {wr}")
            bad += len(hits)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
