#!/usr/bin/env python
"""Compile the headline scan kernel to ISA and check that its tile loop touches no scratch
(`scan_packed_kernel<2,64,false>` carries 36 B of scratch under its 128-VGPR cap
-- the spilled registers are saved before the tile loop and restored after it; inside the loop only
the cold list-flush branch stores one).

    python tools/check_scan_isa.py [--m 64] [--r 1] [--out profiles/r02_scan_isa.json]
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=64)
    ap.add_argument("--r", type=int, default=1)
    ap.add_argument("--rm", type=int, default=2, help="registers of the merged list (fused finish); 0 = three-launch kernel")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        subprocess.check_call([
            os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17",
            "-ffp-contract=off", "-fno-fast-math", "-DNDEBUG", f"-DTPQ_PACKED_M={a.m}", "-x", "hip",
            "--cuda-device-only", "-S", os.path.join(ROOT, "torchpq_amd", "csrc", "scan_packed.hip"),
            "-o", asm], stderr=subprocess.DEVNULL)
        text = open(asm).read()
    name = f"_ZN3tpq18scan_packed_kernelILi{a.r}ELi{a.m}ELb0ELi{a.rm}EEEvNS_8ScanArgsENS_12ResidualArgsEf"
    start = text.index(f"\n{name}:")
    body = text[start:text.index("s_endpgm", start)].splitlines()
    # loops: header label -> [first line, last line carrying "Header=<label>"]
    loops = {}
    for i, line in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):.*Loop Header: Depth=1", line)
        if m:
            loops[m.group(1)[2:]] = [i, i]
    for i, line in enumerate(body):
        m = re.search(r"Header=(BB\d+_\d+)", line)
        if m and m.group(1) in loops:
            loops[m.group(1)][1] = i
    # the tile loop = the depth-1 loop with the most LDS look-ups (v_perm_b32 address builds)
    def count(lo, hi, pat):
        return sum(1 for l in body[lo:hi + 1] if re.search(pat, l))
    tile = max(loops.items(), key=lambda kv: count(kv[1][0], kv[1][1], r"\bv_perm_b32\b"))
    lo, hi = tile[1]
    out = {
        "kernel": f"scan_packed_kernel<{a.r},{a.m},false,{a.rm}>",
        "isa_lines": len(body),
        "tile_loop": {"header": tile[0], "first_line": lo, "last_line": hi,
                      "v_perm_b32": count(lo, hi, r"\bv_perm_b32\b"),
                      "ds_read_b32": count(lo, hi, r"\bds_read_b32\b"),
                      "global_load_dwordx4": count(lo, hi, r"\bglobal_load_dwordx4\b"),
                      "v_pk_add_f32": count(lo, hi, r"\bv_pk_add_f32\b"),
                      "scratch_ops_inside": count(lo, hi, r"\bscratch_")},
        "scratch_stores_before_loop": count(0, lo - 1, r"\bscratch_store"),
        "scratch_loads_after_loop": count(hi + 1, len(body) - 1, r"\bscratch_load"),
        "scratch_ops_total": count(0, len(body) - 1, r"\bscratch_"),
    }
    m = re.search(rf"\.name:\s+{re.escape(name)}\n(?:.*\n){{0,12}}", text)
    blk = text[m.start() - 1500:m.end() + 600]
    for key in ("private_segment_fixed_size", "vgpr_count", "vgpr_spill_count", "sgpr_count"):
        mm = re.findall(rf"\.{key}:\s+(\d+)", blk)
        if mm:
            out[key] = int(mm[-1] if key != "private_segment_fixed_size" else mm[-1])
    # in-loop scratch accesses are tolerated only in the cold list-flush branch (the bitonic merge of
    # a full 64-entry candidate queue into the wave's sorted list: v_cmp_gt_u64 networks), which a
    # wave enters once per few hundred tiles; the streaming path (load, look-ups, threshold test)
    # must be scratch-free
    hot = []
    for i in range(lo, hi + 1):
        if re.search(r"\bscratch_", body[i]):
            near = body[max(lo, i - 60):i]
            if not any("v_cmp_gt_u64" in l for l in near):
                hot.append(i)
    out["tile_loop"]["scratch_ops_in_list_flush_branch"] = out["tile_loop"]["scratch_ops_inside"] - len(hot)
    out["tile_loop"]["scratch_ops_on_streaming_path"] = len(hot)
    print(json.dumps(out, indent=1))
    if a.out:
        open(a.out, "w").write(json.dumps(out, indent=1) + "\n")
    assert not hot, f"the streaming path of the tile loop touches scratch at ISA lines {hot}"


if __name__ == "__main__":
    main()
