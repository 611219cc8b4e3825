"""CPU checks of the analysis tools' own logic (no compiler, no GPU)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_store_hazard_checker_finds_the_pattern_it_documents():
    """the sequence hipcc emitted in probe_sims_kernel: a 16-byte buffer store, two independent VALU
    instructions, then a write of the store's first data register"""
    chk = _load("check_store_hazard")
    bad = """
_ZN3tpq5lloyd17probe_sims_kernelILi8EEEvNS0_13ProbeSimsArgsE:
	v_permlane32_swap_b32_e32 v20, v22
	buffer_store_dwordx4 v[20:23], v214, s[36:39], 32 offen
	v_mul_f32_e32 v6, v215, v11
	v_max_f32_e32 v7, v12, v12
	v_max_f32_e32 v20, v5, v4
	s_endpgm
""".splitlines()
    hits = chk.scan_lines(bad, 3)
    assert len(hits) == 1 and hits[0][3] == 3 and "v_max_f32_e32 v20" in hits[0][4]
    assert chk.scan_lines(bad, 2) == []           # the window is a distance in instructions
    # the fixed epilogue: stores back to back, a pad, the registers rewritten much later
    good = """
kernel:
	buffer_store_dwordx4 v[16:19], v213, s[36:39], 0 offen
	buffer_store_dwordx4 v[20:23], v213, s[36:39], 32 offen
	s_nop 7
	buffer_load_dwordx4 v[112:115], v172, s[20:23], s84 offen
	v_mfma_f32_32x32x16_f16 v[32:47], v[132:135], v[48:51], 0
	s_endpgm
""".splitlines()
    assert chk.scan_lines(good, 3) == []
    # global stores name the address first, the data second; 8-byte stores are not the pattern; a label ends the window
    other = """
kernel:
	global_store_dwordx4 v[6:7], v[0:3], off offset:1024
	v_cvt_pk_bf16_f32 v3, v4, v10
	buffer_store_dwordx2 v[8:9], v1, s[0:3], 0 offen
	v_mov_b32_e32 v8, 0
	scratch_store_dwordx4 off, v[40:43], off offset:48
.LBB0_1:
	v_mov_b32_e32 v43, 0
""".splitlines()
    hits = chk.scan_lines(other, 3)
    assert [h[2].split()[0] for h in hits] == ["global_store_dwordx4"] and hits[0][3] == 1
