"""The ISA lint the build runs (bionumpy_amd/csrc/isa_lint.py) on listings with known defects, and on an object of the
library as built."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(ROOT, "bionumpy_amd", "csrc", "isa_lint.py"))
lint = importlib.util.module_from_spec(spec)
spec.loader.exec_module(lint)


def _findings(listing):
    out = []
    for name, items in lint.parse_listing(listing).items():
        out += [("wait",) + f for f in lint.check_kernel(name, items)] + [("scc",) + f for f in lint.check_scc(name, items)]
    return out


GOOD = """
kernel_a:
	s_load_dwordx2 s[4:5], s[0:1], 0x0
	s_waitcnt lgkmcnt(0)
	v_mov_b32_e32 v1, s4
	global_load_dwordx2 v[2:3], v[0:1], off
	global_load_dwordx2 v[4:5], v[0:1], off offset:8
	s_waitcnt vmcnt(1)
	v_add_u32_e32 v6, v2, v3
	s_waitcnt vmcnt(0)
	v_add_u32_e32 v6, v4, v6
	ds_read_b64 v[8:9], v6
	s_waitcnt lgkmcnt(0)
	v_add_u32_e32 v6, v8, v9
	s_endpgm
"""


def test_waits_are_recognised():
    assert _findings(GOOD) == []
    # the second load's value used under vmcnt(1): only the FIRST has landed
    bad = GOOD.replace("\ts_waitcnt vmcnt(0)\n", "")
    f = _findings(bad)
    assert len(f) == 1 and f[0][0] == "wait" and "v_add_u32_e32 v6, v4, v6" in f[0][2]
    # a scalar load is out of order: lgkmcnt(1) says nothing about it
    f = _findings(GOOD.replace("s_waitcnt lgkmcnt(0)\n\tv_mov_b32_e32 v1, s4", "s_waitcnt lgkmcnt(1)\n\tv_mov_b32_e32 v1, s4"))
    assert any("v_mov_b32_e32 v1, s4" in x[2] for x in f)
    # the LDS read used without its wait
    f = _findings(GOOD.replace("\ts_waitcnt lgkmcnt(0)\n\tv_add_u32_e32 v6, v8, v9", "\tv_add_u32_e32 v6, v8, v9"))
    assert len(f) == 1 and ('v', 8) in f[0][3]


def test_pending_loads_across_a_loop():
    loop = """
kernel_b:
	global_load_dword v2, v[0:1], off
.LBB0_1:
	s_waitcnt vmcnt(0)
	v_add_u32_e32 v3, v2, v3
	global_load_dword v2, v[0:1], off
	s_cbranch_scc1 .LBB0_1
	s_waitcnt vmcnt(0)
	v_mov_b32_e32 v4, v2
	s_endpgm
"""
    assert _findings(loop) == []
    # without the wait at the head of the loop the value of the previous trip is read in flight
    assert len(_findings(loop.replace(".LBB0_1:\n\ts_waitcnt vmcnt(0)\n", ".LBB0_1:\n"))) >= 1
    # a load into the destination of an older load of the same kind is not a hazard (they return in order)
    assert _findings(loop.replace("\tv_add_u32_e32 v3, v2, v3\n", "")) == []


def test_the_dropped_scc_copy_is_recognised():
    bad = """
kernel_c:
	s_add_u32 s4, s4, 1
	v_cmp_lt_i64_e32 vcc, s[6:7], v[2:3]
	s_cselect_b64 s[8:9], s[10:11], s[12:13]
	s_endpgm
"""
    f = _findings(bad)
    assert len(f) == 1 and f[0][0] == "scc"
    assert _findings(bad.replace("\tv_cmp_lt_i64_e32 vcc, s[6:7], v[2:3]\n\ts_cselect", "\tv_cmp_lt_i64_e32 vcc, s[6:7], v[2:3]\n\ts_and_b64 vcc, exec, vcc\n\ts_cselect")) == []


def test_an_object_of_the_library_is_clean():
    obj = os.path.join(ROOT, "bionumpy_amd", "csrc", "build", "scan.o")
    if not os.path.exists(obj):                                  # (a snapshot without the build directory: the library was linted when built)
        return
    findings, n_kernels, n_ins = lint.lint_objects([obj], verbose=False)
    assert findings == [] and n_kernels >= 1 and n_ins > 100
