"""CPU: the static ISA guard (tools/check_isa.py, `make check-isa`) -- every k_render / wf_intersect instance keeps its
wave-uniform loop state (work item, pass number, ray-range cursor) in scalar registers.  Round 2's hang came from such a
value living in a VGPR and being spilled under a partial exec mask (profiles/r02/v_*)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_isa  # noqa: E402


GOOD = """
_ZN5ptamd8k_renderILb0ELi9EEEvNS_12RenderParamsE:
	s_load_dwordx2 s[6:7], s[0:1], 0x180
	v_mov_b32_e32 v3, 1
	global_atomic_add v3, v39, v3, s[6:7] sc0
	s_waitcnt vmcnt(0)
	v_readfirstlane_b32 s4, v3
.LBB0_1:
	s_add_i32 s6, s59, 1
	v_mov_b32_e32 v2, s6
	global_store_dword v39, v2, s[4:5] sc1
	s_endpgm
"""


def test_checker_accepts_the_scalar_form():
    (name, body), = list(check_isa.instances(GOOD))
    errs, n_atomic, n_pub = check_isa.check(name, body)
    assert errs == [] and n_atomic == 1 and n_pub == 1


def test_checker_rejects_a_work_item_kept_in_a_vgpr():
    bad = GOOD.replace("v_readfirstlane_b32 s4, v3", "ds_bpermute_b32 v3, v40, v3")
    (name, body), = list(check_isa.instances(bad))
    assert any("returning atomic" in e for e in check_isa.check(name, body)[0])
    bad = GOOD.replace("v_mov_b32_e32 v2, s6\n", "v_add_u32_e32 v2, 1, v17\n")        # pass number from a long-lived VGPR
    (name, body), = list(check_isa.instances(bad))
    assert any("agent-scope store" in e for e in check_isa.check(name, body)[0])
    # a copy of a scalar spilled lane by lane and taken for wave-uniform again after the reload: round 2's bug
    spill = "v_mov_b32_e32 v2, s6\n\tscratch_store_dword off, v2, off offset:8\n"
    back = "\tscratch_load_dword v7, off, off offset:8\n\ts_waitcnt vmcnt(0)\n\tv_readfirstlane_b32 s9, v7\n\ts_endpgm"
    bad = GOOD.replace("v_mov_b32_e32 v2, s6\n", spill).replace("\ts_endpgm", back)
    (name, body), = list(check_isa.instances(bad))
    assert any("scratch store" in e for e in check_isa.check(name, body)[0])
    # the same store is fine for a per-lane variable that only STARTS from a scalar (reloaded and used lane by lane)
    fine = GOOD.replace("v_mov_b32_e32 v2, s6\n", spill).replace("\ts_endpgm", "\tscratch_load_dword v7, off, off offset:8\n\tv_add_u32_e32 v7, 1, v7\n\ts_endpgm")
    (name, body), = list(check_isa.instances(fine))
    assert check_isa.check(name, body)[0] == []


def test_every_kernel_instance_of_the_library_passes():
    """`make check-isa` compiles pt_kernels.hip / pt_wavefront.hip to gfx950 assembly (cached under build/isa) and runs the
    checker over every k_render / wf_intersect instance."""
    r = subprocess.run(["make", "-s", "check-isa"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    n = int(last.split()[0])
    assert n >= 50 and last.endswith("0 failed"), last
