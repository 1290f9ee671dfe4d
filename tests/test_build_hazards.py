"""The hand-written matrix-instruction streams of the screening kernels are inline assembly: the compiler cannot pad the hazards of
instructions it does not see.  gfx90a+ wants two wait states between a VALU write of a VGPR and a matrix instruction that reads it;
round 6 met the pattern twice (a register-allocator copy of a parked fragment right in front of its first use: wrong, timing-dependent
ranks).  scripts/check_mfma_hazards.py compiles kge_rank.hip to device assembly (hipcc cross-compiles without a GPU) and scans it."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_mfma_hazards", os.path.join(ROOT, "scripts", "check_mfma_hazards.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)


def test_scanner_sees_the_pattern_and_counts_wait_states():
    bad = """
_ZN3kge1kEv:
	v_accvgpr_read_b32 v50, a244
	v_accvgpr_read_b32 v51, a245
	v_mfma_i32_32x32x32_i8 v[32:47], a[148:151], v[48:51], 0
"""
    hits = chk.scan(bad)
    assert len(hits) == 2 and hits[0][0] == "_ZN3kge1kEv" and {h[1] for h in hits} == {0, 1}
    padded = bad.replace("\tv_mfma", "\ts_nop 1\n\tv_mfma")
    assert chk.scan(padded) == []
    one_short = bad.replace("\tv_mfma", "\ts_nop 0\n\tv_mfma")
    assert len(chk.scan(one_short)) == 1   # (the nearer copy has one wait state behind it, the farther two)
    unrelated = bad.replace("v[48:51]", "v[52:55]")
    assert chk.scan(unrelated) == []
    # a ds_read in between is an instruction: one wait state
    assert len(chk.scan(bad.replace("\tv_mfma", "\tds_read_b128 v[0:3], v9\n\tv_mfma"))) == 1


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_rank_kernels_have_no_unpadded_valu_write_in_front_of_a_matrix_instruction():
    text = chk.device_asm(os.path.join(ROOT, "ampligraph_amd", "csrc", "kge_rank.hip"))
    assert "v_mfma_i32_32x32x32_i8" in text
    hits = chk.scan(text)
    assert hits == [], "\n".join("%s: %s -> %s" % (k, w, m) for k, _, w, m in hits[:10])
    # rank_screen_kernel_r<13> fills the register file and parks a few dwords that are live across its tile loop
    # (tests/test_kernel_resources.py): stored before the loop, loaded behind it -- never between two matrix instructions
    kernel, lines = None, {}
    for raw in text.split("\n"):
        l = raw.split(";")[0].strip()
        if l.endswith(":") and l.startswith("_ZN3kge20rank_screen_kernel_r"):
            kernel = l[:-1]
            lines[kernel] = []
        elif l.startswith(".Lfunc_end"):
            kernel = None
        elif kernel and l and not l.startswith("."):
            lines[kernel].append(l)
    assert len(lines) >= 3
    for k, ls in lines.items():
        mf = [i for i, l in enumerate(ls) if l.startswith("v_mfma")]
        inside = [l for l in ls[mf[0]:mf[-1] + 1] if l.startswith("scratch_")]
        assert inside == [], (k, inside[:4])
