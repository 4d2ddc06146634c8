"""CPU-side lint of the comb walker's gfx950 ISA (hipcc cross-compiles; no GPU): the DPP source hazard the compiler's
hazard recogniser cannot see inside inline assembly (tools/dpp_hazard_check.py; round-4 advisor finding)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("dpp_hazard_check", os.path.join(ROOT, "tools", "dpp_hazard_check.py"))
lint = importlib.util.module_from_spec(spec)
spec.loader.exec_module(lint)

DPP = "v_subrev_f32_dpp v20, v10, v10 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"


def test_the_scan_sees_a_hazard_and_its_cures():
    assert len(lint.check(["v_pk_add_f32 v[10:11], v[1:2], v[3:4]", DPP])) == 1                        # back to back
    assert len(lint.check(["v_pk_add_f32 v[10:11], v[1:2], v[3:4]", "v_mov_b32_e32 v5, v6", DPP])) == 1     # one wait state
    assert lint.check(["v_pk_add_f32 v[10:11], v[1:2], v[3:4]", "s_nop 1", DPP]) == []                  # two wait states
    assert lint.check(["v_pk_add_f32 v[10:11], v[1:2], v[3:4]", "v_mov_b32_e32 v5, v6", "s_nop 0", DPP]) == []
    assert lint.check(["v_pk_add_f32 v[12:13], v[1:2], v[3:4]", DPP]) == []                             # another register
    # only src0 goes through DPP: a fresh src1 is ordinary forwarding
    assert lint.check(["v_mov_b32_e32 v30, v6", "v_add_f32_dpp v20, v10, v30 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"]) == []


def test_comb_walker_has_no_dpp_source_hazard():
    assert lint.main([]) == 0
