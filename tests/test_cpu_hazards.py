"""tools/check_isa_hazards.py: the hand-written gfx950 ISA of the path rests on wait states inserted by hand (the assembler pads nothing inside an asm
statement, LLVM's hazard recognizer does not look into one) and the parity tests cannot be relied on to see a missing one.  The lint must (a) pass on the
shipped library, (b) flag every rule's hazard in a deliberately broken sequence and (c) accept the same sequence with the wait states in -- over straight
code, over a taken branch and over a loop's back edge.  Snippets are assembled for gfx950 with the image's clang; no GPU is involved."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_isa_hazards as H  # noqa: E402

CLANG = "/opt/rocm/lib/llvm/bin/clang"


def lint_snippet(tmp_path, body, name="k"):
    src = tmp_path / (name + ".s")
    src.write_text(f"\t.text\n\t.globl {name}\n\t.p2align 8\n\t.type {name},@function\n{name}:\n{body}\n\ts_endpgm\n")
    obj = tmp_path / (name + ".o")
    subprocess.check_call([CLANG, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(src), "-o", str(obj)])
    text = subprocess.check_output([H.OBJDUMP, "-d", str(obj)], text=True)
    return sorted({f[1] for f in H.lint_text(text)}), H.lint_text(text)


# rule -> (broken sequence, the same with the wait states in)
CASES = {
    "DPP_VGPR": ("\tv_mov_b32 v1, v2\n\ts_nop 0\n\tv_max_i32_dpp v1, v1, v1 row_shr:1 row_mask:0xf bank_mask:0xf",
                 "\tv_mov_b32 v1, v2\n\ts_nop 1\n\tv_max_i32_dpp v1, v1, v1 row_shr:1 row_mask:0xf bank_mask:0xf"),
    "DPP_EXEC": ("\tv_cmpx_lt_i32 vcc, v1, v2\n\ts_nop 3\n\tv_mov_b32_dpp v3, v4 row_shr:1 row_mask:0xf bank_mask:0xf",
                 "\tv_cmpx_lt_i32 vcc, v1, v2\n\ts_nop 4\n\tv_mov_b32_dpp v3, v4 row_shr:1 row_mask:0xf bank_mask:0xf"),
    "LANE_SEL": ("\tv_readfirstlane_b32 s4, v1\n\ts_nop 2\n\tv_readlane_b32 s5, v2, s4",
                 "\tv_readfirstlane_b32 s4, v1\n\ts_nop 3\n\tv_readlane_b32 s5, v2, s4"),
    "VMEM_SGPR": ("\tv_readfirstlane_b32 s4, v1\n\tv_readfirstlane_b32 s5, v2\n\ts_nop 2\n\tglobal_load_dword v3, v4, s[4:5]\n\ts_waitcnt vmcnt(0)",
                  "\tv_readfirstlane_b32 s4, v1\n\tv_readfirstlane_b32 s5, v2\n\ts_nop 4\n\tglobal_load_dword v3, v4, s[4:5]\n\ts_waitcnt vmcnt(0)"),
    "DSTSEL": ("\tv_sub_u16_sdwa v1, v2, v3 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0\n\tv_add_u32 v4, v1, v1",
               "\tv_sub_u16_sdwa v1, v2, v3 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0\n\tv_mov_b32 v9, v8\n\tv_add_u32 v4, v1, v1"),
    "TRANS": ("\tv_rcp_f32 v1, v2\n\tv_mul_f32 v3, v1, v1", "\tv_rcp_f32 v1, v2\n\ts_nop 0\n\tv_mul_f32 v3, v1, v1"),
    "LANE_EXEC": ("\tv_cmpx_lt_i32 vcc, v1, v2\n\ts_nop 2\n\tv_readfirstlane_b32 s4, v3", "\tv_cmpx_lt_i32 vcc, v1, v2\n\ts_nop 3\n\tv_readfirstlane_b32 s4, v3"),
    "LANE_VGPR": ("\tv_mov_b32 v1, v2\n\tv_readlane_b32 s4, v1, 63", "\tv_mov_b32 v1, v2\n\ts_nop 0\n\tv_readlane_b32 s4, v1, 63"),
    "SGPR_VALU": ("\tv_cmp_eq_u32 vcc, v1, v2\n\ts_nop 0\n\tv_cndmask_b32 v3, v4, v5, vcc", "\tv_cmp_eq_u32 vcc, v1, v2\n\ts_nop 1\n\tv_cndmask_b32 v3, v4, v5, vcc"),
    "STORE_WAR": ("\tglobal_store_dwordx4 v[0:1], v[4:7], off\n\tv_mov_b32 v5, 0\n\ts_waitcnt vmcnt(0)", "\tglobal_store_dwordx4 v[0:1], v[4:7], off\n\ts_nop 0\n\tv_mov_b32 v5, 0\n\ts_waitcnt vmcnt(0)"),
    "DIV_FMAS": ("\tv_cmp_eq_u32 vcc, v1, v2\n\ts_nop 2\n\tv_div_fmas_f32 v3, v4, v5, v6", "\tv_cmp_eq_u32 vcc, v1, v2\n\ts_nop 3\n\tv_div_fmas_f32 v3, v4, v5, v6"),
    "M0_LDS": ("\ts_mov_b32 m0, s4\n\tglobal_load_lds_dwordx4 v[0:1], off\n\ts_waitcnt vmcnt(0)", "\ts_mov_b32 m0, s4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 v[0:1], off\n\ts_waitcnt vmcnt(0)"),
}


@pytest.mark.parametrize("rule", sorted(CASES))
def test_each_rule_flags_the_broken_sequence_and_accepts_the_padded_one(rule, tmp_path):
    broken, padded = CASES[rule]
    rules_hit, detail = lint_snippet(tmp_path, broken, "bad")
    assert rule in rules_hit, (rule, detail)
    f = [d for d in detail if d[1] == rule][0]
    assert f[4] < f[5] == H.RULES[rule]                            # found fewer wait states than the rule asks for
    rules_hit, detail = lint_snippet(tmp_path, padded, "good")
    assert rule not in rules_hit, (rule, [H.describe(d) for d in detail])


def test_hazards_are_followed_over_branches_and_back_edges(tmp_path):
    # the producer sits before a taken branch, the consumer at its target: the branch itself is the only wait state in between
    over_branch = "\tv_cmp_eq_u32 vcc, v1, v2\n\ts_branch Ltarget\n\tv_mov_b32 v9, v9\n\tv_mov_b32 v9, v9\nLtarget:\n\tv_cndmask_b32 v3, v4, v5, vcc"
    hit, _ = lint_snippet(tmp_path, over_branch, "br")
    assert hit == ["SGPR_VALU"]
    # ... and a loop whose last instruction writes what its first one reads through DPP (straight code would look clean: the write comes AFTER the read)
    loop = "Lloop:\n\tv_max_i32_dpp v1, v1, v1 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_add_u32 s4, s4, -1\n\ts_cmp_lg_u32 s4, 0\n\tv_mov_b32 v1, v2\n\ts_cbranch_scc1 Lloop"
    hit, detail = lint_snippet(tmp_path, loop, "loop")
    assert hit == ["DPP_VGPR"] and detail[0][4] == 1               # one wait state (the branch) where two are needed
    fixed = loop.replace("\tv_mov_b32 v1, v2\n", "\tv_mov_b32 v1, v2\n\ts_nop 0\n")
    hit, _ = lint_snippet(tmp_path, fixed, "loopok")
    assert hit == []
    # a scalar write of the same register in between ends the dependency (the nearest writer is not a VALU)
    killed = "\tv_readfirstlane_b32 s4, v1\n\ts_mov_b32 s4, 7\n\tv_readlane_b32 s5, v2, s4"
    hit, _ = lint_snippet(tmp_path, killed, "kill")
    assert hit == []


def test_the_shipped_library_has_no_open_hazard():
    lib = os.path.join(ROOT, "minimap2-fpga_amd", "libmm2chain_hip.so")
    findings, n_kernels, n_insns = H.lint_library(lib)
    assert n_kernels > 50 and n_insns > 100000                      # every code object of the library was read
    assert not findings, "\n".join(H.describe(f) for f in findings[:10])


def test_a_broken_build_of_the_real_loop_is_caught(tmp_path):
    """The sequence the lint found in round 5, as it stood in csrc/radix_replay.h before the fix (a compare into VCC followed at once by the subtract-with-borrow that
    reads it), inside the loop it came from: must be flagged; the fixed order must not."""
    before = ("Lstep:\n\tv_cmp_eq_u32 vcc, v44, v40\n\ts_waitcnt lgkmcnt(0)\n\tds_write_b64 v40, v[50:51]\n\tv_cndmask_b32 v45, v48, v50, vcc\n\tv_cndmask_b32 v46, v49, v51, vcc\n"
              "\tv_cmp_eq_u32 vcc, v44, v52\n\tv_subb_co_u32 v54, vcc, v45, 0, vcc\n\tv_lshlrev_b32 v54, 2, v54\n\ts_cbranch_vccnz Lstep")
    hit, detail = lint_snippet(tmp_path, before, "before")
    assert "SGPR_VALU" in hit and any(d[4] == 0 for d in detail)
    after = before.replace("\tv_cmp_eq_u32 vcc, v44, v52\n", "\tv_cmp_eq_u32 vcc, v44, v52\n\tv_sub_u32 v55, v41, v57\n\ts_nop 0\n")
    hit, detail = lint_snippet(tmp_path, after, "after")
    assert hit == [], [H.describe(d) for d in detail]
