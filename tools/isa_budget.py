#!/usr/bin/env python3
"""Instruction budget of the hand-written anchor loop of chain_dp_tile (csrc/chain_dp_tile.h, MM2C_SCAN_TILE_ASM): expands the string
macros of the header, cuts the sequence at its labels and counts the instructions of every block by issue class (the classes of
tools/ubench/issue_rate.hip: plain VALU / VALU that involves the scalar side / SALU / branch / LDS / VMEM / waits).  The blocks are then
summed along the paths an anchor takes, so that the per-anchor PMC figures of profiles/r2_mixed.md can be decomposed.
    python tools/isa_budget.py [--tab] > profiles/r2_isa_budget.md"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, "minimap2-fpga_amd/csrc/chain_dp_tile.h")).read()

TOKEN = re.compile(r'"((?:[^"\\]|\\.)*)"|(MM2C_[A-Z0-9_]+)(\(([^()]*)\))?|\b(SCORE|ADDF|SEG_RD|SEG_LK|SEG_HF|SEG_TAIL|SEG_END|SEG_DONE|XQ1|NEXT_XQ|RFILTER|OLDADDR|BACK|OWNFILTER|FARFILTER|RDXQ|R|CTRL|X|Q|D|S0SEL|S1SEL|V)\b')


def logical_defines(src):
    out, cur = {}, None
    for line in src.split("\n"):
        if cur is not None:
            cur[1].append(line.rstrip("\\"))
            if not line.rstrip().endswith("\\"):
                out[cur[0]] = (cur[2], " ".join(cur[1])); cur = None
            continue
        m = re.match(r"#define (MM2C_[A-Z0-9_]+)(\(([^)]*)\))?\s(.*)", line)
        if m:
            params = [p.strip() for p in m.group(3).split(",")] if m.group(3) else []
            body = m.group(4)
            if body.rstrip().endswith("\\"):
                cur = (m.group(1), [body.rstrip("\\")], params)
            else:
                out[m.group(1)] = (params, body)
    return out


DEFS = logical_defines(SRC)


def expand(body, env):
    text = ""
    for m in TOKEN.finditer(body):
        if m.group(1) is not None:
            text += m.group(1).replace("\\n", "\n").replace("\\t", " ")
        elif m.group(2):
            name = m.group(2)
            if name not in DEFS:
                continue
            params, b = DEFS[name]
            args = []
            if m.group(4) is not None:
                args = [a.strip() for a in re.findall(r'"(?:[^"\\]|\\.)*"|[A-Za-z_]+', m.group(4))]
            sub = dict(env)
            for p, a in zip(params, args):
                sub[p] = a if a.startswith('"') else env.get(a, '""')
            text += expand(b, sub)
        else:
            v = env.get(m.group(5))
            if v is None:
                continue
            text += expand(v, env) if not v.startswith('"') else v[1:-1].replace("\\n", "\n").replace("\\t", " ")
    return text


def classify(ins):
    op = ins.split()[0]
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop")):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("v_"):
        scalar_side = (op.startswith(("v_cmp", "v_readlane", "v_writelane", "v_readfirstlane", "v_mbcnt")) or "_dpp" in op or "row_" in ins or "wave_" in ins
                       or (op.startswith("v_cndmask") and "e64" in op))
        return "valu_s" if scalar_side else "valu"
    return "other"


def blocks(tab, far=False, wide=False):
    params, body = DEFS["MM2C_SCAN_TILE_ASM"]
    body = body[body.index("asm volatile("):]
    body = body[:body.index(": [best]")].replace("SEG_END(SCORE, FARFILTER)", "SEG_END").replace("MM2C_READ_ANCHOR(SEG_RD, RDXQ)", "MM2C_READ_ANCHOR")
    v = "FAR" if far else "LEAN"
    r = "_W" if wide else "_C"                       # the 32-bit x / q ring or the compact one (the default of the library where it applies)
    env = {"SCORE": "MM2C_SCORE_TAB" if tab else "MM2C_SCORE_CMP", "ADDF": "MM2C_ADDF_TAB" if tab else "MM2C_ADDF_CMP",
           "SEG_RD": "MM2C_RD_" + v, "SEG_LK": "MM2C_LK_" + v, "SEG_HF": "MM2C_HF_" + v, "SEG_TAIL": "MM2C_TAIL_" + v, "SEG_END": "MM2C_END_" + v, "SEG_DONE": '""',
           "XQ1": "MM2C_XQ1" + r, "NEXT_XQ": "MM2C_NEXT_XQ" + r, "RFILTER": "MM2C_RFILTER" + r, "OLDADDR": "MM2C_OLDADDR" + r, "BACK": "MM2C_BACK" + r,
           "OWNFILTER": "MM2C_OWNFILTER" + r, "FARFILTER": "MM2C_FARFILTER" + r, "RDXQ": "MM2C_RDXQ" + r}
    text = expand(body, env)
    out, cur = [], ("entry", [])
    for line in text.split("\n"):
        line = line.strip()
        if not line:
            continue
        m = re.match(r"(L[a-z0-9]+)_%=:", line)
        if m:
            out.append(cur); cur = (m.group(1), [])
        else:
            cur[1].append(line)
    out.append(cur)
    return out


CLASSES = ("valu", "valu_s", "salu", "branch", "lds", "vmem", "wait")


def count(instrs):
    c = dict.fromkeys(CLASSES, 0)
    for i in instrs:
        c[classify(i)] += 1
    return c


def cut(instrs, until=None, after=None):
    """instructions up to and including the first one that starts with `until`, or those behind the first one starting with `after`"""
    if until:
        for k, i in enumerate(instrs):
            if i.startswith(until):
                return instrs[:k + 1]
    if after:
        for k, i in enumerate(instrs):
            if i.startswith(after):
                return instrs[k + 1:]
    return instrs


if __name__ == "__main__":
    tab, far, wide = "--tab" in sys.argv, "--far" in sys.argv, "--wide" in sys.argv
    B = dict(blocks(tab, far, wide))
    order = [n for n, _ in blocks(tab, far, wide)]
    print(f"# Instruction budget of the hand-written anchor loop (`scan_tile_asm_{'tab' if tab else 'cmp'}{'_far' if far else ''}{'' if wide else '_c'}`: {'32-bit' if wide else 'compact'} x / q ring), from `tools/isa_budget.py`\n")
    print("Classes as measured by `tools/ubench/issue_rate.hip` (`profiles/r2_issue_rate.md`): plain VALU ≈0.84 per SIMD and ns; VALU that involves the scalar side")
    print("(`v_cmp`, `v_readlane`/`v_writelane`, DPP, `v_cndmask` with an SGPR mask, `v_mbcnt`) and SALU ≈0.55; `ds_read` 0.29.\n")
    print("## Blocks between labels (straight-line instruction counts)\n")
    print("| block | plain VALU | scalar-side VALU | SALU | branch | LDS | VMEM | waitcnt / nop |")
    print("|---|---|---|---|---|---|---|---|")
    for n in order:
        c = count(B[n])
        print(f"| `{n}` | " + " | ".join(str(c[k]) for k in CLASSES) + " |")

    def path(*parts):
        tot = dict.fromkeys(CLASSES, 0)
        for p in parts:
            for k, v in count(p).items():
                tot[k] += v
        return tot

    fixed = path(B["Lk"][:B["Lk"].index(next(i for i in B["Lk"] if i.startswith("s_cbranch_scc0 Lloop")))+1], B["Ldone"])
    own = path(cut(cut(B["Lk"], after="s_cbranch_scc0 Lloop"), until="s_cbranch_scc0 Lloop"))
    empty = path(cut(B["Lloop"], until="s_cbranch_vccz"))
    # chunk with a surviving lane from the f / p ring, fold A (no lane beats the running best, no marked lane), back to the loop
    if far:
        scored = path(cut(B["Lloop"], after="s_cbranch_vccz"), cut(B["Lold"], until="s_branch Lhf"), cut(B["Lhf"], until="s_cbranch_scc0 Lmk"),
                      cut(B["Lmk"], until="s_cbranch_scc0 Lret"), cut(B["Lret"], until="s_cbranch_scc0 Lloop"))
    else:
        scored = path(cut(B["Lloop"], after="s_cbranch_vccz"), cut(B["Lold"], until="s_branch Lhf"), B["Lhf"], cut(B["Lmk"], until="s_cbranch_scc0 Lloop"))
    b0 = path(cut(B["Limp"], until="s_branch Lret"))
    b1 = path(cut(B["Limp"], until="s_cbranch_scc0 Lslow"), cut(B["Lslow"], until="s_branch Lret"))
    b2 = path(cut(B["Limp"], until="s_cbranch_vccnz Lslow2"), B["Lslow2"], cut(B["Lslow"], until="s_cbranch_scc1 Lb2"), cut(B["Lb2"], until="s_branch Ltk"), cut(B["Ltk"], until="s_ff1"), B["Laf"][:2])
    print("\n## Paths\n")
    print("| path | plain VALU | scalar-side VALU | SALU | branch | LDS | VMEM | waitcnt / nop |")
    print("|---|---|---|---|---|---|---|---|")
    for name, c in (("per anchor, fixed: scalars by `v_readfirstlane` under a one-hot exec, first x / q request, commit by `v_mov` under the same mask", fixed),
                    ("+ own-tile chunk, no lane passes the filters", own),
                    ("per older tile, no lane passes (`Lloop` … `s_cbranch_vccz`)", empty),
                    ("+ a lane passes, f/p from the LDS ring, stamps, score, fold A (no lane beats the best)", scored),
                    ("+ fold B0 (the first surviving lane is the only new maximum: closed form)", b0),
                    ("+ fold B1 (one candidate that is not the first surviving lane, no marks, no skips so far)", b1),
                    ("+ fold B2 (prefix max by DPP, closed-form skip counter, take the result)", b2)):
        print(f"| {name} | " + " | ".join(str(c[k]) for k in CLASSES) + " |")
