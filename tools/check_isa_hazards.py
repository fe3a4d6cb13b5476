#!/usr/bin/env python3
"""Hazard lint for the hand-written gfx950 ISA of the path (csrc/chain_dp_tile.h, csrc/chain_dp_coop.h, csrc/radix_replay.h and every other asm
statement of the library).

The assembler pads nothing inside an `asm` statement and LLVM's hazard recognizer does not look into one (/opt/skills/guides/cdna_hip_programming.md
section 5.7 item 2), so the wait states the hardware does not interlock are inserted by hand -- and a missing one gives wrong values on some waves of some
launches, which the parity tests cannot be relied on to see (round 3's advisor found one by reading).  This tool disassembles every gfx950 code object
bundled in the library (llvm-objdump -d), rebuilds the control flow of each kernel (fall-through + branch targets, the hand-written loops' local labels
included) and, for every consumer instruction of a rule below, walks BACKWARDS over all paths for the nearest producer, counting wait states
(one per instruction, s_nop N = N + 1).  Whole kernels are linted, compiler code and hand-written code alike: the compiler's part is expected to be
clean (its hazard recognizer ran), which is also what validates the rule set -- a rule that is too strict shows up as findings in kernels that have no
asm statement at all.

Rules = the software-managed dependencies of gfx940 / gfx950.  Source: the CDNA3 instruction-set guide, section 4.5 "Manually Inserted Wait States
(NOPs)", table "Required User-Inserted Wait States" (the table the kernel sources' header comments cite by hand; the guide's companion
cdna_asm_programming.md calls it section 4.1 Table 11), with the gfx940 additions that LLVM's GCNHazardRecognizer enforces for compiler code (constant names
in brackets).  The documents are not available offline in the build image: rows are cited by their first-column text.

  id        producer -> consumer                                                                       states   row / LLVM constant
  DPP_VGPR  VALU writes a VGPR -> VALU DPP reads it                                                       2     "VALU writes VGPR -> VALU DPP reads that VGPR" [DppVgprWaitStates]
  DPP_EXEC  VALU writes EXEC -> VALU DPP                                                                  5     "VALU writes EXEC -> VALU DPP op" [DppExecWaitStates]
  LANE_SEL  VALU writes an SGPR / VCC -> v_readlane / v_writelane with it as lane select                  4     "VALU writes SGPR/VCC (readlane, cmp, add/sub, div_scale) -> V_{READ,WRITE}LANE using that SGPR/VCC as the lane select" [RWLaneWaitStates]
  VMEM_SGPR VALU writes an SGPR -> VMEM reads it (address, offset, descriptor)                            5     "VALU writes SGPR -> VMEM reads that SGPR" [VmemSgprWaitStates]
  DSTSEL    VALU with SDWA dst_sel != DWORD writes a VGPR -> VALU reads it                                1     gfx940 [Shift16DefWaitstates, hasDstSelForwardingHazard]
  TRANS     VALU transcendental writes a VGPR -> non-transcendental VALU reads it                         1     gfx940 [TransDefWaitstates, hasTransForwardingHazard]
  LANE_EXEC VALU writes EXEC -> v_readlane / v_readfirstlane / v_writelane                                4     gfx940 [VALUWriteEXECRWLane]
  LANE_VGPR VALU writes a VGPR -> v_readlane / v_readfirstlane reads it                                   1     gfx940 [VALUWriteVGPRReadlaneRead]
  SGPR_VALU VALU writes an SGPR / VCC -> VALU reads it as an operand (v_cndmask mask, carry-in, scalar)   2     gfx940 [VALUWriteSGPRVALURead]
  STORE_WAR VMEM store of more than 64 bits -> VALU overwrites its data VGPRs                             1     "VMEM store more than 64 bits -> write of the VGPRs holding the writedata" [VmemStoreHazWaitStates]; the guide's practice is s_nop 1
  DIV_FMAS  VALU writes VCC -> v_div_fmas                                                                 4     "VALU writes VCC (including v_div_scale) -> V_DIV_FMAS" [DivFMasWaitStates]
  VCCZ      VALU writes VCC / EXEC -> VALU reads vccz / execz as data                                     5     "VALU that sets VCC or EXEC followed by a VALU that uses EXECZ or VCCZ as a data source"
  M0_LDS    SALU writes M0 -> LDS-DMA (buffer / global load ... lds), ds add-tid, s_movrel                1     "SALU writes M0 -> GDS, S_SENDMSG or LDS add-TID instruction, buffer_store_LDS_dword, scratch or VINTERP"
  SWAP      VALU writes a VGPR -> v_permlane16_swap / v_permlane32_swap operand                           2     gfx950, /opt/skills/guides/cdna_hip_programming.md (the v_permlane*_swap rule)

Usage: check_isa_hazards.py libmm2chain_hip.so [-v] [--kernels REGEX]       exit 0 = no finding
The Makefile runs it after linking, next to check_lds_layout.py; tests/test_cpu_hazards.py runs it on the shipped library and on deliberately broken
sequences (assembled with the image's clang).
"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from check_lds_layout import code_objects  # noqa: E402

OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")

RULES = {"DPP_VGPR": 2, "DPP_EXEC": 5, "LANE_SEL": 4, "VMEM_SGPR": 5, "DSTSEL": 1, "TRANS": 1, "LANE_EXEC": 4, "LANE_VGPR": 1, "SGPR_VALU": 2,
         "STORE_WAR": 1, "DIV_FMAS": 4, "VCCZ": 5, "M0_LDS": 1, "SWAP": 2}

REG_RE = re.compile(r"\b([vsa])(\d+)\b|\b([vsa])\[(\d+):(\d+)\]|\b(vcc|exec)(_lo|_hi)?\b|\b(m0)\b|\b(src_vccz|src_execz|vccz|execz)\b")
TRANS_RE = re.compile(r"^v_(exp|log|rcp|rcp_iflag|rsq|sqrt|sin|cos)_(f16|f32|f64|legacy_f32)")
TWO_DST_RE = re.compile(r"^v_((add|sub|subrev|addc|subb|subbrev)_co_u32|div_scale_f(32|64)|mad_(u64_u32|i64_i32))")
SALU_NO_DST = re.compile(r"^s_(cmp|cmpk|bitcmp|branch|cbranch|waitcnt|nop|endpgm|barrier|setprio|sleep|setpc|sendmsg|sethalt|trap|icache_inv|dcache|"
                         r"setreg|set_gpr_idx|incperflevel|decperflevel|ttracedata|code_end|wakeup|rfe|setvskip|version|clause|delay|wait_)")


def regs_of(text):
    """register tokens named in an operand string -> set of names ('v12', 's4', 'vcc_lo', 'exec_hi', 'm0', 'vccz')"""
    out = set()
    for m in REG_RE.finditer(text):
        if m.group(1):
            out.add(m.group(1) + m.group(2))
        elif m.group(3):
            out.update(m.group(3) + str(k) for k in range(int(m.group(4)), int(m.group(5)) + 1))
        elif m.group(6):
            out.update([m.group(6) + m.group(7)] if m.group(7) else [m.group(6) + "_lo", m.group(6) + "_hi"])
        elif m.group(8):
            out.add("m0")
        elif m.group(9):
            out.add(m.group(9).replace("src_", ""))
    return out


class Insn:
    __slots__ = ("addr", "mnem", "ops", "mods", "text", "defs", "uses", "kind", "ws", "target", "dpp", "dstsel", "trans", "lane_sel", "wide_store_data", "uses_m0_lds")

    def __init__(self, addr, text, target):
        self.addr, self.text, self.target = addr, text, target
        parts = text.split(None, 1)
        self.mnem = parts[0]
        rest = parts[1] if len(parts) > 1 else ""
        raw = [o.strip() for o in rest.split(",")] if rest else []
        self.ops, self.mods = [], ""
        for k, o in enumerate(raw):
            if k == len(raw) - 1 and " " in o and not o.startswith("hwreg") and not o.startswith("vmcnt") and not o.startswith("lgkmcnt") and not o.startswith("expcnt"):
                first, mods = o.split(None, 1)
                self.ops.append(first)
                self.mods = mods
            else:
                self.ops.append(o)
        m = self.mnem
        self.kind = ("valu" if m.startswith("v_") else "vmem" if m.startswith(("buffer_", "global_", "flat_", "scratch_", "tbuffer_", "image_")) else
                     "lds" if m.startswith("ds_") else "smem" if m.startswith(("s_load", "s_buffer_load", "s_store", "s_buffer_store", "s_scratch", "s_atomic", "s_dcache", "s_memtime", "s_memrealtime")) else
                     "salu" if m.startswith("s_") else "other")
        self.ws = 1
        if m == "s_nop":
            try:
                self.ws = int(self.ops[0], 0) + 1
            except (ValueError, IndexError):
                self.ws = 1
        self.dpp = "_dpp" in m
        self.trans = bool(TRANS_RE.match(m))
        ds = re.search(r"dst_sel:(\w+)", self.mods)
        self.dstsel = "_sdwa" in m and ds is not None and ds.group(1) != "DWORD"
        self.lane_sel = set()
        self.wide_store_data = set()
        self.uses_m0_lds = False
        self.defs, self.uses = set(), set()
        self._classify()

    def _classify(self):
        m, ops = self.mnem, self.ops
        opregs = [regs_of(o) for o in ops]
        alluse = lambda idx: set().union(*[opregs[k] for k in idx]) if idx else set()   # noqa: E731
        n = len(ops)
        if self.kind == "valu":
            if m.startswith("v_nop"):
                return
            if m.startswith("v_swap_b32") or "_swap_b32" in m:
                self.defs = alluse(range(min(2, n))); self.uses = alluse(range(n)); return
            ndst = 2 if TWO_DST_RE.match(m) else 1
            if m.startswith(("v_cmpx_",)):
                self.defs = alluse(range(min(1, n))) | {"exec_lo", "exec_hi"}
                self.uses = alluse(range(1, n))
            else:
                self.defs = alluse(range(min(ndst, n)))
                self.uses = alluse(range(ndst, n))
            if m.startswith(("v_writelane_b32",)):
                self.uses |= self.defs                           # the other lanes keep their value
            if self.dpp or "_sdwa" in m:
                self.uses |= {r for r in self.defs if r[0] == "v"}   # old / preserved halves of the destination
            if m.startswith(("v_readlane_b32", "v_writelane_b32")) and n >= 3:
                self.lane_sel = {r for r in opregs[2] if r[0] == "s" or r.startswith("vcc") or r == "m0"}
            if m.startswith("v_div_fmas"):
                self.uses |= {"vcc_lo", "vcc_hi"}
            if m.startswith(("v_mac_", "v_fmac_", "v_madak", "v_madmk", "v_dot")) or m.startswith("v_pk_fmac"):
                self.uses |= {r for r in self.defs if r[0] == "v"}
        elif self.kind == "salu":
            if SALU_NO_DST.match(m):
                self.uses = alluse(range(n)); return
            self.defs = alluse(range(min(1, n)))
            self.uses = alluse(range(1, n))
            if "saveexec" in m:
                self.defs |= {"exec_lo", "exec_hi"}; self.uses |= {"exec_lo", "exec_hi"}
            if m.startswith(("s_movrel", "s_cmovk")):
                pass
        elif self.kind in ("vmem", "lds", "smem"):
            store = ("store" in m) or m.startswith(("ds_write", "ds_gws")) or (m.startswith("ds_") and ("_add" in m or "_max" in m or "_min" in m or "_or" in m or "_and" in m or "_xor" in m or "_sub" in m or "_inc" in m or "_dec" in m) and "_rtn" not in m)
            atomic_noret = self.kind == "vmem" and "atomic" in m and "glc" not in self.mods and " sc0" not in (" " + self.mods)
            if store or atomic_noret:
                self.uses = alluse(range(n))
                if self.kind == "vmem" and re.search(r"dwordx[34]$", m):
                    # data VGPRs: global_store_dwordx4 v[addr], v[data:..], s[..]|off ; buffer_store_dwordx4 v[data..], v_off, s[rsrc], soffset
                    didx = 0 if m.startswith(("buffer_", "tbuffer_")) else 1
                    if didx < n:
                        soff_is_reg = m.startswith(("buffer_", "tbuffer_")) and n >= 4 and bool(regs_of(ops[3]))
                        if not soff_is_reg:
                            self.wide_store_data = {r for r in opregs[didx] if r[0] == "v"}
            else:
                self.defs = alluse(range(min(1, n)))
                self.uses = alluse(range(1, n))
            if self.kind == "vmem" and (re.search(r"\blds\b", self.mods) or "_lds_" in m):
                self.uses_m0_lds = True; self.defs = set(); self.uses = alluse(range(n))
            if m.startswith("ds_") and "addtid" in m:
                self.uses_m0_lds = True
        if m.startswith("s_movrel"):
            self.uses_m0_lds = True


def disassemble(elf_bytes):
    with tempfile.NamedTemporaryFile(suffix=".elf", delete=False) as fh:
        fh.write(elf_bytes); path = fh.name
    try:
        return subprocess.check_output([OBJDUMP, "-d", path], text=True, errors="replace")
    finally:
        os.unlink(path)


HEAD_RE = re.compile(r"^([0-9a-f]{8,16}) <(.+)>:\s*$")
LINE_RE = re.compile(r"^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):\s*(.*)$")


def parse_kernels(asm_text):
    """{kernel name: [Insn]} -- hand-written local labels (names that start with L) stay inside the kernel they were written in"""
    labels, lines, kernels, cur = {}, asm_text.splitlines(), {}, None
    for ln in lines:
        h = HEAD_RE.match(ln)
        if h:
            labels[h.group(2)] = int(h.group(1), 16)
    for ln in lines:
        h = HEAD_RE.match(ln)
        if h:
            if not h.group(2).startswith("L") or cur is None:
                cur = h.group(2); kernels[cur] = []
            continue
        m = LINE_RE.match(ln)
        if not m or cur is None:
            continue
        text, addr, tail = m.group(1), int(m.group(2), 16), m.group(3)
        target = None
        mn = text.split(None, 1)[0]
        if mn.startswith(("s_branch", "s_cbranch")) and not mn.startswith("s_cbranch_g_fork") and not mn.startswith("s_cbranch_join"):
            # SOPP: the low 16 bits of the instruction word are the signed dword offset from the next instruction (the operand text may be a label objdump knows,
            # a label it does not show -- two symbols at one address -- or the raw number)
            w = re.match(r"([0-9A-Fa-f]{8})\b", tail)
            if w:
                imm = int(w.group(1), 16) & 0xffff
                target = addr + 4 + 4 * (imm - 0x10000 if imm & 0x8000 else imm)
        kernels[cur].append(Insn(addr, text, target))
    return kernels


def lint_kernel(insns):
    """[(rule, producer Insn, consumer Insn, states found, states needed)]"""
    idx = {i.addr: k for k, i in enumerate(insns)}
    preds = [[] for _ in insns]
    for k, i in enumerate(insns):
        falls = not (i.mnem in ("s_branch", "s_endpgm", "s_setpc_b64", "s_swappc_b64", "s_rfe_b64") )
        if falls and k + 1 < len(insns):
            preds[k + 1].append(k)
        if i.target is not None and i.target in idx:
            preds[idx[i.target]].append(k)
    findings = []

    def search(k_cons, need, is_producer, kills):
        """nearest producers within `need` wait states before insns[k_cons] over all paths; kills(insn): a later non-hazardous write of the same resource ends the path"""
        hits, stack, seen = [], [(p, 0) for p in preds[k_cons]], {}
        while stack:
            k, acc = stack.pop()
            if acc >= need or seen.get(k, 1 << 30) <= acc:
                continue
            seen[k] = acc
            x = insns[k]
            if is_producer(x):
                hits.append((x, acc)); continue
            if kills(x):
                continue
            for p in preds[k]:
                stack.append((p, acc + x.ws))
        return hits

    for kc, c in enumerate(insns):
        checks = []          # (rule, is_producer, kills)
        if c.kind == "valu":
            vuse = {r for r in c.uses if r[0] == "v"}
            suse = {r for r in c.uses if r[0] == "s" or r.startswith("vcc_")}
            if c.dpp:
                for r in vuse:
                    checks.append(("DPP_VGPR", lambda x, r=r: x.kind == "valu" and r in x.defs, lambda x, r=r: r in x.defs))
                checks.append(("DPP_EXEC", lambda x: x.kind == "valu" and "exec_lo" in x.defs, lambda x: "exec_lo" in x.defs or "exec_hi" in x.defs))
            for r in c.lane_sel:
                checks.append(("LANE_SEL", lambda x, r=r: x.kind == "valu" and r in x.defs, lambda x, r=r: r in x.defs))
            if c.mnem.startswith(("v_readlane_b32", "v_readfirstlane_b32", "v_writelane_b32")):
                checks.append(("LANE_EXEC", lambda x: x.kind == "valu" and "exec_lo" in x.defs, lambda x: "exec_lo" in x.defs or "exec_hi" in x.defs))
                if not c.mnem.startswith("v_writelane"):
                    for r in {q for q in regs_of(c.ops[1]) if q[0] == "v"} if len(c.ops) > 1 else ():
                        checks.append(("LANE_VGPR", lambda x, r=r: x.kind == "valu" and r in x.defs, lambda x, r=r: r in x.defs))
            for r in vuse:
                checks.append(("DSTSEL", lambda x, r=r: x.kind == "valu" and x.dstsel and r in x.defs, lambda x, r=r: r in x.defs))
                if not c.trans:
                    checks.append(("TRANS", lambda x, r=r: x.kind == "valu" and x.trans and r in x.defs, lambda x, r=r: r in x.defs))
            for r in suse - c.lane_sel:
                checks.append(("SGPR_VALU", lambda x, r=r: x.kind == "valu" and r in x.defs, lambda x, r=r: r in x.defs))
            if c.mnem.startswith("v_div_fmas"):
                checks.append(("DIV_FMAS", lambda x: x.kind == "valu" and "vcc_lo" in x.defs, lambda x: "vcc_lo" in x.defs))
            if "vccz" in c.uses or "execz" in c.uses:
                checks.append(("VCCZ", lambda x: x.kind == "valu" and ("vcc_lo" in x.defs or "exec_lo" in x.defs), lambda x: False))
            if c.mnem.startswith(("v_permlane16_swap", "v_permlane32_swap")):
                for r in {q for q in (c.defs | c.uses) if q[0] == "v"}:
                    checks.append(("SWAP", lambda x, r=r: x.kind == "valu" and r in x.defs, lambda x, r=r: r in x.defs))
            for r in {q for q in c.defs if q[0] == "v"}:
                checks.append(("STORE_WAR", lambda x, r=r: r in x.wide_store_data, lambda x: False))
        elif c.kind == "vmem":
            for r in {q for q in c.uses if q[0] == "s" or q.startswith("vcc_")}:
                checks.append(("VMEM_SGPR", lambda x, r=r: x.kind == "valu" and r in x.defs, lambda x, r=r: r in x.defs))
        if c.uses_m0_lds:
            checks.append(("M0_LDS", lambda x: x.kind == "salu" and "m0" in x.defs, lambda x: "m0" in x.defs))
        for rule, is_p, kills in checks:
            for p, acc in search(kc, RULES[rule], is_p, kills):
                findings.append((rule, p, c, acc, RULES[rule]))
    # one line per (rule, producer, consumer)
    uniq = {}
    for f in findings:
        key = (f[0], f[1].addr, f[2].addr)
        if key not in uniq or f[3] < uniq[key][3]:
            uniq[key] = f
    return sorted(uniq.values(), key=lambda f: f[2].addr)


def lint_text(asm_text, kernel_filter=None):
    out = []
    for name, insns in parse_kernels(asm_text).items():
        if kernel_filter and not re.search(kernel_filter, name):
            continue
        for f in lint_kernel(insns):
            out.append((name,) + f)
    return out


def lint_library(path, kernel_filter=None, verbose=False):
    data = open(path, "rb").read()
    findings, n_kernels, n_insns = [], 0, 0
    for _triple, elf in code_objects(data):
        text = disassemble(elf)
        ks = parse_kernels(text)
        for name, insns in ks.items():
            if kernel_filter and not re.search(kernel_filter, name):
                continue
            n_kernels += 1; n_insns += len(insns)
            fs = lint_kernel(insns)
            if verbose:
                print(f"{len(insns):7d} instructions, {len(fs)} finding(s): {name[:140]}")
            findings += [(name,) + f for f in fs]
    return findings, n_kernels, n_insns


def describe(f):
    name, rule, p, c, got, need = f
    return (f"{rule}: {got} wait state(s) between producer and consumer, {need} needed\n    kernel   {name[:160]}\n"
            f"    producer {p.addr:#x}: {p.text}\n    consumer {c.addr:#x}: {c.text}")


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    kf = None
    if "--kernels" in sys.argv:
        kf = sys.argv[sys.argv.index("--kernels") + 1]
        args = [a for a in args if a != kf]
    fs, nk, ni = lint_library(args[0], kf, verbose="-v" in sys.argv)
    for f in fs:
        print(describe(f), file=sys.stderr)
    if fs:
        raise SystemExit(f"{args[0]}: {len(fs)} hazard finding(s) in {nk} kernels ({ni} instructions)")
    print(f"{args[0]}: {nk} kernels, {ni} instructions, no software-managed hazard left open ({len(RULES)} rules)")
