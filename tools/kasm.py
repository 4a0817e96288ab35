#!/usr/bin/env python3
"""kasm -- a small assembler toolkit for hand-scheduled gfx950 (CDNA4) code.

The walk kernel's inner loop is generated, not typed: `tools/gen_walk_asm.py` describes the arithmetic of one
kangaroo jump on VIRTUAL registers with the builder below; this module then
  1. list-schedules every basic block by critical path while honouring the gfx950 issue hazards that hipcc does not
     pad inside an asm statement (VALU-written SGPR -> VALU read: 2 wait states, -> VMEM read: 5; 128-bit store data
     overwritten by a VALU: 2; VGPR written by a VALU -> v_readfirstlane: 1),
  2. allocates physical registers (even-aligned pairs / quads where the ISA wants tuples, loop-carried values pinned),
  3. verifies the final text order again (hazards across block boundaries, s_waitcnt coverage of every load consumer)
     and pads with s_nop only where something is still violated,
  4. prints the text, either as a stand-alone listing or as the body of a HIP `asm volatile` statement.
`tools/kasm_emu.py` executes the printed text lane by lane, so the generator is tested on the CPU against big-integer
arithmetic before a GPU sees it.
"""
from __future__ import annotations

import collections
import re

# ---------------------------------------------------------------------------------------------------------------
# registers
# ---------------------------------------------------------------------------------------------------------------


class Reg:
    """one 32-bit register (virtual until `phys` is set).  kind: 'v' VGPR, 's' SGPR, 'x' special (vcc/exec/scc)"""

    __slots__ = ("kind", "name", "phys", "tup", "idx", "pinned", "operand", "uid")
    _n = 0

    def __init__(self, kind, name, phys=None, operand=None, pinned=False):
        self.kind, self.name, self.phys, self.operand, self.pinned = kind, name, phys, operand, pinned
        self.tup, self.idx = None, 0
        Reg._n += 1
        self.uid = Reg._n

    def __repr__(self):
        return f"<{self.kind}:{self.name}:{self.phys}>"


class Tup:
    """consecutive registers, first one aligned to `align` (gfx90a+: 64-bit and wider VGPR operands are even-aligned)"""

    def __init__(self, regs, align=2):
        self.regs, self.align = list(regs), align
        for i, r in enumerate(self.regs):
            assert r.tup is None, f"{r} already in a tuple"
            r.tup, r.idx = self, i

    def __getitem__(self, i):
        return self.regs[i]

    def __len__(self):
        return len(self.regs)

    @property
    def lo(self):
        return self.regs[0]

    @property
    def hi(self):
        return self.regs[1]

    def sub(self, i, n=2):
        """an aligned sub-tuple as an operand (e.g. half of a quad)"""
        return SubTup(self, i, n)

    def __repr__(self):
        return f"<tup {self.regs[0].name} x{len(self.regs)}>"


class SubTup:
    def __init__(self, tup, i, n):
        assert i % 2 == 0 or n == 1
        self.regs = tup.regs[i:i + n]

    def __getitem__(self, i):
        return self.regs[i]

    def __len__(self):
        return len(self.regs)


def regs_of(x):
    if isinstance(x, Reg):
        return [x]
    if isinstance(x, (Tup, SubTup)):
        return list(x.regs)
    return []


VCC_LO = Reg("x", "vcc_lo", phys="vcc_lo")
VCC_HI = Reg("x", "vcc_hi", phys="vcc_hi")
VCC = Tup([VCC_LO, VCC_HI])
EXEC_LO = Reg("x", "exec_lo", phys="exec_lo")
EXEC_HI = Reg("x", "exec_hi", phys="exec_hi")
EXEC = Tup([EXEC_LO, EXEC_HI])
SCC = Reg("x", "scc", phys="scc")
MEMTOK = Reg("x", "mem", phys="mem")  # orders memory instructions and waits among themselves


def fmt_operand(x):
    if isinstance(x, Reg):
        if x.operand is not None:
            return x.operand
        if x.kind == "x":
            return x.phys
        assert x.phys is not None, f"unallocated {x}"
        return f"{x.kind}{x.phys}"
    if isinstance(x, (Tup, SubTup)):
        r0 = x.regs[0]
        if r0.kind == "x":
            return {"vcc_lo": "vcc", "exec_lo": "exec"}[r0.phys]
        if r0.operand is not None:
            return r0.operand
        for i, r in enumerate(x.regs):
            assert r.phys == r0.phys + i, f"tuple not consecutive: {x.regs}"
        return f"{r0.kind}[{r0.phys}:{r0.phys + len(x.regs) - 1}]"
    if isinstance(x, int):
        return str(x) if -16 <= x <= 64 else hex(x)
    return str(x)


# ---------------------------------------------------------------------------------------------------------------
# instructions
# ---------------------------------------------------------------------------------------------------------------


class Ins:
    __slots__ = ("op", "args", "defs", "uses", "cls", "mods", "comment", "order", "label", "target", "vm", "barrier",
                 "accum", "asap")

    def __init__(self, op, args=(), defs=(), uses=(), cls="valu", mods="", comment="", label=None, target=None, vm=None,
                 barrier=False):
        self.op, self.args, self.cls, self.mods, self.comment = op, list(args), cls, mods, comment
        self.defs = [r for d in defs for r in regs_of(d)]
        self.uses = [r for u in uses for r in regs_of(u)]
        self.label, self.target = label, target
        self.vm = vm  # for waits: (vmcnt, lgkmcnt) or None components
        self.barrier = barrier  # scheduling barrier (keeps its place)
        self.order = 0
        self.accum = None  # register (tuple) this instruction ORs into: such instructions commute among themselves
        self.asap = False  # issue as soon as ready whatever the critical path says (it ends live ranges and starts none)

    def text(self):
        if self.cls == "label":
            return f"{self.label}:"
        s = self.op
        if self.args:
            s += " " + ", ".join(fmt_operand(a) for a in self.args)
        if self.mods:
            s += " " + self.mods
        return s

    def __repr__(self):
        return f"Ins({self.op} {self.args})"


class Block:
    def __init__(self, name, schedule=True):
        self.name, self.ins, self.schedule = name, [], schedule


class Asm:
    """builder: emits into the current block"""

    def __init__(self):
        self.blocks = []
        self.cur = None
        self.block("entry")
        self._lbl = 0

    # -- structure
    def block(self, name, schedule=True):
        self.cur = Block(name, schedule)
        self.blocks.append(self.cur)
        return self.cur

    def emit(self, ins):
        self.cur.ins.append(ins)
        return ins

    def label(self, name):
        """a label starts a new block"""
        self.block(name)
        self.emit(Ins("", cls="label", label=name, barrier=True))

    def newlabel(self, stem):
        self._lbl += 1
        return f".Lk_{stem}_{self._lbl}_%="

    # -- registers
    def v(self, name, pinned=False):
        return Reg("v", name, pinned=pinned)

    def vt(self, name, n, pinned=False):
        return Tup([Reg("v", f"{name}{i}", pinned=pinned) for i in range(n)])

    def s(self, name, pinned=False):
        return Reg("s", name, pinned=pinned)

    def st(self, name, n=2, pinned=False):
        return Tup([Reg("s", f"{name}{i}", pinned=pinned) for i in range(n)], align=min(n, 4))

    def operand(self, kind, text, n=1):
        """a register the compiler allocates: an inline-asm operand such as %3 (n > 1: prints as a range by itself)"""
        if n == 1:
            return Reg(kind, text, operand=text)
        return Tup([Reg(kind, f"{text}.{i}", operand=text) for i in range(n)])

    # -- VALU
    def v_mad_u64_u32(self, d, cout, a, b, c, comment=""):
        """d(64) = a*b + c(64); carry-out of the 64-bit sum -> cout (SGPR pair / vcc)"""
        return self.emit(Ins("v_mad_u64_u32", [d, cout, a, b, c], defs=[d, cout], uses=[a, b, c, EXEC], comment=comment))

    def _carry3(self, op, d, cout, x, y, cin=None, comment=""):
        args = [d, cout, x, y] + ([cin] if cin is not None else [])
        return self.emit(Ins(op, args, defs=[d, cout], uses=[x, y, EXEC] + ([cin] if cin is not None else []), comment=comment))

    def v_add_co_u32(self, d, cout, x, y, **k):
        return self._carry3("v_add_co_u32_e64", d, cout, x, y, **k)

    def v_addc_co_u32(self, d, cout, x, y, cin, **k):
        return self._carry3("v_addc_co_u32_e64", d, cout, x, y, cin, **k)

    def v_sub_co_u32(self, d, cout, x, y, **k):
        return self._carry3("v_sub_co_u32_e64", d, cout, x, y, **k)

    def v_subb_co_u32(self, d, cout, x, y, cin, **k):
        return self._carry3("v_subb_co_u32_e64", d, cout, x, y, cin, **k)

    def _v2(self, op, d, *srcs, comment=""):
        return self.emit(Ins(op, [d, *srcs], defs=[d], uses=[*srcs, EXEC], comment=comment))

    def v_mov_b32(self, d, x, **k):
        return self._v2("v_mov_b32_e32", d, x, **k)

    def v_and_b32(self, d, x, y, **k):
        return self._v2("v_and_b32_e32", d, x, y, **k)

    def v_or_b32(self, d, x, y, **k):
        return self._v2("v_or_b32_e32", d, x, y, **k)

    def v_add_u32(self, d, x, y, **k):
        return self._v2("v_add_u32_e32", d, x, y, **k)

    def v_sub_u32(self, d, x, y, **k):
        return self._v2("v_sub_u32_e32", d, x, y, **k)

    def v_lshlrev_b32(self, d, sh, x, **k):
        return self._v2("v_lshlrev_b32_e32", d, sh, x, **k)

    def v_lshrrev_b32(self, d, sh, x, **k):
        return self._v2("v_lshrrev_b32_e32", d, sh, x, **k)

    def v_alignbit_b32(self, d, hi, lo, sh, **k):
        return self._v2("v_alignbit_b32", d, hi, lo, sh, **k)

    def v_and_or_b32(self, d, x, y, z, **k):
        return self._v2("v_and_or_b32", d, x, y, z, **k)

    def v_lshl_add_u32(self, d, x, sh, y, **k):
        return self._v2("v_lshl_add_u32", d, x, sh, y, **k)

    def v_accum3(self, op, acc, x, y):
        """acc = op(acc, x, y) for a commutative, associative op (v_max3_u32 / v_min3_u32): accumulations into the same
        register commute, like s_or_accum -- the scheduler orders them only against ordinary accesses of `acc`"""
        ins = self.emit(Ins(op, [acc, acc, x, y], defs=[], uses=[x, y, EXEC]))
        ins.accum = acc
        return ins

    def v_cmp_le_u32(self, sd, x, y, **k):
        return self._v2("v_cmp_le_u32_e64", sd, x, y, **k)

    def v_cmp_gt_u32(self, sd, x, y, **k):
        return self._v2("v_cmp_gt_u32_e64", sd, x, y, **k)

    def v_cmp_eq_u32(self, sd, x, y, **k):
        return self._v2("v_cmp_eq_u32_e64", sd, x, y, **k)

    def v_cmp_lt_u32(self, sd, x, y, **k):
        return self._v2("v_cmp_lt_u32_e64", sd, x, y, **k)

    def v_cndmask_b32(self, d, x0, x1, m, **k):
        return self._v2("v_cndmask_b32_e64", d, x0, x1, m, **k)

    def v_mbcnt_lo(self, d, m, x, **k):
        return self._v2("v_mbcnt_lo_u32_b32", d, m, x, **k)

    def v_mbcnt_hi(self, d, m, x, **k):
        return self._v2("v_mbcnt_hi_u32_b32", d, m, x, **k)

    def v_readfirstlane_b32(self, sd, x, **k):
        return self._v2("v_readfirstlane_b32", sd, x, **k)

    # -- SALU
    def _s(self, op, d, *srcs, scc=False, reads_scc=False, comment=""):
        defs = [d] if d is not None else []
        if scc:
            defs.append(SCC)
        uses = list(srcs) + ([SCC] if reads_scc else [])
        return self.emit(Ins(op, ([d] if d is not None else []) + list(srcs), defs=defs, uses=uses, cls="salu", comment=comment))

    def s_mov_b32(self, d, x, **k):
        return self._s("s_mov_b32", d, x, **k)

    def s_mov_b64(self, d, x, **k):
        return self._s("s_mov_b64", d, x, **k)

    def s_or_b64(self, d, x, y, **k):
        return self._s("s_or_b64", d, x, y, **k)

    def s_or_accum(self, acc, m):
        """acc |= m.  Accumulations into the same register commute: the scheduler orders them only against ordinary
        readers / writers of `acc`.  (Clobbers SCC without saying so: s_cmp is a barrier, nothing slips behind it.)"""
        x = self.emit(Ins("s_or_b64", [acc, acc, m], defs=[], uses=[m], cls="salu"))
        x.accum = acc
        return x

    def s_and_b64(self, d, x, y, **k):
        return self._s("s_and_b64", d, x, y, **k)

    def s_andn2_b64(self, d, x, y, **k):
        return self._s("s_andn2_b64", d, x, y, **k)

    def s_add_u32(self, d, x, y, **k):
        return self._s("s_add_u32", d, x, y, scc=True, **k)

    def s_addc_u32(self, d, x, y, **k):
        return self._s("s_addc_u32", d, x, y, scc=True, reads_scc=True, **k)

    def s_sub_u32(self, d, x, y, **k):
        return self._s("s_sub_u32", d, x, y, scc=True, **k)

    def s_add_i32(self, d, x, y, **k):
        return self._s("s_add_i32", d, x, y, scc=True, **k)

    def s_mul_i32(self, d, x, y, **k):
        return self._s("s_mul_i32", d, x, y, **k)

    def s_lshl_b32(self, d, x, y, **k):
        return self._s("s_lshl_b32", d, x, y, **k)

    def s_lshr_b32(self, d, x, y, **k):
        return self._s("s_lshr_b32", d, x, y, **k)

    def s_lshl_b64(self, d, x, y, **k):
        return self._s("s_lshl_b64", d, x, y, **k)

    def s_and_b32(self, d, x, y, **k):
        return self._s("s_and_b32", d, x, y, **k)

    def s_bcnt1_i32_b64(self, d, x, **k):
        return self._s("s_bcnt1_i32_b64", d, x, **k)

    def s_ff1_i32_b64(self, d, x, **k):
        return self._s("s_ff1_i32_b64", d, x, **k)

    def s_cselect_b32(self, d, x, y, **k):
        return self._s("s_cselect_b32", d, x, y, reads_scc=True, **k)

    def s_cmp(self, cc, x, y, **k):
        """cc like 'lg_u64', 'eq_u32', 'lt_u32', 'ge_u32'.  A barrier: logical SALU ops clobber SCC untracked."""
        x = self._s(f"s_cmp_{cc}", None, x, y, scc=True, **k)
        x.barrier = True
        return x

    def s_and_saveexec_b64(self, d, x, **k):
        return self.emit(Ins("s_and_saveexec_b64", [d, x], defs=[d, EXEC, SCC], uses=[x, EXEC], cls="salu", barrier=True))

    def s_mov_exec(self, x):
        return self.emit(Ins("s_mov_b64", [EXEC, x], defs=[EXEC], uses=[x], cls="salu", barrier=True))

    def s_nop(self, n=0):
        return self.emit(Ins("s_nop", [n], cls="nop", barrier=True))

    # -- control flow (each ends its block)
    def branch(self, op, target, uses=()):
        self.emit(Ins(op, [target], uses=list(uses), cls="branch", target=target, barrier=True))
        self.block(f"after_{target}")

    def s_cbranch_scc1(self, t):
        self.branch("s_cbranch_scc1", t, [SCC])

    def s_cbranch_scc0(self, t):
        self.branch("s_cbranch_scc0", t, [SCC])

    def s_cbranch_execz(self, t):
        self.branch("s_cbranch_execz", t, [EXEC])

    def s_branch(self, t):
        self.branch("s_branch", t)

    # -- memory.  MEMTOK keeps memory instructions and waits in program order among themselves.
    def _mem(self, op, args, defs, uses, cls, mods):
        return self.emit(Ins(op, args, defs=list(defs) + [MEMTOK], uses=list(uses) + [MEMTOK, EXEC], cls=cls, mods=mods))

    def global_load(self, width, d, voff, sbase, offset=0, nt=False, coherent=False):
        """d <- [sbase + zext(voff) + offset]; width in dwords (1, 2, 4).  coherent: sc0 sc1 (served by L2, never by a line the
        vector L1 still holds: what a load must carry to see an L2 atomic of the same kernel)"""
        op = {1: "global_load_dword", 2: "global_load_dwordx2", 4: "global_load_dwordx4"}[width]
        mods = (f"offset:{offset}" if offset else "") + (" nt" if nt else "") + (" sc0 sc1" if coherent else "")
        return self._mem(op, [d, voff, sbase], [d], [voff, sbase], "vmem_ld", mods.strip())

    def global_store(self, width, voff, data, sbase, offset=0, nt=False):
        op = {1: "global_store_dword", 2: "global_store_dwordx2", 4: "global_store_dwordx4"}[width]
        mods = (f"offset:{offset}" if offset else "") + (" nt" if nt else "")
        return self._mem(op, [voff, data, sbase], [], [voff, data, sbase], "vmem_st", mods.strip())

    def global_atomic_add_rtn(self, d, voff, data, sbase):
        return self._mem("global_atomic_add", [d, voff, data, sbase], [d], [voff, data, sbase], "vmem_ld", "sc0")

    def global_atomic_add_x2(self, voff, data, sbase):
        """[sbase + zext(voff)] += data (64 bit), no return value: counted in vmcnt like a store"""
        return self._mem("global_atomic_add_x2", [voff, data, sbase], [], [voff, data, sbase], "vmem_st", "")

    def ds_read_b64(self, d, addr, offset=0):
        return self._mem("ds_read_b64", [d, addr], [d], [addr], "lds", f"offset:{offset}" if offset else "")

    def ds_read2_b64(self, d, addr, o0, o1):
        mods = " ".join(m for m in (f"offset0:{o0}" if o0 else "", f"offset1:{o1}" if o1 else "") if m)
        return self._mem("ds_read2_b64", [d, addr], [d], [addr], "lds", mods)

    def s_load(self, width, d, sbase, offset=0):
        op = {1: "s_load_dword", 2: "s_load_dwordx2", 4: "s_load_dwordx4", 8: "s_load_dwordx8", 16: "s_load_dwordx16"}[width]
        return self._mem(op, [d, sbase, hex(offset)], [d], [sbase], "smem", "")

    def s_waitcnt(self, vmcnt=None, lgkmcnt=None, regs=()):
        """`regs`: the registers this wait makes valid -- consumers of a load depend on the wait, not on the load"""
        parts = []
        if vmcnt is not None:
            parts.append(f"vmcnt({vmcnt})")
        if lgkmcnt is not None:
            parts.append(f"lgkmcnt({lgkmcnt})")
        rr = [r for x in regs for r in regs_of(x)]
        return self.emit(Ins("s_waitcnt", [" ".join(parts)], defs=rr + [MEMTOK], uses=rr + [MEMTOK], cls="wait", vm=(vmcnt, lgkmcnt)))

    def raw(self, text, defs=(), uses=(), cls="other", barrier=True):
        return self.emit(Ins(text, [], defs=defs, uses=uses, cls=cls, barrier=barrier))

    def keep(self, *regs):
        """marks registers as live up to this point (e.g. outputs of the whole program)"""
        return self.emit(Ins("", [], uses=regs, cls="keep", barrier=False))


# ---------------------------------------------------------------------------------------------------------------
# hazards (gfx940/gfx950; LLVM's GCNHazardRecognizer is the source of the distances)
# ---------------------------------------------------------------------------------------------------------------
# required number of OTHER issue slots between producer and consumer ("wait states")
WS_VALU_SGPR_TO_VALU = 2
WS_VALU_SGPR_TO_VMEM = 5
WS_STORE_DATA_TO_VALU_WRITE = 2  # 128-bit (and 96-bit) store data
WS_VALU_VGPR_TO_READLANE = 1
# not a hazard but a stall: an SALU instruction that reads an SGPR a VALU has just written waits for the VALU result
SOFT_VALU_SGPR_TO_SALU = 6
# issue slots the scheduler tries to put between a ds_read and the s_waitcnt that makes its data valid (a latency to hide,
# not a hazard: 1 = wait right behind the read)
import os as _os
LDS_WAIT_SLOTS = int(_os.environ.get("KASM_LDSLAT", "128"))  # +0.7 % against 1 (profiles/r03_ab_flags_valu.txt)


def _is_sgprish(r):
    return r.kind == "s" or r in (VCC_LO, VCC_HI)


class HazardState:
    """distance (in issue slots) since the events that matter"""

    def __init__(self):
        self.valu_sgpr = {}  # reg -> slot of the last VALU write
        self.valu_vgpr = {}  # reg -> slot of the last VALU write (for readfirstlane)
        self.store_data = {}  # reg -> slot of the 128-bit store that reads it

    def copy(self):
        h = HazardState()
        h.valu_sgpr, h.valu_vgpr, h.store_data = dict(self.valu_sgpr), dict(self.valu_vgpr), dict(self.store_data)
        return h

    def need(self, ins, pos, soft=False):
        """how many more slots `ins` has to wait before it may issue at slot `pos` (0 = fine)"""
        need = 0
        if ins.cls in ("valu", "vmem_ld", "vmem_st", "lds"):
            ws = WS_VALU_SGPR_TO_VALU if ins.cls == "valu" else WS_VALU_SGPR_TO_VMEM
            for r in ins.uses:
                if _is_sgprish(r) and r in self.valu_sgpr:
                    need = max(need, self.valu_sgpr[r] + ws + 1 - pos)
        if soft and ins.cls in ("salu", "branch"):
            for r in ins.uses:
                if _is_sgprish(r) and r in self.valu_sgpr:
                    need = max(need, self.valu_sgpr[r] + SOFT_VALU_SGPR_TO_SALU + 1 - pos)
        if ins.cls == "valu":
            for r in ins.defs:
                if r in self.store_data:
                    need = max(need, self.store_data[r] + WS_STORE_DATA_TO_VALU_WRITE + 1 - pos)
            if ins.op.startswith("v_readfirstlane") or ins.op.startswith("v_readlane"):
                for r in ins.uses:
                    if r in self.valu_vgpr:
                        need = max(need, self.valu_vgpr[r] + WS_VALU_VGPR_TO_READLANE + 1 - pos)
        return max(need, 0)

    def issue(self, ins, pos):
        if ins.cls == "valu":
            for r in ins.defs:
                if _is_sgprish(r):
                    self.valu_sgpr[r] = pos
                elif r.kind == "v":
                    self.valu_vgpr[r] = pos
        else:
            for r in ins.defs:  # an SALU / SMEM write replaces the VALU-written value
                self.valu_sgpr.pop(r, None)
        if ins.cls == "vmem_st" and ("x4" in ins.op or "x3" in ins.op):
            for r in regs_of(ins.args[1]):
                self.store_data[r] = pos

    def merge(self, other, shift):
        """state at a join: `other` was recorded `shift` slots ago at the branch"""
        for name in ("valu_sgpr", "valu_vgpr", "store_data"):
            mine, theirs = getattr(self, name), getattr(other, name)
            for r, p in theirs.items():
                mine[r] = max(mine.get(r, -10**9), p + shift)


def slots_of(ins):
    """issue slots an instruction occupies for hazard purposes"""
    if ins.cls in ("label", "keep"):
        return 0
    if ins.cls == "nop":
        return int(ins.args[0]) + 1
    return 1


# ---------------------------------------------------------------------------------------------------------------
# scheduling
# ---------------------------------------------------------------------------------------------------------------


def schedule_block(block, hz, window=None, sgpr_limit=28, vgpr_limit=150):
    """list scheduling by critical path; returns the new instruction list.  `hz`: HazardState at block entry
    (positions relative to slot 0 = first slot of this block); it is advanced to the block's end."""
    ins = block.ins
    n = len(ins)
    for i, x in enumerate(ins):
        x.order = i
    if not block.schedule or n <= 2:
        out, pos = [], 0
        for x in ins:
            w = hz.need(x, pos)
            if w:
                out.append(Ins("s_nop", [w - 1], cls="nop"))
                pos += w
            hz.issue(x, pos)
            out.append(x)
            pos += slots_of(x)
        return out, pos
    succ = [[] for _ in range(n)]
    npred = [0] * n
    lat = {}

    def edge(a, b, w=1):
        if a == b:
            return
        key = (a, b)
        if key in lat:
            lat[key] = max(lat[key], w)
            return
        lat[key] = w
        succ[a].append(b)
        npred[b] += 1

    last_def, readers = {}, collections.defaultdict(list)
    last_barrier = None
    since_barrier = []
    accums = collections.defaultdict(list)  # register -> accumulating instructions since its last ordinary access
    for i, x in enumerate(ins):
        if x.accum is not None:
            for r in regs_of(x.accum):
                if r in last_def:
                    edge(last_def[r], i, 1)
                for q in readers[r]:
                    edge(q, i, 0)
                accums[r].append(i)
        else:
            for r in x.uses + x.defs:
                if accums.get(r):
                    for q in accums[r]:
                        edge(q, i, 1)
            for r in x.defs:
                accums.pop(r, None)
        for r in x.uses:
            if r in last_def:
                a = last_def[r]
                w = 1
                if ins[a].cls == "valu" and _is_sgprish(r):
                    w = {"valu": 3, "vmem_ld": 6, "vmem_st": 6, "lds": 6}.get(x.cls, 1)
                if ins[a].cls == "lds" and x.cls == "wait" and r is not MEMTOK:
                    w = LDS_WAIT_SLOTS  # the data is ~LDS_WAIT_SLOTS issue slots away: fill them instead of parking the wave
                edge(a, i, w)
        for r in x.defs:
            if r in last_def:
                edge(last_def[r], i, 1)
            for q in readers[r]:
                edge(q, i, 0)
        if last_barrier is not None:
            edge(last_barrier, i, 1)
        if x.barrier:
            for q in since_barrier:
                edge(q, i, 0)
            last_barrier, since_barrier = i, []
        else:
            since_barrier.append(i)
        for r in x.uses:
            readers[r].append(i)
        for r in x.defs:
            last_def[r] = i
            readers[r] = []
    # critical path to the end of the block
    prio = [0] * n
    for i in range(n - 1, -1, -1):
        p = 0
        for j in succ[i]:
            p = max(p, prio[j] + max(lat[(i, j)], 1 if slots_of(ins[j]) else 0))
        prio[i] = p
    ready = [i for i in range(n) if npred[i] == 0]
    earliest = [0] * n
    # register pressure: virtual SGPRs (carry masks) that are defined and still have readers to come.  Critical-path
    # order alone issues every ready multiply first and lets their carry consumers trail: beyond `sgpr_limit` live
    # scalars only instructions that create none may issue.
    def tracked(r):
        return r.kind == "s" and not r.pinned and r.operand is None and not (r.tup is not None and any(q.pinned for q in r.tup.regs))

    def tracked_v(r):
        return r.kind == "v" and not r.pinned and r.operand is None and not (r.tup is not None and any(q.pinned for q in r.tup.regs))

    uses_left, defs_left = collections.Counter(), collections.Counter()
    for x in ins:
        for r in set(x.uses):
            if tracked(r) or tracked_v(r):
                uses_left[r] += 1
        for r in set(x.defs):
            if tracked(r) or tracked_v(r):
                defs_left[r] += 1
    live = set()
    live_v = set()
    out, pos, done = [], 0, 0
    while done < n:
        best, best_key = None, None
        lo_order = min(ready)
        over = len(live) >= sgpr_limit
        over_v = len(live_v) >= vgpr_limit
        for relax in (False, True):
            for i in ready:
                if window is not None and i - lo_order > window:
                    continue
                if earliest[i] > pos:
                    continue
                x = ins[i]
                if not relax:
                    if over and any(tracked(r) and r not in live for r in x.defs):
                        continue
                    if over_v and any(tracked_v(r) and r not in live_v for r in x.defs) and not any(
                            tracked_v(r) and uses_left[r] == 1 for r in x.uses):
                        continue
                if hz.need(x, pos) > 0:
                    continue
                soft = hz.need(x, pos, soft=True) > 0
                key = (soft, -(prio[i] + (10**6 if x.asap else 0)), i)
                if best is None or key < best_key:
                    best, best_key = i, key
            if best is not None or not (over or over_v):
                break
        if best is None:
            # nothing may issue: advance one slot (a hazard pad)
            out.append(Ins("s_nop", [0], cls="nop"))
            pos += 1
            continue
        x = ins[best]
        ready.remove(best)
        hz.issue(x, pos)
        out.append(x)
        pos += slots_of(x)
        done += 1
        for r in set(x.defs):
            if r in defs_left:
                defs_left[r] -= 1
                if uses_left[r] > 0:
                    (live if r.kind == "s" else live_v).add(r)
        for r in set(x.uses):
            if r in uses_left:
                uses_left[r] -= 1
                if uses_left[r] == 0 and defs_left[r] == 0:
                    live.discard(r)
                    live_v.discard(r)
        for j in succ[best]:
            npred[j] -= 1
            earliest[j] = max(earliest[j], (pos - slots_of(x)) + lat[(best, j)])
            if npred[j] == 0:
                ready.append(j)
    # merge consecutive nops
    merged = []
    for x in out:
        if x.cls == "nop" and merged and merged[-1].cls == "nop" and not merged[-1].barrier and not x.barrier and int(merged[-1].args[0]) < 7:
            merged[-1].args[0] = int(merged[-1].args[0]) + int(x.args[0]) + 1
        else:
            merged.append(x)
    return merged, pos


def schedule(asm, window=None, sgpr_limit=28, vgpr_limit=150):
    """schedules every block in text order, carrying the hazard state along fall-through paths; branch targets get the
    merged state of their predecessors (backward branches are padded until they carry no pending hazard)"""
    at_branch = collections.defaultdict(list)  # label -> [(HazardState, slot)]
    hz, base = HazardState(), 0
    seen_labels = set()
    for b in asm.blocks:
        if b.ins and b.ins[0].cls == "label":
            lbl = b.ins[0].label
            seen_labels.add(lbl)
            for st, slot in at_branch.get(lbl, []):
                hz.merge(st, base - slot)  # the taken path skips the slots in between: same distance from the join as from the branch
        # positions inside schedule_block are relative to the block: rebase the state
        rel = HazardState()
        for name in ("valu_sgpr", "valu_vgpr", "store_data"):
            setattr(rel, name, {r: p - base for r, p in getattr(hz, name).items() if base - p < 16})
        new, length = schedule_block(b, rel, window, sgpr_limit, vgpr_limit)
        # a backward branch must not carry pending hazards into the loop header
        if new and new[-1].cls == "branch" and new[-1].target in seen_labels:
            pend = 0
            for d, ws in ((rel.valu_sgpr, WS_VALU_SGPR_TO_VMEM), (rel.store_data, WS_STORE_DATA_TO_VALU_WRITE), (rel.valu_vgpr, 1)):
                for r, p in d.items():
                    pend = max(pend, p + ws + 1 - (length - 1))
            if pend > 0:
                new.insert(len(new) - 1, Ins("s_nop", [pend - 1], cls="nop"))
                length += pend
                for d in (rel.valu_sgpr, rel.store_data, rel.valu_vgpr):
                    d.clear()
        b.ins = new
        hz = HazardState()
        for name in ("valu_sgpr", "valu_vgpr", "store_data"):
            setattr(hz, name, {r: p + base for r, p in getattr(rel, name).items()})
        base += length
        if new and new[-1].cls == "branch" and new[-1].target not in seen_labels:
            at_branch[new[-1].target].append((hz.copy(), base))
        if new and new[-1].op == "s_branch":
            hz = HazardState()  # no fall-through


# ---------------------------------------------------------------------------------------------------------------
# register allocation
# ---------------------------------------------------------------------------------------------------------------


def linear(asm):
    return [x for b in asm.blocks for x in b.ins]


def allocate(asm, vpool, spool):
    """vpool / spool: lists of physical register numbers this program may use.  Pinned registers live for the whole
    program; everything else gets its first-definition .. last-use interval (extended over loops it is live into)."""
    prog = linear(asm)
    first, last = {}, {}
    for p, x in enumerate(prog):
        for r in x.uses:
            if r.kind in ("v", "s") and r.operand is None and r.phys is None and r not in first and not r.pinned and not (
                    r.tup is not None and any(q.pinned for q in r.tup.regs)):
                raise RuntimeError(f"{r} is read by '{x.op}' before anything defines it: loop-carried values must be pinned")
        for r in x.defs + x.uses:
            if r.kind in ("v", "s") and r.operand is None and r.phys is None:
                first.setdefault(r, p)
                last[r] = p
    # loops: label position .. backward branch position
    label_pos = {x.label: p for p, x in enumerate(prog) if x.cls == "label"}
    loops = [(label_pos[x.target], p) for p, x in enumerate(prog) if x.cls == "branch" and x.target in label_pos and label_pos[x.target] < p]
    changed = True
    while changed:
        changed = False
        for r in first:
            for (h, e) in loops:
                # live into the loop header from outside, or defined in the loop and used before its definition (carried)
                if first[r] < h <= last[r] and last[r] < e:
                    last[r] = e
                    changed = True
    groups = {}
    for r in first:
        g = r.tup if r.tup is not None else r
        if isinstance(g, Tup) and any(q.kind == "x" for q in g.regs):
            continue
        s, e = groups.get(g, (10**9, -1))
        pinned = r.pinned
        s, e = min(s, first[r]), max(e, last[r])
        if pinned:
            s, e = -1, len(prog)
        groups[g] = (s, e)
    # a tuple with one pinned member is pinned as a whole
    for g in list(groups):
        if isinstance(g, Tup) and any(q.pinned for q in g.regs):
            groups[g] = (-1, len(prog))
    used = {"v": set(), "s": set()}
    for kind, pool in (("v", vpool), ("s", spool)):
        # (ties in creation order -- never by id(): the generated text must not depend on where objects landed in memory)
        items = sorted((se, (g.regs[0].uid if isinstance(g, Tup) else g.uid), g) for g, se in groups.items()
                       if (g.regs[0].kind if isinstance(g, Tup) else g.kind) == kind)
        free = set(pool)
        active = []  # (end, regs)
        for (s, e), _, g in items:
            for a in list(active):
                if a[0] < s:
                    free.update(a[1])
                    active.remove(a)
            n = len(g.regs) if isinstance(g, Tup) else 1
            align = g.align if isinstance(g, Tup) else 1
            got = None
            for base in sorted(free):
                if base % align:
                    continue
                if all((base + i) in free for i in range(n)):
                    got = base
                    break
            if got is None:
                raise RuntimeError(f"out of {kind} registers at {g} (interval {s}..{e}); pool {len(pool)}")
            regs = [got + i for i in range(n)]
            free.difference_update(regs)
            active.append((e, regs))
            used[kind].update(regs)
            if isinstance(g, Tup):
                for q, ph in zip(g.regs, regs):
                    q.phys = ph
            else:
                g.phys = got
    return used


# ---------------------------------------------------------------------------------------------------------------
# verification of the final text order
# ---------------------------------------------------------------------------------------------------------------


def verify(asm):
    """re-checks hazards on the final linear order (forward branches merge, backward branches must be clean) and that
    every consumer of a loaded register sits behind an s_waitcnt that covers the load on every path that reaches it.
    Returns a list of problems."""
    prog = linear(asm)
    problems = []
    hz = HazardState()
    pos = 0
    at_branch = collections.defaultdict(list)
    labels = set()
    # memory: per pending register, its counter class and how many operations of that class were issued behind it
    pending = {}  # reg -> [kind, age]; None = this point is unreachable in linear order (behind an s_branch)
    pend_at = collections.defaultdict(list)

    def merge_pending(a, b):
        if a is None:
            return None if b is None else {r: list(v) for r, v in b.items()}
        if b is None:
            return a
        out = {r: list(v) for r, v in a.items()}
        for r, (k, age) in b.items():
            if r in out:
                out[r][1] = min(out[r][1], age)
            else:
                out[r] = [k, age]
        return out

    for x in prog:
        if x.cls == "label":
            labels.add(x.label)
            for st, slot in at_branch.get(x.label, []):
                hz.merge(st, pos - slot)
            for st in pend_at.get(x.label, []):
                pending = merge_pending(pending, st)
            if pending is None:
                pending = {}
            continue
        if x.cls == "keep":
            continue
        if pending is None:
            pending = {}  # code behind an unconditional branch without a label: treat as reachable, nothing pending
        w = hz.need(x, pos)
        if w:
            problems.append(f"hazard: '{x.text()}' needs {w} more wait state(s) at slot {pos}")
        if x.cls == "wait":
            vmc, lgc = x.vm
            for r, (k, age) in list(pending.items()):
                if (k == "vm" and vmc is not None and age >= vmc) or (k == "lg" and lgc is not None and age >= lgc):
                    del pending[r]
        else:
            for r in x.uses + x.defs:
                if r in pending:
                    problems.append(f"waitcnt: '{x.text()}' touches {fmt_operand(r)} with its load still outstanding")
        if x.cls in ("vmem_ld", "vmem_st"):
            for v in pending.values():
                if v[0] == "vm":
                    v[1] += 1
            if x.cls == "vmem_ld":
                for r in x.defs:
                    if r is not MEMTOK:
                        pending[r] = ["vm", 0]
        elif x.cls in ("lds", "smem"):
            for v in pending.values():
                if v[0] == "lg":
                    v[1] += 1
            for r in x.defs:
                if r is not MEMTOK:
                    pending[r] = ["lg", 0]
        hz.issue(x, pos)
        pos += slots_of(x)
        if x.cls == "branch":
            if x.target in labels:  # backward
                for d, ws in ((hz.valu_sgpr, WS_VALU_SGPR_TO_VMEM), (hz.store_data, WS_STORE_DATA_TO_VALU_WRITE)):
                    for r, p in d.items():
                        if p + ws + 1 > pos:
                            problems.append(f"backward branch '{x.text()}' carries a pending hazard on {fmt_operand(r)}")
                if pending:
                    problems.append(f"backward branch '{x.text()}' with loads outstanding into {[fmt_operand(r) for r in pending]}")
            else:
                at_branch[x.target].append((hz.copy(), pos))
                pend_at[x.target].append({r: list(v) for r, v in pending.items()})
            if x.op == "s_branch":
                pending = None
                hz = HazardState()
    return problems


# ---------------------------------------------------------------------------------------------------------------
# output
# ---------------------------------------------------------------------------------------------------------------


def listing(asm, comments=True):
    out = []
    for x in linear(asm):
        if x.cls == "keep":
            continue
        t = x.text()
        if x.cls != "label":
            t = "\t" + t
        if comments and x.comment:
            t += f"\t; {x.comment}"
        out.append(t)
    return out


def stats(asm, blocks=None):
    h = collections.Counter()
    for b in asm.blocks:
        if blocks is not None and b.name not in blocks:
            continue
        for x in b.ins:
            if x.cls in ("label", "keep"):
                continue
            h[x.cls] += 1
            if x.cls == "nop":
                h["nop_states"] += int(x.args[0]) + 1
            if x.cls == "valu":
                op = re.sub(r"_e(32|64)$", "", x.op)
                h["valu_fast" if op in FAST_OPS else "valu_slow"] += 1
                if op == "v_mov_b32":
                    h["v_mov"] += 1
                if op == "v_mad_u64_u32":
                    h["mad"] += 1
    return dict(h)


FAST_OPS = {"v_mov_b32", "v_add_u32", "v_sub_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_lshlrev_b32"}
