#!/usr/bin/env python3
"""Generate kangaroo_amd/csrc/kng_walk_asm.h: the walk kernel's per-kangaroo loop (the k-loop of walk_body,
kng_engine.hip) as ONE scheduled gfx950 asm statement per distance layout.

What the loop does per iteration is exactly walk_body's iteration (GPUCompute.h:67-105 of the reference): P += J[x & 31]
with the running batch inverse, d += jD, DP test and wave-compacted DP records, prefix product of the next jump's dx.
What changes is who schedules it: tools/kasm.py list-schedules the whole iteration (six products, seven subtractions,
LDS reads, prefetch loads) as one dependency graph, so carry chains are padded by independent multiplies instead of
s_nop, operands stay in 32-bit registers from load to store (no repacking moves), and the prefetch registers rotate with
one move per word instead of the compiler's copies.

Exact-path protocol.  The short forms (single-chain fold, two-limb borrow fix-up, low-word distance add) flag the lanes
for which they are not exact in an SGPR mask.  When any lane of the wave is flagged, the iteration is abandoned BEFORE
anything is stored: the statement returns with k = that iteration and `inv` untouched (`acc` is undefined: the caller
re-reads it from the product plane, S[slot(k-1)], or takes 1 for k = 0), and the C++ caller runs this one iteration with
the generic code (walk_core in kng_engine.hip), then re-enters at k + 1.

Interface (see KNG_WALK_ASM_LOOP at the end of the generated header):
    in/out  inv[8], acc[8]   32-bit limbs of the running inverse / running prefix product
    in/out  k                next iteration to execute (SGPR); == G on normal completion
    in/out  voff             byte offset of kangaroo slot(k) in a 16-byte plane ((slot * L + t) * 16)
    in      args             device pointer to a WalkAsmArgs block (plane bases, DP buffer, mask)
    in      stride           +-L * 16: byte distance from slot(k) to slot(k+1) in a 16-byte plane
    in      G                kangaroos of this lane (wave-uniform)
    in      lds_tab          LDS byte address of the limb-major jump table
Planes are addressed as SGPR base + 32-bit VGPR byte offset, so a herd is limited to 2^28 kangaroos on this path
(4 GiB per 16-byte plane); the engine keeps the C++ loop for anything larger.
"""
from __future__ import annotations

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kasm  # noqa: E402
import kfield  # noqa: E402
from kasm import EXEC, Asm, Tup  # noqa: E402

# WalkAsmArgs layout (kng_engine.hip)
OFF_PLANES = 0x00  # x01 x23 y01 y23 dlo dhi s01 s23 : 8 pointers
OFF_DP = 0x40  # dp_mask(8) dp_count(8) dp_items(8) max_found(4) pad(4)

NOPS = {"x": 0}


class Loop:
    def __init__(self, dsplit: bool, abl=None):
        self.dsplit = dsplit
        self._abl_arg = abl
        A = self.A = Asm()
        # ---- operands (compiler-allocated)
        self.INV = [A.operand("v", f"%{i}") for i in range(8)]
        self.ACCop = [A.operand("v", f"%{8 + i}") for i in range(8)]
        self.k = A.operand("s", "%16")
        self.voff = A.operand("v", "%17")
        self.args = A.operand("s", "%18", 2)
        self.stride = A.operand("s", "%19")
        self.G = A.operand("s", "%20")
        self.ldstab = A.operand("s", "%21")
        # ---- pinned state
        self.planes = A.st("planes", 16, pinned=True)  # 8 pointers
        self.dpblk = A.st("dpblk", 8, pinned=True)  # mask, count ptr, items ptr, max_found, pad
        self.P = {n: self.planes.sub(2 * i, 2) for i, n in enumerate(["x01", "x23", "y01", "y23", "dlo", "dhi", "s01", "s23"])}
        self.dp_mask_lo, self.dp_mask_hi = self.dpblk[0], self.dpblk[1]
        self.dp_count = self.dpblk.sub(2, 2)
        self.dp_items = self.dpblk.sub(4, 2)
        self.max_found = self.dpblk[6]
        self.k977 = A.s("k977", pinned=True)
        self.v977 = A.v("v977", pinned=True)
        self.rare = A.st("rare", 2, pinned=True)
        q = lambda n: A.vt(n, 4, pinned=True)  # noqa: E731
        self.CX, self.CY = [q("cx0"), q("cx1")], [q("cy0"), q("cy1")]
        self.NX, self.NY = [q("nx0"), q("nx1")], [q("ny0"), q("ny1")]
        self.NB = [q("nb0"), q("nb1")]  # neighbour product; reloaded in place for the next kangaroo once P1 has read it
        # running prefix product: lives in two quads inside the statement (it is stored every iteration) and is updated
        # IN PLACE by P6 -- after an exact-path exit its old value is re-read from the product plane (S[slot(k-1)], or 1)
        self.ACCq = [q("acc0"), q("acc1")]
        self.CD, self.ND = q("cd"), q("nd")  # distance: (lo0, lo1, hi0, hi1); DSPLIT streams only the low pair
        self.F = kfield.Field(A, self.k977, self.rare)
        self.F.elide_first_carry = os.environ.get("KASM_ELIDE", "1") == "1"  # +2.5 % under the VGPR flags (profiles/r03_ab_elide.txt); -0.7 % when it cost a scalar OR (round 2)
        self.F.flag_mode = os.environ.get("KASM_FLAGS", "valu")  # +1.7 % against "salu" (profiles/r03_ab_flags_valu.txt)
        self.s_near_hi, self.s_near_lo = A.s("near_hi", pinned=True), A.s("near_lo", pinned=True)
        self.unroll = int(os.environ.get("KASM_UNROLL", "2"))
        # measurement builds only (WRONG results on purpose, see tools/r3_sensitivity.sh): comma list of
        #   nos      no traffic of the product planes S (the neighbour product is whatever the registers hold)
        #   nosA     ... in iteration A only           } together: the cost/benefit proxy of storing every OTHER
        #   plusmulA one extra product in iteration A  } prefix product (-32 B/jump, +1/2 multiplication per jump)
        #   nostore  x, y, d not written back
        #   noglobal no global loads / stores inside the loop at all;  nolds: jump-table words taken from other registers
        #   noflags  the exactness flags are not collected (no s_or of lane masks)
        self.abl = set(self._abl_arg) if self._abl_arg is not None else set(filter(None, os.environ.get("KASM_ABL", "").split(",")))
        self.in_loop = False
        # cache-hint experiments (default: x, y stream with nt; products and distances plain): letters of KASM_NT flip one each
        #   S product loads nt   s product stores nt   d distance loads + stores nt   X x/y loads plain   x x/y stores plain
        self.nt = os.environ.get("KASM_NT", "")
        # low-word distance streaming: a carry out of the low word either leaves the loop for the exact path ("exit": rounds 1-3a,
        # only sensible when carries are ~2^-23 per jump) or is added to the high word by an L2 atomic without leaving ("atomic":
        # one cold block per carrying wave-iteration, which makes the layout pay for jump distances up to 2^58)
        self.carry_mode = os.environ.get("KASM_CARRY", "atomic")
        if "noflags" in self.abl:
            A.s_or_accum = lambda acc, m: None

    @staticmethod
    def limbs(quads):
        return [r for qd in quads for r in qd.regs]

    # ------------------------------------------------------------------------------------------------
    def load_fe(self, quads, voff, p0, p1, nt, asap=False):
        if "noglobal" in self.abl and self.in_loop:
            return
        self.A.global_load(4, quads[0], voff, self.P[p0], nt=nt).asap = asap
        self.A.global_load(4, quads[1], voff, self.P[p1], nt=nt).asap = asap

    def load_d(self, quad, voff8, asap=False):
        A = self.A
        if "noglobal" in self.abl and self.in_loop:
            return
        A.global_load(2, quad.sub(0, 2), voff8, self.P["dlo"], nt="d" in self.nt).asap = asap
        if not self.dsplit:
            A.global_load(2, quad.sub(2, 2), voff8, self.P["dhi"], nt="d" in self.nt).asap = asap

    def set_one(self, quads):
        A = self.A
        for i, r in enumerate(self.limbs(quads)):
            A.v_mov_b32(r, 1 if i == 0 else 0)

    def build(self):
        """entry; loop { iteration A (state set 0 -> set 1); iteration B (set 1 -> set 0) }; exits.  With UNROLL = 1 only
        iteration A exists and its tail moves set 1 back to set 0 (27 moves per kangaroo-jump)."""
        A = self.A
        unroll = self.unroll
        L_exit, L_nb1, L_nbd = ".Lkw_exit_%=", ".Lkw_nb1_%=", ".Lkw_nbd_%="
        # ---- the two state sets: x, y, distance, running inverse, plane offset
        q = lambda n: A.vt(n, 4, pinned=True)  # noqa: E731
        set0 = dict(X=self.CX, Y=self.CY, D=self.CD, INV=self.INV, voff=self.voff)
        set1 = dict(X=self.NX, Y=self.NY, D=self.ND, voff=A.v("voff1", pinned=True),
                    INV=[r for qd in (q("inv1a"), q("inv1b")) for r in qd.regs] if unroll == 2 else None)
        # ================= entry (unscheduled) =================
        A.cur.schedule = False
        A.s_nop(4)  # operands may come fresh from a VALU (readfirstlane) : VALU-written SGPR -> SMEM/VMEM
        A.s_load(16, self.planes, self.args, OFF_PLANES)
        A.s_load(8, self.dpblk, self.args, OFF_DP)
        A.s_mov_b32(self.k977, 977)
        A.v_mov_b32(self.v977, 977)
        A.s_mov_b64(self.rare, 0)
        A.s_mov_b32(self.s_near_hi, (1 << 32) - kfield.Field.NEAR)
        A.s_mov_b32(self.s_near_lo, kfield.Field.NEAR)
        for d, s_ in zip(self.limbs(self.ACCq), self.ACCop):
            A.v_mov_b32(d, s_)
        A.s_waitcnt(lgkmcnt=0, regs=[self.planes, self.dpblk])
        voff8 = A.v("voff8e")
        A.v_lshrrev_b32(voff8, 1, self.voff)
        self.load_fe(self.CX, self.voff, "x01", "x23", True)
        self.load_fe(self.CY, self.voff, "y01", "y23", True)
        self.load_d(self.CD, voff8)
        k1 = A.s("k1e")
        A.s_add_u32(k1, self.k, 1)
        A.s_cmp("lt_u32", k1, self.G)
        A.s_cbranch_scc0(L_nb1)
        A.cur.schedule = False
        voffn = A.v("voffn_e")
        A.v_add_u32(voffn, self.stride, self.voff)
        self.load_fe(self.NB, voffn, "s01", "s23", False)
        A.s_branch(L_nbd)
        A.label(L_nb1)
        A.cur.schedule = False
        self.set_one(self.NB)
        A.label(L_nbd)
        A.cur.schedule = False
        A.s_waitcnt(vmcnt=0, regs=self.CX + self.CY + [self.CD] + self.NB)
        # ================= loop =================
        L_loop = ".Lkw_loop_%="
        self.in_loop = True
        if unroll == 1:
            self.iteration("a", set0, set1, L_loop, L_loop, L_exit, copy_back=True)
        else:
            L_b = ".Lkw_loopb_%="
            self.iteration("a", set0, set1, L_loop, L_b, L_exit)
            self.iteration("b", set1, set0, L_b, L_loop, L_exit)
        # ================= exit =================
        A.label(L_exit)
        A.cur.schedule = False
        for d, s_ in zip(self.ACCop, self.limbs(self.ACCq)):
            A.v_mov_b32(d, s_)  # (undefined after an exact-path exit: P6 may have started to overwrite it)
        A.s_waitcnt(vmcnt=0, lgkmcnt=0)
        A.s_nop(1)
        return self

    def iteration(self, tag, cur, nxt, L_top, L_next, L_exit, copy_back=False):
        """one kangaroo: state in `cur`, the next kangaroo's state is prefetched into (and the new running inverse written
        to) `nxt`.  Falls out of the loop to L_exit when k reaches G or a lane needs the exact path."""
        A, F = self.A, self.F
        T = lambda n: f"{n}_{tag}"  # noqa: E731
        L_rare, L_nodp, L_fix = f".Lkw_rare{tag}_%=", f".Lkw_nodp{tag}_%=", f".Lkw_fix{tag}_%="
        CX, CY, CD, INV, voff = cur["X"], cur["Y"], cur["D"], cur["INV"], cur["voff"]
        NX, NY, ND = nxt["X"], nxt["Y"], nxt["D"]
        A.label(L_top)
        A.cur.schedule = False
        # clamped strides for the prefetch: slot(k+1) if it exists else slot(k); slot(k+2) likewise
        s1, s2, k1, k2 = A.s(T("s1")), A.s(T("s2")), A.s(T("k1")), A.s(T("k2"))
        A.s_add_u32(k1, self.k, 1)
        A.s_cmp("lt_u32", k1, self.G)
        A.s_cselect_b32(s1, self.stride, 0)
        A.s_add_u32(k2, self.k, 2)
        A.s_cmp("lt_u32", k2, self.G)
        A.s_cselect_b32(s2, self.stride, 0)
        A.s_mov_b64(self.rare, 0)
        # ---------------- block A: everything up to the exactness check (scheduled as one graph)
        A.block("A")
        if F.flag_mode == "valu":
            F.begin_flags(T("fl"), self.s_near_hi, self.s_near_lo)
        voffn = A.v(T("voffn")) if copy_back else nxt["voff"]
        voffnn, voffn8 = A.v(T("voffnn")), A.v(T("voffn8"))
        early = os.environ.get("KASM_EARLYLD", "0") == "1"  # the prefetch at the very top of the iteration, not where the critical path leaves room
        A.v_add_u32(voffn, s1, voff).asap = early
        A.v_add_u32(voffnn, s2, voffn).asap = early
        A.v_lshrrev_b32(voffn8, 1, voffn).asap = early
        self.load_fe(NX, voffn, "x01", "x23", "X" not in self.nt, asap=early)
        self.load_fe(NY, voffn, "y01", "y23", "X" not in self.nt, asap=early)
        self.load_d(ND, voffn8, asap=early)
        # jump table entry j = x & 31
        cx, cy = self.limbs(CX), self.limbs(CY)
        jidx, laddr = A.v(T("jidx")), A.v(T("laddr"))
        A.v_and_b32(jidx, 31, cx[0])
        A.v_lshl_add_u32(laddr, jidx, 3, self.ldstab)
        JX, JY = [A.vt(T("jx0"), 4), A.vt(T("jx1"), 4)], [A.vt(T("jy0"), 4), A.vt(T("jy1"), 4)]
        JD = A.vt(T("jd"), 4)
        if "nolds" in self.abl:
            JX, JY, JD = CY, CX, CD
        else:
            A.ds_read2_b64(JX[0], laddr, 0, 32)
            A.ds_read2_b64(JX[1], laddr, 64, 96)
            A.ds_read2_b64(JY[0], laddr, 128, 160)
            A.ds_read2_b64(JY[1], laddr, 192, 224)
            A.ds_read_b64(JD.sub(0, 2), laddr, 2048)
            if not self.dsplit:
                A.ds_read_b64(JD.sub(2, 2), laddr, 2048 + 256)
            A.s_waitcnt(lgkmcnt=0, regs=JX + JY + [JD.sub(0, 2)] + ([] if self.dsplit else [JD.sub(2, 2)]))
        jx, jy = self.limbs(JX), self.limbs(JY)
        nb = self.limbs(self.NB)
        # P1: invk = inv * nb ; dx, dy ; P2: inv' = inv * dx
        IK = kfield.fe_mul(F, INV, nb, tag=T("p1"))
        # the product behind the next kangaroo, straight into the registers P1 has just read (write-after-read
        # dependencies place the loads behind P1's last multiply): no second register set, no moves
        no_s = "nos" in self.abl or ("nosA" in self.abl and tag == "a")
        if not no_s:
            self.load_fe(self.NB, voffnn, "s01", "s23", "S" in self.nt, asap=os.environ.get("KASM_EARLYLD", "0") == "1")
        dx = kfield.fe_sub(F, cx, jx, tag=T("dx"), k977_v=self.v977)
        dy = kfield.fe_sub(F, cy, jy, tag=T("dy"), k977_v=self.v977)
        INVn = kfield.fe_mul(F, INV, dx, out=nxt["INV"], tag=T("p2"), exact_tail=True)  # = 1 (mod p) behind the last kangaroo
        if "plusmulA" in self.abl and tag == "a":
            XM = kfield.fe_mul(F, dx, dy, tag=T("px"))  # measurement only: the extra product of the every-other-product form
            dy = kfield.fe_sub(F, XM, jy, tag=T("dyx"), k977_v=self.v977)  # (kept alive by feeding P3)
        # P3: s = dy * invk ; P4: s^2
        S = kfield.fe_mul(F, dy, IK, tag=T("p3"))
        SQ = kfield.fe_sqr(F, S, tag=T("p4"))
        # rx = s^2 - jx - cx ; ry = (cx - rx) * s - cy
        r0 = kfield.fe_sub(F, SQ, jx, tag=T("ra"), k977_v=self.v977)
        RXq = [A.vt(T("rx0"), 4), A.vt(T("rx1"), 4)]
        RX = kfield.fe_sub(F, r0, cx, out=self.limbs(RXq), tag=T("rx"), k977_v=self.v977)
        Tm = kfield.fe_sub(F, cx, RX, tag=T("t"), k977_v=self.v977)
        Y0 = kfield.fe_mul(F, Tm, S, tag=T("p5"))
        RYq = [A.vt(T("ry0"), 4), A.vt(T("ry1"), 4)]
        RY = kfield.fe_sub(F, Y0, cy, out=self.limbs(RYq), tag=T("ry"), k977_v=self.v977)
        # d += jD  (raw 128-bit add, GPUMath.h:119-121)
        DN = A.vt(T("dnew"), 4)
        cd = CD.regs
        dc = A.st(T("dcar"), 2)
        A.v_add_co_u32(DN[0], dc, cd[0], JD[0])
        A.v_addc_co_u32(DN[1], dc, cd[1], JD[1], dc)
        if self.dsplit:
            if self.carry_mode != "atomic":
                A.s_or_accum(self.rare, dc)  # carry out of the low word: the high word is updated on the exact path
        else:
            A.v_addc_co_u32(DN[2], dc, cd[2], JD[2], dc)
            A.v_addc_co_u32(DN[3], "vcc", cd[3], JD[3], dc)
        # prefix product of the next jump's dx: acc' = acc * (rx - J[rx & 31].x)
        jidx2, laddr2 = A.v(T("jidx2")), A.v(T("laddr2"))
        A.v_and_b32(jidx2, 31, RX[0])
        A.v_lshl_add_u32(laddr2, jidx2, 3, self.ldstab)
        JX2 = [A.vt(T("jxn0"), 4), A.vt(T("jxn1"), 4)]
        if "nolds" in self.abl:
            JX2 = CY
        else:
            A.ds_read2_b64(JX2[0], laddr2, 0, 32)
            A.ds_read2_b64(JX2[1], laddr2, 64, 96)
            A.s_waitcnt(lgkmcnt=0, regs=JX2)
        dx2 = kfield.fe_sub(F, RX, self.limbs(JX2), tag=T("dx2"), k977_v=self.v977)
        ACCq = self.ACCq
        ACCn = kfield.fe_mul(F, self.limbs(ACCq), dx2, out=self.limbs(ACCq), tag=T("p6"))  # in place (write-after-read ordered)
        # distinguished point?  (x.limb3 & dpMask) == 0, GPUCompute.h:96
        t1, t2 = A.v(T("dpt1")), A.v(T("dpt2"))
        A.v_and_b32(t1, self.dp_mask_lo, RX[6])
        A.v_and_b32(t2, self.dp_mask_hi, RX[7])
        A.v_or_b32(t1, t1, t2)
        DPM = A.st(T("dpm"), 2)
        A.v_cmp_eq_u32(DPM, 0, t1)
        if F.flag_mode == "valu":
            F.end_flags(T("fl"))
        # everything the stores and the commit need must be complete here
        A.keep(*INVn, *RX, *RY, *ACCn, *DN.regs[:2 if self.dsplit else 4], DPM, *([dc] if self.dsplit and self.carry_mode == "atomic" else []))
        A.s_cmp("lg_u64", self.rare, 0)
        A.s_cbranch_scc1(L_rare)
        # ---------------- block B: stores, DP records, commit
        A.cur.name = "B"
        voff8 = A.v(T("voff8"))
        A.v_lshrrev_b32(voff8, 1, voff)
        n_after = 0  # stores of this iteration issued behind the prefetch loads
        if "nostore" not in self.abl and "noglobal" not in self.abl:
            A.global_store(4, voff, RXq[0], self.P["x01"], nt="x" not in self.nt)
            A.global_store(4, voff, RXq[1], self.P["x23"], nt="x" not in self.nt)
            A.global_store(4, voff, RYq[0], self.P["y01"], nt="x" not in self.nt)
            A.global_store(4, voff, RYq[1], self.P["y23"], nt="x" not in self.nt)
            A.global_store(2, voff8, DN.sub(0, 2), self.P["dlo"], nt="d" in self.nt)
            n_after += 5
            if not self.dsplit:
                A.global_store(2, voff8, DN.sub(2, 2), self.P["dhi"], nt="d" in self.nt)
                n_after += 1
        if not no_s and "noglobal" not in self.abl:
            A.global_store(4, voff, ACCq[0], self.P["s01"], nt="s" in self.nt)
            A.global_store(4, voff, ACCq[1], self.P["s23"], nt="s" in self.nt)
            n_after += 2
        if self.dsplit and self.carry_mode == "atomic":
            # ---- cold: lanes whose low word carried add 1 to their high word, at L2, without waiting for anything
            L_nc = f".Lkw_nocarry{tag}_%="
            A.s_cmp("lg_u64", dc, 0)
            A.s_cbranch_scc0(L_nc)
            A.cur.schedule = False
            A.raw("; cold path")
            SAVEC, ONEC = A.st(T("savec"), 2), A.vt(T("onec"), 2)
            A.s_mov_b64(SAVEC, EXEC)
            A.v_mov_b32(ONEC[0], 1)
            A.v_mov_b32(ONEC[1], 0)
            A.s_mov_exec(dc)
            A.global_atomic_add_x2(voff8, ONEC, self.P["dhi"])
            A.s_mov_exec(SAVEC)
            A.label(L_nc)
            A.cur.schedule = False
        A.s_cmp("lg_u64", DPM, 0)
        A.s_cbranch_scc0(L_nodp)
        # ---- cold: wave-compacted DP records (emit_dp of kng_engine.hip; GPUCompute.h:96-105)
        A.cur.schedule = False
        A.raw("; cold path")
        SAVE, ONE, KEEP = A.st(T("save"), 2), A.st(T("one"), 2), A.st(T("keep"), 2)
        scnt, slead, sbase = A.s(T("scnt")), A.s(T("slead")), A.s(T("sbase"))
        vpos, vcnt, vzero, vbase, vrec = A.v(T("vpos")), A.v(T("vcnt")), A.v(T("vzero")), A.v(T("vbase")), A.v(T("vrec"))
        KQ = A.vt(T("kq"), 4)
        A.s_mov_b64(SAVE, EXEC)
        if self.dsplit:
            A.s_mov_exec(DPM)
            A.global_load(2, DN.sub(2, 2), voff8, self.P["dhi"], coherent=self.carry_mode == "atomic")  # (sees this kernel's L2 atomics)
            A.s_mov_exec(SAVE)
        A.s_bcnt1_i32_b64(scnt, DPM)
        A.v_mbcnt_lo(vpos, DPM[0], 0)
        A.v_mbcnt_hi(vpos, DPM[1], vpos)
        A.s_ff1_i32_b64(slead, DPM)
        A.s_lshl_b64(ONE, 1, slead)
        A.v_mov_b32(vcnt, scnt)
        A.v_mov_b32(vzero, 0)
        A.s_mov_exec(ONE)
        A.global_atomic_add_rtn(vbase, vzero, vcnt, self.dp_count)
        A.s_waitcnt(vmcnt=0, regs=[vbase] + ([DN.sub(2, 2)] if self.dsplit else []))
        A.v_readfirstlane_b32(sbase, vbase)
        A.s_mov_exec(DPM)
        A.v_add_u32(vpos, sbase, vpos)
        A.v_cmp_lt_u32(KEEP, vpos, self.max_found)
        A.v_lshlrev_b32(vrec, 6, vpos)
        A.v_lshrrev_b32(KQ[0], 4, voff)
        A.v_mov_b32(KQ[1], 0)
        A.v_mov_b32(KQ[2], 0)
        A.v_mov_b32(KQ[3], 0)
        A.s_and_b64(KEEP, KEEP, DPM)
        A.s_mov_exec(KEEP)
        A.global_store(4, vrec, RXq[0], self.dp_items)
        A.global_store(4, vrec, RXq[1], self.dp_items, offset=16)
        A.global_store(4, vrec, DN, self.dp_items, offset=32)
        A.global_store(4, vrec, KQ, self.dp_items, offset=48)
        A.s_mov_exec(SAVE)
        # ---- the prefetch must have landed: everything issued behind it may still be in flight
        A.label(L_nodp)
        A.cur.schedule = False
        A.s_waitcnt(vmcnt=n_after, regs=NX + NY + [ND.sub(0, 2)] + ([] if self.dsplit else [ND.sub(2, 2)]) + self.NB)
        if copy_back:
            A.block("commit")
            for d, s_ in zip(INV, INVn):
                A.v_mov_b32(d, s_)
            for dq, sq in ((CX, NX), (CY, NY)):
                for d, s_ in zip(self.limbs(dq), self.limbs(sq)):
                    A.v_mov_b32(d, s_)
            for i in range(2 if self.dsplit else 4):
                A.v_mov_b32(CD[i], ND[i])
            A.v_mov_b32(voff, voffn)
        A.block("next", schedule=False)
        k1b = A.s(T("k1b"))
        A.s_add_u32(self.k, self.k, 1)
        A.s_add_u32(k1b, self.k, 1)
        A.s_cmp("lt_u32", k1b, self.G)
        A.s_cbranch_scc1(L_fix)
        A.cur.schedule = False
        self.set_one(self.NB)  # the last kangaroo of the pass has no neighbour product: nb = 1 (walk_body: invk = inv)
        A.label(L_fix)
        A.cur.schedule = False
        A.s_cmp("lt_u32", self.k, self.G)
        A.s_cbranch_scc1(L_next)
        A.cur.schedule = False
        A.s_branch(L_exit)
        # ---- exact-path exit of this copy: the caller gets the running inverse as it was BEFORE this kangaroo
        A.label(L_rare)
        A.cur.schedule = False
        A.raw("; cold path")
        if INV is not self.INV:
            for d, s_ in zip(self.INV, INV):
                A.v_mov_b32(d, s_)
        A.s_branch(L_exit)


VPOOL = list(range(64, 256))
SPOOL = list(range(36, 100))


# the ALU ceiling of the loop, measured live by bench.py (engine option "asm" 2): the same schedule with every global and LDS
# access of the per-kangaroo loop and the collection of the exactness flags left out -- what remains is the VALU stream
VALU_ONLY = ("noglobal", "nolds", "noflags")


def generate(dsplit, vpool=VPOOL, spool=SPOOL, abl=None):
    lp = Loop(dsplit, abl).build()
    # scheduler pressure limits (live carry masks / live temporaries beyond which only instructions that free registers issue)
    kasm.schedule(lp.A, sgpr_limit=int(os.environ.get("KASM_SLIM", "28")), vgpr_limit=int(os.environ.get("KASM_VLIM", "150")))
    used = kasm.allocate(lp.A, vpool, spool)
    probs = kasm.verify(lp.A)
    return lp, used, probs


def c_string(lines):
    out = []
    pad = int(os.environ.get("KASM_PARANOID", "0"))  # debugging aid: s_nop behind every instruction (hazard or logic?)
    for t in lines:
        t = t.split("\t;")[0].rstrip()
        out.append('        "' + t.replace("\t", " ").strip() + '\\n"')
        if pad and not t.strip().endswith(":") and not t.strip().startswith(";"):
            out.append(f'        "s_nop {pad - 1}\\n"')
    return "\n".join(out)


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    parts = []
    info = {}
    for dsplit, abl in ((True, None), (False, None), (True, VALU_ONLY)):
        lp, used, probs = generate(dsplit, abl=abl)
        if probs:
            raise SystemExit("\n".join(probs))
        text = kasm.listing(lp.A, comments=False)
        st = kasm.stats(lp.A, blocks={"A", "B", "commit", "next"})
        if abl is None:
            info[dsplit] = (st, used)
        clob = [f'"v{n}"' for n in sorted(used["v"])] + [f'"s{n}"' for n in sorted(used["s"])] + ['"vcc"', '"scc"', '"memory"']
        name = "KNG_WALK_ASM_TEXT_VALU" if abl else ("KNG_WALK_ASM_TEXT_DSPLIT" if dsplit else "KNG_WALK_ASM_TEXT_FULL")
        what = ("MEASUREMENT ONLY, WRONG RESULTS ON PURPOSE: the DSPLIT loop without its global / LDS accesses and flag collection (ALU ceiling, option \"asm\" 2)"
                if abl else ('low word streams (DSPLIT)' if dsplit else 'both words stream'))
        parts.append(f"// distance layout: {what}; hot blocks: {st}\n"
                     f"#define {name} \\\n" + " \\\n".join(c_string(text).split("\n")) + "\n"
                     f"#define {name.replace('TEXT', 'CLOBBERS')} {', '.join(clob)}\n")
    hdr = '''// GENERATED by tools/gen_walk_asm.py (tools/kasm.py scheduler + allocator, tools/kfield.py arithmetic) -- do not edit.
// The per-kangaroo loop of the walk kernel as one scheduled asm statement; see the generator for the protocol.
// Replaces the loop body of ComputeKangaroos (GPU/GPUCompute.h:52-105 of the reference).
#pragma once

''' + "\n".join(parts) + '''
// operands: %0-%7 inv, %8-%15 acc, %16 k, %17 voff (all read-write); %18 args, %19 stride, %20 G, %21 lds table address
#define KNG_WALK_ASM_LOOP(TEXT, CLOBBERS, inv, acc, k, voff, args, stride, G, ldstab)                                         \\
    asm volatile(TEXT                                                                                                          \\
                 : "+v"(inv[0]), "+v"(inv[1]), "+v"(inv[2]), "+v"(inv[3]), "+v"(inv[4]), "+v"(inv[5]), "+v"(inv[6]), "+v"(inv[7]), \\
                   "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]), \\
                   "+s"(k), "+v"(voff)                                                                                         \\
                 : "s"(args), "s"(stride), "s"(G), "s"(ldstab)                                                                 \\
                 : CLOBBERS)
'''
    path = os.path.join(root, "kangaroo_amd", "csrc", "kng_walk_asm.h")
    open(path, "w").write(hdr)
    for ds, (st, used) in info.items():
        print(f"dsplit={ds}: hot blocks {st}; {len(used['v'])} VGPRs (max v{max(used['v'])}), {len(used['s'])} SGPRs")
    print("wrote", path)


if __name__ == "__main__":
    main()
