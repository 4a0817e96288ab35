#!/usr/bin/env python3
"""kfield -- secp256k1 field arithmetic on 32-bit limbs, written against tools/kasm.py's builder.

Same integers as kangaroo_amd/csrc/kng_field.h (which restates GPU/GPUMath.h:476-516, :810-907 of the reference):
  mul  : 256x256 -> 512 Comba (product scanning), v_mad_u64_u32 with the carry in an SGPR pair, one v_addc per MAD
  fold : S = lo + hi*(2^32+977) in ONE carry chain (E/O trick of kng_field.h fe_fold32), T = S >> 256 folded again,
         last carry dropped.  Conditions under which the short form is not exact are ORed into `rare`.
  sub  : a - b, + p on borrow (exact except for the borrow leaving the low 64 bits, which is ORed into `rare`)
Everything is emitted in dependency order only; kasm.schedule() interleaves the chains so that the 2 wait states
between a VALU carry write and its VALU reader are filled by independent work instead of s_nop.

Register discipline of the product (one v_mov per column): even column 2m accumulates in the pair
PW[m] whose halves end up as (w[2m], w[2m+1]) -- the aligned pairs the fold's 64-bit addends need; odd columns use a
temporary pair.  gfx90a+ wants 64-bit VGPR operands even-aligned, so the high half of one column's accumulator cannot
serve as the low half of the next one's: that costs the move.
"""
from __future__ import annotations

from kasm import Asm

DUMMY = "vcc"  # carry-outs nobody reads go to vcc, untracked: scheduled code never keeps a live value in vcc

P = (1 << 256) - (1 << 32) - 977
K = (1 << 32) + 977


class Field:
    def __init__(self, A: Asm, k977_s, rare):
        """k977_s: SGPR holding 977; rare: SGPR pair that collects "short form not exact" lane masks"""
        self.A, self.k977, self.rare = A, k977_s, rare
        self.n = 0
        # a column's first multiply-add starts from a carry-in below 2^36 and overflows 64 bits only when its product is
        # within 2^36 of 2^64 (~2^-28): flag it for the exact path instead of counting it with a v_addc
        self.elide_first_carry = False
        # How the "short form not exact" conditions reach `rare`.
        #   "salu": every carry-out that signals one is ORed into `rare` (one s_or_b64 per condition: 11 per fold, 2 per sub)
        #   "valu": a SUPERSET of the conditions is tracked in two VGPRs instead -- every condition needs some word within
        #           1024 of 2^32 (resp. below 1024), so a running v_max3_u32 / v_min3_u32 over those words and ONE
        #           comparison each per iteration decide (begin_flags / end_flags).  5 VALU per fold and 1 per sub replace
        #           11 + 2 scalar ORs, each of which stalls on the VALU-written lane masks it reads (profiles/r03_stalls.txt).
        #           The price: ~1.2e-3 of the wave-iterations take the exact path instead of ~0.6e-3.
        self.flag_mode = "salu"
        self.mx = self.mn = None
        self.pending_near_max = []  # words comba() wants tracked with its product's fold (first-carry elision)

    # ---- "valu" flag mode: per scheduled region, begin_flags() ... arithmetic ... end_flags()
    NEAR = 1024  # every flagged condition implies a word >= 2^32 - NEAR (max side) or < NEAR (min side); 977 + 1 < NEAR

    def begin_flags(self, tag, s_hi, s_lo):
        """s_hi / s_lo: SGPRs holding 2^32 - NEAR and NEAR (VOP3 takes no literal on gfx9)"""
        A = self.A
        self.mx, self.mn = A.v(f"{tag}mx"), A.v(f"{tag}mn")
        self.s_hi, self.s_lo = s_hi, s_lo
        A.v_mov_b32(self.mx, 0)
        A.v_mov_b32(self.mn, -1)

    def end_flags(self, tag):
        """ORs the two verdicts into `rare`"""
        A = self.A
        f1, f2 = A.st(f"{tag}fmx", 2), A.st(f"{tag}fmn", 2)
        A.v_cmp_le_u32(f1, self.s_hi, self.mx)  # some tracked word >= 2^32 - NEAR
        A.v_cmp_gt_u32(f2, self.s_lo, self.mn)  # some tracked word < NEAR
        A.s_or_accum(self.rare, f1)
        A.s_or_accum(self.rare, f2)
        self.mx = self.mn = None

    def near_max(self, words, tag):
        """track words of which one must be within NEAR of 2^32 for a condition to hold"""
        A = self.A
        words = list(words)
        loc = None
        # local tree first (independent of the running maximum), one link into the running maximum at the end
        while len(words) > 2 or (loc is not None and len(words) > 1):
            if loc is None:
                loc = A.v(self.uid(f"{tag}lm"))
                A._v2("v_max3_u32", loc, words[0], words[1], words[2]).asap = True
                words = words[3:]
            else:
                nl = A.v(self.uid(f"{tag}lm"))
                A._v2("v_max3_u32", nl, loc, words[0], words[1]).asap = True
                loc, words = nl, words[2:]
        rest = ([loc] if loc is not None else []) + words
        assert 1 <= len(rest) <= 2
        A.v_accum3("v_max3_u32", self.mx, rest[0], rest[-1]).asap = True

    def near_min(self, a, b):
        self.A.v_accum3("v_min3_u32", self.mn, a, b).asap = True

    def uid(self, stem):
        self.n += 1
        return f"{stem}{self.n}"


def comba(F: Field, a, b, tag="m"):
    """a, b: 8 Regs each.  Returns PW: 8 pairs, PW[m] = (w[2m], w[2m+1]) (module docstring: register discipline)."""
    A = F.A
    PW = [A.vt(F.uid(f"{tag}pw"), 2) for _ in range(8)]
    # accumulator of every column, created up front: even columns in PW, odd ones in temporaries
    accs = [PW[k // 2] if k % 2 == 0 else A.vt(F.uid(f"{tag}t"), 2) for k in range(15)]
    for k in range(15):
        acc = accs[k]
        terms = [(i, k - i) for i in range(max(0, k - 7), min(7, k) + 1)]
        if k == 0:
            addend = 0
        else:
            prev = accs[k - 1]
            A.v_mov_b32(acc.lo, prev.hi, comment=f"col {k} <- carry word of col {k - 1}")
            if k % 2 == 0:
                # prev is an odd column's temporary: its low word is w[k-1]; park it beside w[k-2]
                A.v_mov_b32(PW[(k - 1) // 2].hi, prev.lo, comment=f"pack w{k - 1}")
            if k == 1:
                A.v_mov_b32(acc.hi, 0)  # column 0 cannot carry
            addend = acc
        can_overflow = 0 < k < 14
        carries = []
        for n_, (i, j) in enumerate(terms):
            # column 1 starts from X0 < 2^32: a*b + X0 <= 2^64 - 2^32, its first product cannot overflow at all
            first_safe = n_ == 0 and (k == 1 or F.elide_first_carry)
            elided = can_overflow and n_ == 0 and k > 1 and F.elide_first_carry
            c = A.st(F.uid("c"), 2) if can_overflow and not (n_ == 0 and k == 1) and not (elided and F.flag_mode == "valu") else DUMMY
            A.v_mad_u64_u32(acc, c, a[i], b[j], addend if n_ == 0 else acc, comment=f"a{i}*b{j}")
            if elided:
                # the first product of a column starts from (carry word, carry count) < 9 * 2^32: it overflows 64 bits only when
                # BOTH factors are within 10 of 2^32.  The column's first factors are a0 (k <= 7) or b7 (k > 7).
                if F.flag_mode == "valu":
                    F.pending_near_max = [a[0], b[7]]  # fold() tracks them with the product's other words
                else:
                    A.s_or_accum(F.rare, c)
            elif can_overflow and not first_safe:
                carries.append(c)
        if carries:
            nhi = accs[k + 1].hi
            for n_, c in enumerate(carries):
                A.v_addc_co_u32(nhi, DUMMY, 0, 0 if n_ == 0 else nhi, c)
    # column 13's low word (w13) still sits in its temporary; column 14 is PW[7] = (w14, w15)
    # (the k == 14 iteration above packed w13 into PW[6].hi)
    return PW


def comba_sqr(F: Field, a, tag="q"):
    """a: 8 Regs.  a^2 = 2*O + D with O = sum_{i<j} a_i a_j 2^(32(i+j)) (28 MADs through the same column machinery) and
    D = sum a_i^2 2^(64 i), whose eight 64-bit squares do not overlap (8 MADs, no carries); the doubling is one
    v_alignbit_b32 per limb feeding the final 16-limb carry chain, which writes the (w[2m], w[2m+1]) pairs directly.
    Same 512-bit integer as comba(a, a): 36 instead of 64 multiplies (kng_mul32.h sqr_wide32 is the round-1 form)."""
    A = F.A
    PW = [A.vt(F.uid(f"{tag}pw"), 2) for _ in range(8)]
    D = [A.vt(F.uid(f"{tag}d"), 2) for _ in range(8)]
    for m in range(8):
        A.v_mad_u64_u32(D[m], DUMMY, a[m], a[m], 0, comment=f"a{m}^2")
    accs = {k: A.vt(F.uid(f"{tag}o"), 2) for k in range(1, 14)}
    o = [None] * 16  # o[k] = limb k of O
    hi_last = None
    for k in range(1, 14):
        acc = accs[k]
        terms = [(i, k - i) for i in range(max(0, k - 7), min(7, k) + 1) if i < k - i]
        if k == 1:
            addend = 0
        else:
            prev = accs[k - 1]
            A.v_mov_b32(acc.lo, prev.hi, comment=f"col {k} <- carry word of col {k - 1}")
            if k <= 3:
                A.v_mov_b32(acc.hi, 0)  # columns 1 and 2 (one product each, small carry-in) cannot overflow 64 bits
            addend = acc
        can_overflow = k >= 3
        carries = []
        for n_, (i, j) in enumerate(terms):
            elided = can_overflow and n_ == 0 and F.elide_first_carry
            c = A.st(F.uid("c"), 2) if can_overflow and not (elided and F.flag_mode == "valu") else DUMMY
            A.v_mad_u64_u32(acc, c, a[i], a[j], addend if n_ == 0 else acc, comment=f"a{i}*a{j}")
            if elided:
                # as in comba(): the column's first product a_i * a_j has i = 0 (k <= 7) or j = 7 (k > 7)
                if F.flag_mode == "valu":
                    F.pending_near_max = [a[0], a[7]]
                else:
                    A.s_or_accum(F.rare, c)
            elif can_overflow:
                carries.append(c)
        if can_overflow:
            if k < 13:
                nhi = accs[k + 1].hi
            else:
                nhi = hi_last = A.v(F.uid(f"{tag}o15"))
            if not carries:
                A.v_mov_b32(nhi, 0)
            for n_, c in enumerate(carries):
                A.v_addc_co_u32(nhi, DUMMY, 0, 0 if n_ == 0 else nhi, c)
        o[k] = acc.lo
    o[14], o[15] = accs[13].hi, hi_last
    # w = 2*O + D
    C = A.st(F.uid("qc"), 2)
    A.v_mov_b32(PW[0].lo, D[0].lo, comment="w0 = a0^2 low (O has no limb 0)")
    for k in range(1, 16):
        dbl = A.v(F.uid(f"{tag}dbl"))
        if k == 1:
            A.v_lshlrev_b32(dbl, 1, o[1])
        else:
            A.v_alignbit_b32(dbl, o[k], o[k - 1], 31, comment=f"(2*O) limb {k}")
        d_k = D[k // 2][k % 2]
        w_k = PW[k // 2][k % 2]
        if k == 1:
            A.v_add_co_u32(w_k, C, dbl, d_k)
        elif k < 15:
            A.v_addc_co_u32(w_k, C, dbl, d_k, C)
        else:
            A.v_addc_co_u32(w_k, DUMMY, dbl, d_k, C)  # a^2 < 2^512
    return PW


def fe_sqr(F: Field, a, out=None, tag="q"):
    return fold(F, comba_sqr(F, a, tag), out, tag)


def fold(F: Field, PW, out=None, tag="f", exact_tail=False):
    """PW: 8 pairs (w[2m], w[2m+1]).  out: optional list of 8 Regs to receive the result (out[0], out[1] must be an
    even-aligned pair).  Returns the 8 result registers."""
    A = F.A
    e = [A.vt(F.uid(f"{tag}e"), 2) for _ in range(4)]
    o = [A.vt(F.uid(f"{tag}o"), 2) for _ in range(4)]
    flags = []
    for j in range(4):
        ce, co = A.st(F.uid("ce"), 2), A.st(F.uid("co"), 2)
        A.v_mad_u64_u32(e[j], ce, PW[4 + j].lo, F.k977, PW[j], comment=f"E{j} = w{8 + 2 * j}*977 + (w{2 * j},w{2 * j + 1})")
        A.v_mad_u64_u32(o[j], co, PW[4 + j].hi, F.k977, PW[4 + j], comment=f"O{j} = w{9 + 2 * j}*977 + (w{8 + 2 * j},w{9 + 2 * j})")
        flags += [ce, co]
    if out is None:
        r01 = A.vt(F.uid(f"{tag}r"), 2)
        out = [r01.lo, r01.hi] + [A.v(F.uid(f"{tag}r")) for _ in range(6)]
    elif out[0].tup is not None and out[1].tup is out[0].tup and out[0].idx % 2 == 0 and out[1].idx == out[0].idx + 1:
        r01 = out[0].tup.sub(out[0].idx, 2) if len(out[0].tup) > 2 else out[0].tup
    else:
        r01 = A.vt(F.uid(f"{tag}r"), 2)  # out[0], out[1] are no aligned pair (compiler-allocated operands): one move
    # S = E + O<<32 : one chain.  s0 = e0.lo, s1 written over e0.hi so that (s0, s1) stays an aligned pair
    C = A.st(F.uid("fc"), 2)
    A.v_add_co_u32(e[0].hi, C, e[0].hi, o[0].lo, comment="s1")
    s = [e[0].lo, e[0].hi, None, None, None, None, None, None, None]
    s2 = A.v(F.uid(f"{tag}s2"))
    A.v_addc_co_u32(s2, C, e[1].lo, o[0].hi, C, comment="s2")
    s[2] = s2
    srcs = {3: (e[1].hi, o[1].lo), 4: (e[2].lo, o[1].hi), 5: (e[2].hi, o[2].lo), 6: (e[3].lo, o[2].hi), 7: (e[3].hi, o[3].lo)}
    for i in range(3, 8):
        A.v_addc_co_u32(out[i], C, srcs[i][0], srcs[i][1], C, comment=f"s{i}")
    s8 = A.v(F.uid(f"{tag}s8"))
    A.v_addc_co_u32(s8, C, o[3].hi, 0, C, comment="s8; carry-out = bit 32 of T")
    flags.append(C)  # top: T >= 2^32 (hi7 within 978 of 2^32)
    # second fold: T*K = s8*977 + (s8 << 32)
    c1 = A.st(F.uid("c1"), 2)
    A.v_mad_u64_u32(r01, c1, s8, F.k977, e[0], comment="(r0,r1') = s8*977 + (s0,s1)")
    flags.append(c1)
    C2 = A.st(F.uid("fd"), 2)
    A.v_add_co_u32(out[1], C2, r01[1], s8, comment="r1")
    if out[0] is not r01[0]:
        A.v_mov_b32(out[0], r01[0], comment="r0")
    A.v_addc_co_u32(out[2], C2, s2, 0, C2, comment="r2; carry-out leaves limb 2 only when it was all ones")
    flags.append(C2)
    if F.flag_mode == "valu":
        # every condition above needs one of these words within 1024 of 2^32: the odd words of W (E_j: w[2j+1], O_j and the
        # top carry: w[9+2j]), s1 (the second fold's MAD) and s2 (the ripple beyond limb 2) -- see the notes at each flag
        # exact_tail: a product that is congruent to a small number (inv * dx = 1 behind the last kangaroo of a pass) has
        # S = p + small, i.e. s1.. all ones in EVERY lane: its two tail conditions keep their exact flags
        extra, F.pending_near_max = F.pending_near_max, []
        if exact_tail:
            F.near_max([PW[m].hi for m in range(8)] + extra, tag)
            A.s_or_accum(F.rare, c1)
            A.s_or_accum(F.rare, C2)
        else:
            F.near_max([PW[m].hi for m in range(8)] + [s[1], s2] + extra, tag)
    else:
        for m in flags:
            A.s_or_accum(F.rare, m)
    return out


def fe_mul(F: Field, a, b, out=None, tag="m", exact_tail=False):
    return fold(F, comba(F, a, b, tag), out, tag, exact_tail)


def fe_sub(F: Field, x, y, out=None, tag="s", k977_v=None):
    """r = x - y, + p on borrow.  k977_v: VGPR holding 977 (v_cndmask cannot take two SGPRs)."""
    A = F.A
    if out is None:
        out = [A.v(F.uid(f"{tag}r")) for _ in range(8)]
    B = A.st(F.uid("sb"), 2)
    t = [A.v(F.uid(f"{tag}t")) for _ in range(2)]
    A.v_sub_co_u32(t[0], B, x[0], y[0])
    A.v_subb_co_u32(t[1], B, x[1], y[1], B)
    for i in range(2, 8):
        A.v_subb_co_u32(out[i], B, x[i], y[i], B)
    # borrow: subtract 2^256 - p = 2^32 + 977.  The 2^32 goes in as the BORROW-IN of limb 1 (the lane mask B itself);
    # the borrow out of limb 0 (t0 < 977, once in 2^22 borrowing subtractions) and the one out of limb 1 both mean
    # "exact path" -- so no second select and no second chain link.
    q0 = A.v(F.uid(f"{tag}q"))
    A.v_cndmask_b32(q0, 0, k977_v, B)
    D, E = A.st(F.uid("sd"), 2), A.st(F.uid("se"), 2)
    A.v_sub_co_u32(out[0], D, t[0], q0)
    A.v_subb_co_u32(out[1], E, t[1], 0, B)
    if F.flag_mode == "valu":
        F.near_min(t[0], t[1])  # D needs t0 < 977, E needs t1 = 0
    else:
        A.s_or_accum(F.rare, D)
        A.s_or_accum(F.rare, E)
    return out


# ---- plain-integer restatements (what the emulated code must reproduce, and when it may set `rare`)


def ref_mul(a, b):
    """GPUMath.h:810-858 / IntMod.cpp:873-950: fold twice, drop the last carry, no comparison with p"""
    w = a * b
    lo, hi = w & ((1 << 256) - 1), w >> 256
    s = lo + hi * K
    t = s >> 256
    return ((s & ((1 << 256) - 1)) + t * K) & ((1 << 256) - 1)


def ref_sub(a, b):
    r = a - b
    if r < 0:
        r += P
    return r & ((1 << 256) - 1)


def limbs(x, n=8):
    return [(x >> (32 * i)) & 0xFFFFFFFF for i in range(n)]


def unlimbs(v):
    return sum(int(x) << (32 * i) for i, x in enumerate(v))
