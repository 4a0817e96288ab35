"""The generated gfx950 code (tools/kasm.py + tools/kfield.py + tools/gen_*.py), executed on the CPU.

`tools/kasm_emu.py` runs the PRINTED text -- final schedule, physical registers -- lane by lane, so these tests cover the
arithmetic, the list scheduler's ordering and the register allocator without a GPU:
  * fe_mul as one statement (kng_mulasm.h) and fe_sub against the reference's golden vectors; a lane may only differ
    from the reference when it raised the exact-path flag;
  * the whole walk loop (kng_walk_asm.h) against an integer model of walk_body's data flow: states, running products,
    DP records, both distance layouts, G = 1 and 2, exact-path exits and re-entry, 64-lane waves;
  * the static verifier finds no hazard / s_waitcnt problem, and the headers in the tree are what the generators emit.
Hazard TIMING cannot be emulated; the distances are LLVM's (kasm.WS_*) and are checked statically by kasm.verify.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import gen_mulasm  # noqa: E402
import gen_walk_asm  # noqa: E402
import kasm  # noqa: E402
import kfield  # noqa: E402
import kwalk_emu  # noqa: E402
from kasm_emu import Emu  # noqa: E402


def _run_binary_op(text, operands, cases, W=16):
    """operands: dict name -> list of 8 register numbers (a, b, r) + 'rare' pair + 'k977'"""
    out = []
    for c0 in range(0, len(cases), W):
        chunk = cases[c0:c0 + W]
        e = Emu(lanes=W)
        e.exec = (1 << len(chunk)) - 1
        for l, (x, y) in enumerate(chunk):
            for i in range(8):
                e.v[operands["a"][i]][l] = (x >> (32 * i)) & 0xFFFFFFFF
                e.v[operands["b"][i]][l] = (y >> (32 * i)) & 0xFFFFFFFF
            if "v977" in operands:
                e.v[operands["v977"]][l] = 977
        e.s[operands["k977"]] = 977
        e.run(text)
        rare = e.s[operands["rare"]] | (e.s[operands["rare"] + 1] << 32)
        for l in range(len(chunk)):
            out.append((sum(e.v[operands["r"][i]][l] << (32 * i) for i in range(8)), (rare >> l) & 1))
    return out


def test_fe_mul_statement_against_golden_vectors(golden):
    """kng_mulasm.h's statement with its operands bound to registers: every golden ModMulK1 vector (incl. the lazy-fold
    corner cases) is reproduced exactly, or the lane asks for the exact path -- and only edge operands do."""
    A, used = gen_mulasm.build()
    text = kasm.listing(A, comments=False)
    bind = {f"%{i}": f"v{i}" for i in range(8)}
    bind.update({f"%{9 + i}": f"v{16 + i}" for i in range(8)})
    bind.update({f"%{17 + i}": f"v{32 + i}" for i in range(8)})
    bind.update({"%8": "s[2:3]", "%25": "s1"})
    text = [re.sub(r"%(\d+)", lambda m: bind["%" + m.group(1)], t) for t in text]
    assert min(used["v"]) > 40 and min(used["s"]) > 3
    cases = [(int(a, 16), int(b, 16)) for a, b, _ in golden["modmul"]]
    want = [int(r, 16) & ((1 << 256) - 1) for _, _, r in golden["modmul"]]
    got = _run_binary_op(text, {"a": list(range(16, 24)), "b": list(range(32, 40)), "r": list(range(8)), "rare": 2, "k977": 1}, cases)
    flagged = 0
    for (g, rare), w, (a, b) in zip(got, want, cases):
        assert kfield.ref_mul(a, b) == w  # the integer restatement the loop model uses is the reference's integer
        if rare:
            flagged += 1
        else:
            assert g == w, f"{a:x} * {b:x}"
    # random operands practically never need the exact path; the golden set is edge-heavy
    import random

    rnd = random.Random(5)
    rcases = [(rnd.getrandbits(256), rnd.getrandbits(256)) for _ in range(256)]
    rgot = _run_binary_op(text, {"a": list(range(16, 24)), "b": list(range(32, 40)), "r": list(range(8)), "rare": 2, "k977": 1}, rcases)
    assert all(not rare and g == kfield.ref_mul(a, b) for (g, rare), (a, b) in zip(rgot, rcases))
    assert 0 < flagged < len(cases) // 2


def test_fe_sub_against_golden_vectors(golden):
    A = kasm.Asm()
    a = [A.v(f"a{i}", pinned=True) for i in range(8)]
    b = [A.v(f"b{i}", pinned=True) for i in range(8)]
    rare, k977, v977 = A.st("rare", 2, pinned=True), A.s("k977", pinned=True), A.v("v977", pinned=True)
    F = kfield.Field(A, k977, rare)
    out = kfield.fe_sub(F, a, b, k977_v=v977)
    A.keep(*out, rare, k977)
    kasm.schedule(A)
    kasm.allocate(A, list(range(0, 64)), list(range(0, 32)))
    assert not kasm.verify(A)
    text = kasm.listing(A, comments=False)
    cases = [(int(x, 16), int(y, 16)) for x, y, _ in golden["modsub"]]
    want = [int(r, 16) & ((1 << 256) - 1) for _, _, r in golden["modsub"]]  # the reference Int is 320-bit signed: low 256 bits
    ops = {"a": [r.phys for r in a], "b": [r.phys for r in b], "r": [r.phys for r in out], "rare": rare[0].phys, "k977": k977.phys, "v977": v977.phys}
    got = _run_binary_op(text, ops, cases)
    for (g, rare_), w, (x, y) in zip(got, want, cases):
        assert kfield.ref_sub(x, y) == w
        assert rare_ or g == w, f"{x:x} - {y:x}"
    assert sum(r for _, r in got) < len(cases) // 4


@pytest.mark.parametrize("dsplit", [True, False])
def test_walk_loop_verifies_clean(dsplit):
    lp, used, problems = gen_walk_asm.generate(dsplit)
    assert problems == []
    st = kasm.stats(lp.A, blocks={"A", "B", "commit", "next"})
    assert st.get("nop", 0) <= 4, st  # the point of the exercise: independent work pads the carry chains
    assert max(used["v"]) < 256 and max(used["s"]) < 100


@pytest.mark.parametrize("case", [
    dict(L=8, G=5, steps=3, dsplit=True),
    dict(L=8, G=5, steps=2, dsplit=False, jd_bits=100),
    dict(L=8, G=1, steps=2, dsplit=True),
    dict(L=8, G=2, steps=3, dsplit=False, jd_bits=90),
    dict(L=8, G=4, steps=2, dsplit=True, jd_bits=64, seed=3),   # low-word carries in most iterations: added to the high words by L2 atomics, in the loop
    dict(L=64, G=3, steps=2, dsplit=True, dp_bits=2, jd_bits=40, seed=5),  # full wave, several DPs per wave-iteration
    dict(L=64, G=2, steps=2, dsplit=False, dp_bits=1, jd_bits=70, seed=6),
])
def test_walk_loop_against_integer_model(case):
    stats = kwalk_emu.run_case(verbose=False, **case)
    if case.get("jd_bits") == 64:
        assert stats.get("rare_exits", 0) == 0  # (rounds 1-3a left the loop for every carry; the states above include the high words)


def test_walk_loop_dp_overflow_is_counted_not_stored():
    """records beyond max_found are counted (the host reports them lost) but not written"""
    jx, jy, jd = kwalk_emu.random_table(109, 40)  # (not the model's seed: equal seeds put kangaroo 17 ON jump point 17, dx = 0)
    m = kwalk_emu.Model(64, 2, jx, jy, jd, dp_mask=0, seed=9)  # mask 0: every point is distinguished
    acc = m.pass0()
    h = kwalk_emu.Harness(m, True, max_found=50)
    inv = [pow(a % kfield.P, kfield.P - 2, kfield.P) for a in acc]
    h.run_step(inv, acc, backward=True, stats={})
    n, recs = h.dp_records()
    assert n == 128 and len(recs) == 50
    assert set(recs) <= set(m.dps) and len(set(recs)) == 50


def test_generated_headers_are_current():
    for gen, hdr in (("gen_mulasm.py", "kng_mulasm.h"), ("gen_walk_asm.py", "kng_walk_asm.h")):
        path = os.path.join(ROOT, "kangaroo_amd", "csrc", hdr)
        before = open(path).read()
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", gen)], check=True, capture_output=True)
        after = open(path).read()
        assert before == after, f"{hdr} is stale: run tools/{gen}"


def test_valu_flag_mode_is_a_superset_on_the_golden_vectors(golden):
    """flag mode "valu" (the walk loop's default): the exactness conditions are not ORed one by one but bounded by a
    running v_max3 / v_min3 over the words one of which must be within 1024 of 2^32 (resp. below 1024) for any condition to
    hold.  Over every golden ModMulK1 / ModSub vector (edge-heavy: dropped carries, borrow ripples) a lane may differ from
    the reference only when flagged -- for the plain fold and for the exact-tail form the loop uses for inv * dx."""
    for exact_tail, elide in ((False, False), (True, False), (False, True), (True, True)):
        A = kasm.Asm()
        a = [A.v(f"a{i}", pinned=True) for i in range(8)]
        b = [A.v(f"b{i}", pinned=True) for i in range(8)]
        rare, k977, v977 = A.st("rare", 2, pinned=True), A.s("k977", pinned=True), A.v("v977", pinned=True)
        shi, slo = A.s("shi", pinned=True), A.s("slo", pinned=True)
        F = kfield.Field(A, k977, rare)
        F.flag_mode = "valu"
        F.elide_first_carry = elide  # a column's first carry dropped: its superset is "a0 or b7 within 1024 of 2^32"
        A.block("A")
        A.s_mov_b32(shi, (1 << 32) - kfield.Field.NEAR)
        A.s_mov_b32(slo, kfield.Field.NEAR)
        F.begin_flags("t", shi, slo)
        prod = kfield.fe_mul(F, a, b, exact_tail=exact_tail)
        diff = kfield.fe_sub(F, a, b, k977_v=v977)
        F.end_flags("t")
        A.keep(*prod, *diff, rare, k977)
        kasm.schedule(A)
        kasm.allocate(A, list(range(0, 160)), list(range(0, 64)))
        assert not kasm.verify(A)
        text = kasm.listing(A, comments=False)
        ops = {"a": [r.phys for r in a], "b": [r.phys for r in b], "rare": rare[0].phys, "k977": k977.phys, "v977": v977.phys}
        for name, ref, out in (("modmul", kfield.ref_mul, prod), ("modsub", kfield.ref_sub, diff)):
            cases = [(int(x, 16) & ((1 << 256) - 1), int(y, 16) & ((1 << 256) - 1)) for x, y, _ in golden[name]]
            got = _run_binary_op(text, dict(ops, r=[r.phys for r in out]), cases)
            flagged = sum(r for _, r in got)
            for (g, r), (x, y) in zip(got, cases):
                assert r or g == ref(x, y), f"{name} {x:x} {y:x}"
            assert flagged < len(cases) // 2, (name, flagged)
        import random

        rnd = random.Random(11)
        # first-carry corner: both first factors of a column within 10 of 2^32 (a0 with b_k, a_k with b7), all limbs large
        M = (1 << 256) - 1
        edge = [((M - rnd.getrandbits(3)) & M, (M - (rnd.getrandbits(3) << 224)) & M) for _ in range(32)]
        edge += [(((1 << 32) - 1 - rnd.getrandbits(3)) | (rnd.getrandbits(224) << 32), M - rnd.getrandbits(200)) for _ in range(32)]
        egot = _run_binary_op(text, dict(ops, r=[r.phys for r in prod]), edge)
        assert all(r or g == kfield.ref_mul(x, y) for (g, r), (x, y) in zip(egot, edge))
        assert not elide or sum(r for _, r in egot) == len(edge)  # a0 / b7 within 1024 of 2^32: always flagged when elided
        rcases = [(rnd.getrandbits(256), rnd.getrandbits(256)) for _ in range(128)]
        rgot = _run_binary_op(text, dict(ops, r=[r.phys for r in prod]), rcases)
        assert all(not r and g == kfield.ref_mul(x, y) for (g, r), (x, y) in zip(rgot, rcases))  # random operands: never flagged


# ---- the toolkit itself: what the walk loop's correctness on the GPU rests on besides the arithmetic


def _tiny(A, schedule=True):
    """(a, b pinned inputs) -> Asm with one scheduled block"""
    A.block("A", schedule=schedule)


def test_verifier_reports_the_hazards_hipcc_does_not_pad():
    """gfx940+: a VALU that reads an SGPR written by the VALU instruction right before it needs 2 wait states in between,
    a VMEM instruction 5; a VALU that overwrites 128-bit store data 2.  Unscheduled text with the violation is reported,
    the scheduler's output of the same program is clean and contains the pads (nothing else to fill with)."""
    def build(schedule):
        A = kasm.Asm()
        x, y, z = A.v("x", pinned=True), A.v("y", pinned=True), A.v("z", pinned=True)
        q = A.vt("q", 4, pinned=True)
        base = A.st("base", 2, pinned=True)
        c = A.st("c", 2)
        A.block("A", schedule=schedule)
        A.v_add_co_u32(x, c, x, y)          # VALU writes the SGPR pair c
        A.v_addc_co_u32(z, kfield.DUMMY, z, y, c)  # ... and the next VALU reads it: 2 wait states
        A.global_store(4, x, q, base)
        A.v_mov_b32(q[0], y)                # overwrites 128-bit store data: 2 wait states
        A.keep(x, z, q)
        return A

    raw = build(False)
    for b in raw.blocks:
        b.schedule = False
    # emit exactly as written (no scheduling pass): the verifier must object
    kasm.allocate(raw, list(range(0, 32)), list(range(0, 32)))
    probs = kasm.verify(raw)
    assert any("hazard" in p and "v_addc_co_u32" in p for p in probs), probs
    assert any("hazard" in p and "v_mov_b32" in p for p in probs), probs
    fixed = build(True)
    kasm.schedule(fixed)
    kasm.allocate(fixed, list(range(0, 32)), list(range(0, 32)))
    assert kasm.verify(fixed) == []
    ops = [x.op for x in kasm.linear(fixed) if x.cls not in ("label", "keep")]
    assert "s_nop" in ops  # nothing independent to put there


def test_verifier_reports_a_load_consumed_before_its_wait():
    A = kasm.Asm()
    off, base = A.v("off", pinned=True), A.st("base", 2, pinned=True)
    d, r = A.vt("d", 4), A.v("r", pinned=True)
    A.block("A", schedule=False)
    A.global_load(4, d, off, base)
    A.v_mov_b32(r, d[0])  # no s_waitcnt in between
    A.keep(r)
    for b in A.blocks:
        b.schedule = False
    kasm.allocate(A, list(range(0, 32)), list(range(0, 32)))
    probs = kasm.verify(A)
    assert any("waitcnt" in p for p in probs), probs


def test_allocator_aligns_tuples_and_never_overlaps_live_ranges():
    """64-bit and wider VGPR operands must start on an even register (gfx90a+); two values whose live ranges overlap never
    share a register; the order is deterministic (the generated header must not depend on object addresses)."""
    def build():
        A = kasm.Asm()
        ins = [A.v(f"i{k}", pinned=True) for k in range(3)]
        A.block("A")
        pairs = [A.vt(f"p{k}", 2) for k in range(6)]
        singles = [A.v(f"s{k}") for k in range(5)]
        for k, s in enumerate(singles):
            A.v_add_u32(s, ins[k % 3], ins[(k + 1) % 3])
        for k, p in enumerate(pairs):
            A.v_mad_u64_u32(p, kfield.DUMMY, singles[k % 5], ins[k % 3], 0 if k == 0 else pairs[k - 1])
        A.keep(*pairs[-1].regs, *singles)
        kasm.schedule(A)
        used = kasm.allocate(A, list(range(1, 40)), list(range(0, 16)))  # pool starts on an odd register on purpose
        return A, pairs, singles, used

    A, pairs, singles, used = build()
    assert all(p.regs[0].phys % 2 == 0 and p.regs[1].phys == p.regs[0].phys + 1 for p in pairs)
    # live ranges from the final text: a register may be shared only by values that are never live together
    prog = [x for x in kasm.linear(A) if x.cls not in ("label",)]
    first, last = {}, {}
    for pos, x in enumerate(prog):
        for r in x.defs + x.uses:
            if r.kind == "v" and not r.pinned:
                first.setdefault(r, pos)
                last[r] = pos
    regs = list(first)
    for i, a in enumerate(regs):
        for b in regs[i + 1:]:
            if a.phys == b.phys:
                assert last[a] <= first[b] or last[b] <= first[a], (a, b)
    A2, pairs2, singles2, _ = build()
    assert [p.regs[0].phys for p in pairs] == [p.regs[0].phys for p in pairs2]
    assert kasm.listing(A, comments=False) == kasm.listing(A2, comments=False)


def test_scheduler_fills_carry_distances_with_independent_work_and_honours_asap():
    """three independent carry chains: the list scheduler interleaves them instead of padding each link with s_nop; an
    instruction marked asap issues as soon as its operands exist, whatever the critical path says"""
    A = kasm.Asm()
    a = [A.v(f"a{i}", pinned=True) for i in range(6)]
    b = [A.v(f"b{i}", pinned=True) for i in range(6)]
    A.block("A")
    outs = []
    for chain, (x, y) in enumerate(((a, b), (b, a), (a, a))):
        c = A.st(f"c{chain}", 2)
        o = [A.v(f"o{chain}_{i}") for i in range(6)]
        A.v_add_co_u32(o[0], c, x[0], y[0])
        for i in range(1, 6):
            A.v_addc_co_u32(o[i], c, x[i], y[i], c)
        outs += o
    t = A.v("t")
    early = A.v_add_u32(t, a[0], b[0])
    early.asap = True
    A.keep(*outs, t)
    kasm.schedule(A)
    kasm.allocate(A, list(range(0, 64)), list(range(0, 16)))
    assert kasm.verify(A) == []
    prog = [x for x in kasm.linear(A) if x.cls not in ("label", "keep")]
    st = kasm.stats(A)
    assert st.get("nop_states", 0) <= 1, st  # a lone chain needs 2 wait states per link (10 per chain); three fill each other's
    assert prog.index(early) == 0


def test_spills_of_the_headline_kernel_stay_outside_the_scheduled_loop(tmp_path):
    """kng_walk_share_kernel<8, true, true> sits at the two-waves-per-SIMD ceiling -- 256 VGPRs -- WITH
    spills (48 VGPRs, 112 bytes of scratch per lane) around the generated statement.  Checked on the compiler's own listing
    (hipcc -S, no GPU): every scratch_load / scratch_store of the kernel lies in the compiler's code (inversion tree, entry pass,
    exact path), none between the ASMSTART / ASMEND of the per-kangaroo loop; LDS = jump table + exchange buffer; the ceiling
    kernel of bench.py (asm 2) has the same register / LDS footprint, or its rate would not be this kernel's ALU ceiling."""
    import shutil

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_stats

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    s = isa_stats.compile_s(os.path.join(ROOT, "kangaroo_amd", "csrc", "kng_engine.hip"), [])
    try:
        res = isa_stats.resources(s)
    finally:
        os.unlink(s)
    head = res["_Z21kng_walk_share_kernelILi8ELb1ELb1EEv8WalkArgs"]
    assert head["vgprs"] == 256 and head["asm_statements_over_500_lines"] == 1
    assert head["scratch_ops_in_asm_statements"] == 0
    assert 0 < head["scratch_ops_outside"] <= 64 and head["scratch_bytes_per_lane"] <= 128 and head["vgpr_spill_count"] <= 56
    assert head["lds_bytes"] < 20 * 1024
    ceil = res["_Z25kng_walk_valu_only_kernel8WalkArgs"]
    assert (ceil["vgprs"], ceil["lds_bytes"], ceil["scratch_bytes_per_lane"]) == (head["vgprs"], head["lds_bytes"], head["scratch_bytes_per_lane"])
    for name, r in res.items():
        assert r["scratch_ops_in_asm_statements"] == 0, name
