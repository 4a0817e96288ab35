#!/usr/bin/env python3
"""Drive the generated walk loop (tools/gen_walk_asm.py) in the emulator (tools/kasm_emu.py) against a plain-integer
model of walk_body's data flow (kng_engine.hip; GPUCompute.h:52-105 of the reference).  Test infrastructure: used by
tests/test_kasm_cpu.py and runnable by hand (`python tools/kwalk_emu.py`)."""
from __future__ import annotations

import os
import random
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_walk_asm  # noqa: E402
import kasm  # noqa: E402
import kfield  # noqa: E402
from kasm_emu import Emu, Memory  # noqa: E402

P = kfield.P
M256 = (1 << 256) - 1
M128 = (1 << 128) - 1

# operand binding used in emulation (the generator's pools start above these)
BIND = {f"%{i}": f"v{i}" for i in range(8)}
BIND.update({f"%{8 + i}": f"v{8 + i}" for i in range(8)})
BIND.update({"%16": "s0", "%17": "v16", "%18": "s[2:3]", "%19": "s4", "%20": "s5", "%21": "s6"})


def bind(lines):
    import re

    out = []
    for t in lines:
        t = re.sub(r"%(\d+)", lambda m: BIND["%" + m.group(1)], t)
        out.append(t.replace("%=", "0"))
    return out


class Model:
    """the herd and one launch of the walk, in integers.  State arrays indexed by kangaroo index."""

    def __init__(self, L, G, jx, jy, jd, dp_mask, seed=1, dsplit=True):
        rnd = random.Random(seed)
        self.L, self.G, self.N = L, G, L * G
        self.jx, self.jy, self.jd, self.dp_mask = jx, jy, jd, dp_mask
        self.x = [rnd.getrandbits(256) % P for _ in range(self.N)]
        self.y = [rnd.getrandbits(256) % P for _ in range(self.N)]
        self.d = [rnd.getrandbits(100) for _ in range(self.N)]
        self.s = [0] * self.N
        self.dps = []

    def slot(self, k, backward):
        return (self.G - 1 - k) if backward else k

    def pass0(self):
        acc = [None] * self.L
        for t in range(self.L):
            a = None
            for g in range(self.G):
                idx = g * self.L + t
                dx = kfield.ref_sub(self.x[idx], self.jx[self.x[idx] & 31])
                a = kfield.ref_mul(a, dx) if g else dx
                self.s[idx] = a
            acc[t] = a
        return acc

    def iteration(self, t, k, backward, last, inv, acc):
        """one kangaroo of lane t; returns (inv', acc')"""
        G, L = self.G, self.L
        idx = self.slot(k, backward) * L + t
        cx, cy = self.x[idx], self.y[idx]
        j = cx & 31
        dx = kfield.ref_sub(cx, self.jx[j])
        if k + 1 < G:
            nb = self.s[self.slot(k + 1, backward) * L + t]
            invk = kfield.ref_mul(inv, nb)
            inv = kfield.ref_mul(inv, dx)
        else:
            invk = inv
        dy = kfield.ref_sub(cy, self.jy[j])
        s = kfield.ref_mul(dy, invk)
        p2 = kfield.ref_mul(s, s)
        rx = kfield.ref_sub(kfield.ref_sub(p2, self.jx[j]), cx)
        ry = kfield.ref_sub(kfield.ref_mul(kfield.ref_sub(cx, rx), s), cy)
        d = (self.d[idx] + self.jd[j]) & M128
        self.x[idx], self.y[idx], self.d[idx] = rx, ry, d
        if ((rx >> 192) & self.dp_mask) == 0:
            self.dps.append((idx, rx, d))
        if not last:
            dx2 = kfield.ref_sub(rx, self.jx[rx & 31])
            acc = kfield.ref_mul(acc, dx2) if k else dx2
            self.s[idx] = acc
        return inv, acc


def fe_bytes(v, lo):
    return struct.pack("<2Q", (v >> (128 if not lo else 0)) & ((1 << 64) - 1), (v >> (192 if not lo else 64)) & ((1 << 64) - 1))


class Harness:
    def __init__(self, model: Model, dsplit: bool, max_found=4096):
        self.m, self.dsplit = model, dsplit
        lp, used, probs = gen_walk_asm.generate(dsplit)
        assert not probs, probs
        assert min(used["v"]) > 16 and min(used["s"]) > 6
        self.text = bind(kasm.listing(lp.A, comments=False))
        self.used = used
        mem = self.mem = Memory()
        N = model.N
        self.planes = {n: mem.alloc(16 * N) for n in ("x01", "x23", "y01", "y23", "s01", "s23")}
        self.planes["dlo"] = mem.alloc(8 * N)
        self.planes["dhi"] = mem.alloc(8 * N)
        self.dp_count = mem.alloc(64)
        self.dp_items = mem.alloc(64 * max_found)
        self.max_found = max_found
        self.args = mem.alloc(96)
        blk = struct.pack("<8Q", *[self.planes[n] for n in ("x01", "x23", "y01", "y23", "dlo", "dhi", "s01", "s23")])
        blk += struct.pack("<3Q2I", model.dp_mask, self.dp_count, self.dp_items, max_found, 0)
        mem.write(self.args, blk)
        self.lds_tab = 0x2000
        self.exact_dps = []
        self.upload()

    def upload(self):
        m, mem = self.m, self.mem
        for i in range(m.N):
            mem.write(self.planes["x01"] + 16 * i, fe_bytes(m.x[i], True))
            mem.write(self.planes["x23"] + 16 * i, fe_bytes(m.x[i], False))
            mem.write(self.planes["y01"] + 16 * i, fe_bytes(m.y[i], True))
            mem.write(self.planes["y23"] + 16 * i, fe_bytes(m.y[i], False))
            mem.write(self.planes["s01"] + 16 * i, fe_bytes(m.s[i], True))
            mem.write(self.planes["s23"] + 16 * i, fe_bytes(m.s[i], False))
            mem.write(self.planes["dlo"] + 8 * i, struct.pack("<Q", m.d[i] & ((1 << 64) - 1)))
            mem.write(self.planes["dhi"] + 8 * i, struct.pack("<Q", m.d[i] >> 64))

    def download(self):
        m, mem = self.m, self.mem
        rd = lambda n, i, w: int.from_bytes(mem.read(self.planes[n] + w * i, w), "little")  # noqa: E731
        x = [rd("x01", i, 16) | (rd("x23", i, 16) << 128) for i in range(m.N)]
        y = [rd("y01", i, 16) | (rd("y23", i, 16) << 128) for i in range(m.N)]
        s = [rd("s01", i, 16) | (rd("s23", i, 16) << 128) for i in range(m.N)]
        d = [rd("dlo", i, 8) | (rd("dhi", i, 8) << 64) for i in range(m.N)]
        return x, y, d, s

    def new_emu(self):
        e = Emu(lanes=self.m.L, mem=self.mem)
        # limb-major LDS table: jx[4][32] jy[4][32] jd[2][32] 64-bit words
        m = self.m
        for j in range(32):
            for k in range(4):
                struct.pack_into("<Q", e.lds, self.lds_tab + (k * 32 + j) * 8, (m.jx[j] >> (64 * k)) & ((1 << 64) - 1))
                struct.pack_into("<Q", e.lds, self.lds_tab + ((4 + k) * 32 + j) * 8, (m.jy[j] >> (64 * k)) & ((1 << 64) - 1))
            struct.pack_into("<Q", e.lds, self.lds_tab + (8 * 32 + j) * 8, m.jd[j] & ((1 << 64) - 1))
            struct.pack_into("<Q", e.lds, self.lds_tab + (9 * 32 + j) * 8, m.jd[j] >> 64)
        return e

    def run_step(self, inv, acc, backward, stats):
        """one jump of every kangaroo through the asm loop (not the last step of a launch); inv, acc: per-lane ints.
        Returns the new acc list.  The model advances in lock step (it IS the expectation)."""
        m = self.m
        L, G = m.L, m.G
        e = self.new_emu()
        k = 0
        acc = [1] * L  # acc' = acc * dx2 with acc = 1 for the first kangaroo (exact: no reduction happens)
        model_inv, model_acc = list(inv), [None] * L
        while k < G:
            for t in range(L):
                for i in range(8):
                    e.v[i][t] = (inv[t] >> (32 * i)) & 0xFFFFFFFF
                    e.v[8 + i][t] = (acc[t] >> (32 * i)) & 0xFFFFFFFF
                e.v[16][t] = ((m.slot(k, backward) * L + t) * 16) & 0xFFFFFFFF
            e.s[0] = k
            e.s[2], e.s[3] = self.args & 0xFFFFFFFF, self.args >> 32
            e.s[4] = ((-L * 16) if backward else (L * 16)) & 0xFFFFFFFF
            e.s[5] = G
            e.s[6] = self.lds_tab
            e.exec = (1 << L) - 1
            e.run(self.text)
            k_out = e.s[0]
            assert k <= k_out <= G, (k, k_out)
            # the model follows: iterations k .. k_out-1 were completed by the asm
            for kk in range(k, k_out):
                for t in range(L):
                    model_inv[t], model_acc[t] = m.iteration(t, kk, backward, False, model_inv[t], model_acc[t])
            inv = [sum(e.v[i][t] << (32 * i) for i in range(8)) for t in range(L)]
            acc = [sum(e.v[8 + i][t] << (32 * i) for i in range(8)) for t in range(L)]
            if k_out > k:
                if k_out < G:  # the running inverse is dead behind the last kangaroo
                    assert inv == model_inv, f"running inverse differs after iteration {k_out - 1}"
                else:  # (after an exact-path exit the product operand is undefined: the caller re-reads it from memory)
                    assert acc == model_acc, f"running product differs after iteration {k_out - 1}"
            k = k_out
            if k < G:
                # exact-path exit: nothing of iteration k may have been stored; do it in the model and resume behind it
                stats["rare_exits"] = stats.get("rare_exits", 0) + 1
                self.check_memory(f"at the exact-path exit of iteration {k}")
                n_before = len(m.dps)
                for t in range(L):
                    model_inv[t], model_acc[t] = m.iteration(t, k, backward, False, model_inv[t], model_acc[t])
                self.exact_dps += m.dps[n_before:]  # the C++ exact path emits these, not the asm loop
                self.upload()
                inv, acc = list(model_inv), list(model_acc)
                k += 1
        stats["instructions"] = stats.get("instructions", 0) + e.count
        self.last_hist = dict(e.hist)
        return model_acc

    def check_memory(self, what):
        x, y, d, s = self.download()
        m = self.m
        for name, got, want in (("x", x, m.x), ("y", y, m.y), ("s", s, m.s)):
            bad = [i for i in range(m.N) if got[i] != want[i]]
            assert not bad, f"{name} differs {what}: kangaroos {bad[:8]}"
        bad = [i for i in range(m.N) if d[i] != m.d[i]]  # DSPLIT: high words only change on the exact path (model + upload)
        assert not bad, f"d differs {what}: kangaroos {bad[:8]}"

    def dp_records(self):
        n = struct.unpack("<I", self.mem.read(self.dp_count, 4))[0]
        recs = []
        for i in range(min(n, self.max_found)):
            b = self.mem.read(self.dp_items + 64 * i, 64)
            q = struct.unpack("<8Q", b)
            recs.append((q[6], q[0] | (q[1] << 64) | (q[2] << 128) | (q[3] << 192), q[4] | (q[5] << 64)))
        return n, recs


def random_table(seed, jd_bits=60):
    rnd = random.Random(seed)
    return ([rnd.getrandbits(256) % P for _ in range(32)], [rnd.getrandbits(256) % P for _ in range(32)],
            [rnd.getrandbits(jd_bits) | 1 for _ in range(32)])


def run_case(L=8, G=5, steps=3, dsplit=True, dp_bits=3, jd_bits=60, seed=1, verbose=True):
    jx, jy, jd = random_table(seed + 100, jd_bits)
    dp_mask = ((1 << dp_bits) - 1) << (64 - dp_bits) if dp_bits else 0
    m = Model(L, G, jx, jy, jd, dp_mask, seed=seed)
    acc = m.pass0()
    h = Harness(m, dsplit)
    stats = {}
    for step in range(steps):
        inv = [pow(a % P, P - 2, P) for a in acc]
        acc = h.run_step(inv, acc, backward=(step % 2 == 0), stats=stats)
        h.check_memory(f"after step {step}")
    n, recs = h.dp_records()
    want = sorted(m.dps)
    for r in h.exact_dps:
        want.remove(r)
    assert n == len(want), f"DP count {n} vs {len(want)}"
    assert sorted(recs) == want, "DP records differ"
    if verbose:
        print(f"L={L} G={G} steps={steps} dsplit={dsplit} jd_bits={jd_bits}: ok, {n} DPs, {stats}")
    return stats


if __name__ == "__main__":
    run_case(L=8, G=5, steps=3, dsplit=True)
    run_case(L=8, G=5, steps=2, dsplit=False, jd_bits=100)
    run_case(L=8, G=1, steps=2, dsplit=True)
    run_case(L=8, G=2, steps=2, dsplit=False, jd_bits=90)
    run_case(L=8, G=4, steps=2, dsplit=True, jd_bits=64, seed=3)  # carries out of the low word: exact-path exits
    run_case(L=64, G=3, steps=2, dsplit=True, dp_bits=2, seed=5)
