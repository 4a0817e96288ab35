#!/usr/bin/env python3
"""kasm_emu -- executes the gfx950 subset that tools/kasm.py emits, lane by lane, on the CPU.

Test infrastructure for the generated walk loop: the printed text (physical registers, final order) is parsed
again and run on W <= 64 lanes with real EXEC/VCC/SCC semantics, a flat byte-addressed global memory and an LDS
array.  It validates the generator's arithmetic, scheduling order and register allocation; it does NOT model
timing (hazards and s_waitcnt coverage are checked statically by kasm.verify).
"""
from __future__ import annotations

import re
import struct

M32 = 0xFFFFFFFF
M64 = 0xFFFFFFFFFFFFFFFF


class Memory:
    """sparse byte-addressed memory made of registered buffers (base address -> bytearray)"""

    def __init__(self):
        self.bufs = []  # (base, bytearray)
        self.next = 0x7F0000000000

    def alloc(self, data_or_size):
        buf = bytearray(data_or_size) if not isinstance(data_or_size, int) else bytearray(data_or_size)
        base = self.next
        self.next += (len(buf) + 0xFFF) & ~0xFFF
        self.next += 0x1000
        self.bufs.append((base, buf))
        return base

    def _find(self, addr, n):
        for base, buf in self.bufs:
            if base <= addr and addr + n <= base + len(buf):
                return buf, addr - base
        raise MemoryError(f"access of {n} bytes at {addr:#x} outside every buffer")

    def read(self, addr, n):
        buf, o = self._find(addr, n)
        return bytes(buf[o:o + n])

    def write(self, addr, data):
        buf, o = self._find(addr, len(data))
        buf[o:o + len(data)] = data

    def buffer(self, base):
        for b, buf in self.bufs:
            if b == base:
                return buf
        raise KeyError(base)


class Emu:
    def __init__(self, lanes=64, mem=None, lds_bytes=65536):
        self.W = lanes
        self.v = [[0] * lanes for _ in range(512)]
        self.s = [0] * 128
        self.vcc = 0
        self.exec = (1 << lanes) - 1
        self.scc = 0
        self.mem = mem or Memory()
        self.lds = bytearray(lds_bytes)
        self.count = 0
        self.hist = {}

    # ---- operand access
    def sget(self, name):
        if name == "vcc_lo":
            return self.vcc & M32
        if name == "vcc_hi":
            return (self.vcc >> 32) & M32
        if name == "exec_lo":
            return self.exec & M32
        if name == "exec_hi":
            return (self.exec >> 32) & M32
        return self.s[int(name[1:])]

    def s64(self, tok):
        if tok == "vcc":
            return self.vcc
        if tok == "exec":
            return self.exec
        m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
        if m:
            a = int(m.group(1))
            return self.s[a] | (self.s[a + 1] << 32)
        return self.imm(tok) & M64

    def set_s64(self, tok, val):
        val &= M64
        if tok == "vcc":
            self.vcc = val
        elif tok == "exec":
            self.exec = val & ((1 << self.W) - 1)
        else:
            m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
            a = int(m.group(1))
            assert int(m.group(2)) == a + 1 and a % 2 == 0, f"bad SGPR pair {tok}"
            self.s[a], self.s[a + 1] = val & M32, val >> 32

    def s32(self, tok):
        if re.fullmatch(r"s\d+", tok) or tok in ("vcc_lo", "vcc_hi", "exec_lo", "exec_hi"):
            return self.sget(tok)
        return self.imm(tok) & M32

    def set_s32(self, tok, val):
        if tok == "vcc_lo":
            self.vcc = (self.vcc & ~M32) | (val & M32)
        elif tok == "vcc_hi":
            self.vcc = (self.vcc & M32) | ((val & M32) << 32)
        else:
            self.s[int(tok[1:])] = val & M32

    @staticmethod
    def imm(tok):
        return int(tok, 0)

    def vsrc32(self, tok, lane):
        if tok[0] == "v" and tok[1:].isdigit():
            return self.v[int(tok[1:])][lane]
        return self.s32(tok)

    def vsrc64(self, tok, lane):
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        if m:
            a = int(m.group(1))
            assert a % 2 == 0 and int(m.group(2)) == a + 1, f"misaligned 64-bit VGPR operand {tok}"
            return self.v[a][lane] | (self.v[a + 1][lane] << 32)
        return self.s64(tok)

    @staticmethod
    def vrange(tok):
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        if m:
            a, b = int(m.group(1)), int(m.group(2))
            assert a % 2 == 0, f"misaligned VGPR tuple {tok}"
            return list(range(a, b + 1))
        assert tok[0] == "v", tok
        return [int(tok[1:])]

    def lanes(self):
        return [l for l in range(self.W) if (self.exec >> l) & 1]

    # ---- execution
    def run(self, lines, max_steps=10**8):
        prog, labels = [], {}
        for raw in lines:
            t = raw.split(";")[0].strip()
            if not t:
                continue
            if t.endswith(":"):
                labels[t[:-1]] = len(prog)
                continue
            prog.append(t)
        pc, steps = 0, 0
        while pc < len(prog):
            steps += 1
            if steps > max_steps:
                raise RuntimeError("emulation step limit")
            t = prog[pc]
            pc += 1
            op, _, rest = t.partition(" ")
            mods = {}
            flags = set()
            toks = []
            # split operands from modifiers (modifiers are space-separated words after the last operand)
            parts = [p.strip() for p in rest.split(",")] if rest.strip() else []
            if parts:
                lastw = parts[-1].split()
                parts[-1] = lastw[0] if lastw else ""
                for w in lastw[1:]:
                    if ":" in w:
                        k, val = w.split(":")
                        mods[k] = int(val, 0)
                    else:
                        flags.add(w)
            toks = [p for p in parts if p != ""]
            self.count += 1
            self.hist[op] = self.hist.get(op, 0) + 1
            tgt = self.step(op, toks, mods, flags, t)
            if tgt is not None:
                if tgt == "__end__":
                    return
                pc = labels[tgt]

    def step(self, op, a, mods, flags, text):
        op0 = re.sub(r"_e(32|64)$", "", op)
        f = getattr(self, "op_" + op0, None)
        if f is None:
            raise NotImplementedError(f"emulator: {text}")
        return f(a, mods, flags)

    # ---- VALU
    def op_v_mad_u64_u32(self, a, m, f):
        d, co, x, y, c = a
        dr = self.vrange(d)
        assert len(dr) == 2
        carry = 0
        res = {}
        for l in self.lanes():
            r = self.vsrc32(x, l) * self.vsrc32(y, l) + self.vsrc64(c, l)
            if r >> 64:
                carry |= 1 << l
            res[l] = r & M64
        for l, r in res.items():
            self.v[dr[0]][l], self.v[dr[1]][l] = r & M32, r >> 32
        self.set_s64(co, carry)  # inactive lanes read as 0

    def _addsub(self, a, sub, with_cin, rev=False):
        d, co, x, y = a[:4]
        cin = self.s64(a[4]) if with_cin else 0
        carry = 0
        res = {}
        for l in self.lanes():
            xv, yv = self.vsrc32(x, l), self.vsrc32(y, l)
            if rev:
                xv, yv = yv, xv
            c = (cin >> l) & 1
            r = xv - yv - c if sub else xv + yv + c
            if (r < 0) if sub else (r >> 32):
                carry |= 1 << l
            res[l] = r & M32
        dd = int(d[1:])
        for l, r in res.items():
            self.v[dd][l] = r
        self.set_s64(co, carry)

    def op_v_add_co_u32(self, a, m, f):
        self._addsub(a, False, False)

    def op_v_addc_co_u32(self, a, m, f):
        self._addsub(a, False, True)

    def op_v_sub_co_u32(self, a, m, f):
        self._addsub(a, True, False)

    def op_v_subb_co_u32(self, a, m, f):
        self._addsub(a, True, True)

    def _v(self, a, fn):
        d = int(a[0][1:])
        res = {l: fn(*[self.vsrc32(t, l) for t in a[1:]]) & M32 for l in self.lanes()}
        for l, r in res.items():
            self.v[d][l] = r

    def op_v_mov_b32(self, a, m, f):
        self._v(a, lambda x: x)

    def op_v_and_b32(self, a, m, f):
        self._v(a, lambda x, y: x & y)

    def op_v_or_b32(self, a, m, f):
        self._v(a, lambda x, y: x | y)

    def op_v_xor_b32(self, a, m, f):
        self._v(a, lambda x, y: x ^ y)

    def op_v_add_u32(self, a, m, f):
        self._v(a, lambda x, y: x + y)

    def op_v_sub_u32(self, a, m, f):
        self._v(a, lambda x, y: x - y)

    def op_v_lshlrev_b32(self, a, m, f):
        self._v(a, lambda sh, x: x << (sh & 31))

    def op_v_lshrrev_b32(self, a, m, f):
        self._v(a, lambda sh, x: x >> (sh & 31))

    def op_v_alignbit_b32(self, a, m, f):
        self._v(a, lambda hi, lo, sh: ((hi << 32) | lo) >> (sh & 31))

    def op_v_and_or_b32(self, a, m, f):
        self._v(a, lambda x, y, z: (x & y) | z)

    def op_v_max3_u32(self, a, m, f):
        self._v(a, lambda x, y, z: max(x, y, z))

    def op_v_min3_u32(self, a, m, f):
        self._v(a, lambda x, y, z: min(x, y, z))

    def op_v_lshl_add_u32(self, a, m, f):
        self._v(a, lambda x, sh, y: (x << (sh & 31)) + y)

    def op_v_mbcnt_lo_u32_b32(self, a, m, f):
        d = int(a[0][1:])
        res = {l: (bin(self.s32(a[1]) & ((1 << min(l, 32)) - 1)).count("1") + self.vsrc32(a[2], l)) & M32 for l in self.lanes()}
        for l, r in res.items():
            self.v[d][l] = r

    def op_v_mbcnt_hi_u32_b32(self, a, m, f):
        d = int(a[0][1:])
        res = {l: (bin(self.s32(a[1]) & ((1 << max(l - 32, 0)) - 1)).count("1") + self.vsrc32(a[2], l)) & M32 for l in self.lanes()}
        for l, r in res.items():
            self.v[d][l] = r

    def _vcmp(self, a, fn):
        mask = 0
        for l in self.lanes():
            if fn(self.vsrc32(a[1], l), self.vsrc32(a[2], l)):
                mask |= 1 << l
        self.set_s64(a[0], mask)

    def op_v_cmp_eq_u32(self, a, m, f):
        self._vcmp(a, lambda x, y: x == y)

    def op_v_cmp_ne_u32(self, a, m, f):
        self._vcmp(a, lambda x, y: x != y)

    def op_v_cmp_lt_u32(self, a, m, f):
        self._vcmp(a, lambda x, y: x < y)

    def op_v_cmp_le_u32(self, a, m, f):
        self._vcmp(a, lambda x, y: x <= y)

    def op_v_cmp_gt_u32(self, a, m, f):
        self._vcmp(a, lambda x, y: x > y)

    def op_v_cmp_ge_u32(self, a, m, f):
        self._vcmp(a, lambda x, y: x >= y)

    def op_v_cndmask_b32(self, a, m, f):
        d = int(a[0][1:])
        mask = self.s64(a[3])
        res = {l: (self.vsrc32(a[2], l) if (mask >> l) & 1 else self.vsrc32(a[1], l)) for l in self.lanes()}
        for l, r in res.items():
            self.v[d][l] = r

    def op_v_readfirstlane_b32(self, a, m, f):
        ls = self.lanes()
        l = ls[0] if ls else 0
        self.set_s32(a[0], self.vsrc32(a[1], l))

    # ---- SALU
    def op_s_mov_b32(self, a, m, f):
        self.set_s32(a[0], self.s32(a[1]))

    def op_s_mov_b64(self, a, m, f):
        self.set_s64(a[0], self.s64(a[1]))

    def _s64op(self, a, fn):
        r = fn(self.s64(a[1]), self.s64(a[2])) & M64
        self.set_s64(a[0], r)
        self.scc = int(r != 0)

    def op_s_or_b64(self, a, m, f):
        self._s64op(a, lambda x, y: x | y)

    def op_s_and_b64(self, a, m, f):
        self._s64op(a, lambda x, y: x & y)

    def op_s_xor_b64(self, a, m, f):
        self._s64op(a, lambda x, y: x ^ y)

    def op_s_andn2_b64(self, a, m, f):
        self._s64op(a, lambda x, y: x & ~y)

    def op_s_and_b32(self, a, m, f):
        r = self.s32(a[1]) & self.s32(a[2])
        self.set_s32(a[0], r)
        self.scc = int(r != 0)

    def op_s_or_b32(self, a, m, f):
        r = self.s32(a[1]) | self.s32(a[2])
        self.set_s32(a[0], r)
        self.scc = int(r != 0)

    def op_s_add_u32(self, a, m, f):
        r = self.s32(a[1]) + self.s32(a[2])
        self.set_s32(a[0], r)
        self.scc = r >> 32

    def op_s_addc_u32(self, a, m, f):
        r = self.s32(a[1]) + self.s32(a[2]) + self.scc
        self.set_s32(a[0], r)
        self.scc = r >> 32

    def op_s_sub_u32(self, a, m, f):
        r = self.s32(a[1]) - self.s32(a[2])
        self.set_s32(a[0], r)
        self.scc = int(r < 0)

    def op_s_subb_u32(self, a, m, f):
        r = self.s32(a[1]) - self.s32(a[2]) - self.scc
        self.set_s32(a[0], r)
        self.scc = int(r < 0)

    @staticmethod
    def _i32(x):
        return x - (1 << 32) if x & 0x80000000 else x

    def op_s_add_i32(self, a, m, f):
        x, y = self._i32(self.s32(a[1])), self._i32(self.s32(a[2]))
        r = x + y
        self.set_s32(a[0], r)
        self.scc = int(not (-(1 << 31) <= r < (1 << 31)))

    def op_s_sub_i32(self, a, m, f):
        x, y = self._i32(self.s32(a[1])), self._i32(self.s32(a[2]))
        r = x - y
        self.set_s32(a[0], r)
        self.scc = int(not (-(1 << 31) <= r < (1 << 31)))

    def op_s_mul_i32(self, a, m, f):
        self.set_s32(a[0], self.s32(a[1]) * self.s32(a[2]))

    def op_s_lshl_b32(self, a, m, f):
        r = (self.s32(a[1]) << (self.s32(a[2]) & 31)) & M32
        self.set_s32(a[0], r)
        self.scc = int(r != 0)

    def op_s_lshr_b32(self, a, m, f):
        r = self.s32(a[1]) >> (self.s32(a[2]) & 31)
        self.set_s32(a[0], r)
        self.scc = int(r != 0)

    def op_s_lshl_b64(self, a, m, f):
        r = (self.s64(a[1]) << (self.s32(a[2]) & 63)) & M64
        self.set_s64(a[0], r)
        self.scc = int(r != 0)

    def op_s_bcnt1_i32_b64(self, a, m, f):
        r = bin(self.s64(a[1])).count("1")
        self.set_s32(a[0], r)
        self.scc = int(r != 0)

    def op_s_ff1_i32_b64(self, a, m, f):
        x = self.s64(a[1])
        self.set_s32(a[0], (x & -x).bit_length() - 1 if x else M32)

    def op_s_cselect_b32(self, a, m, f):
        self.set_s32(a[0], self.s32(a[1]) if self.scc else self.s32(a[2]))

    def op_s_cselect_b64(self, a, m, f):
        self.set_s64(a[0], self.s64(a[1]) if self.scc else self.s64(a[2]))

    def _scmp(self, a, fn, wide=False):
        g = self.s64 if wide else self.s32
        self.scc = int(fn(g(a[0]), g(a[1])))

    def op_s_cmp_lg_u64(self, a, m, f):
        self._scmp(a, lambda x, y: x != y, True)

    def op_s_cmp_eq_u64(self, a, m, f):
        self._scmp(a, lambda x, y: x == y, True)

    def op_s_cmp_eq_u32(self, a, m, f):
        self._scmp(a, lambda x, y: x == y)

    def op_s_cmp_lg_u32(self, a, m, f):
        self._scmp(a, lambda x, y: x != y)

    def op_s_cmp_lt_u32(self, a, m, f):
        self._scmp(a, lambda x, y: x < y)

    def op_s_cmp_le_u32(self, a, m, f):
        self._scmp(a, lambda x, y: x <= y)

    def op_s_cmp_gt_u32(self, a, m, f):
        self._scmp(a, lambda x, y: x > y)

    def op_s_cmp_ge_u32(self, a, m, f):
        self._scmp(a, lambda x, y: x >= y)

    def op_s_and_saveexec_b64(self, a, m, f):
        old = self.exec
        self.exec = old & self.s64(a[1]) & ((1 << self.W) - 1)
        self.set_s64(a[0], old)
        self.scc = int(self.exec != 0)

    def op_s_nop(self, a, m, f):
        pass

    def op_s_waitcnt(self, a, m, f):
        pass

    def op_s_endpgm(self, a, m, f):
        return "__end__"

    def op_s_branch(self, a, m, f):
        return a[0]

    def op_s_cbranch_scc1(self, a, m, f):
        return a[0] if self.scc else None

    def op_s_cbranch_scc0(self, a, m, f):
        return a[0] if not self.scc else None

    def op_s_cbranch_execz(self, a, m, f):
        return a[0] if self.exec == 0 else None

    def op_s_cbranch_execnz(self, a, m, f):
        return a[0] if self.exec != 0 else None

    def op_s_cbranch_vccz(self, a, m, f):
        return a[0] if self.vcc == 0 else None

    def op_s_cbranch_vccnz(self, a, m, f):
        return a[0] if self.vcc != 0 else None

    # ---- memory
    def _gaddr(self, voff, sbase, mods, lane):
        if sbase == "off":
            return self.vsrc64(voff, lane) + mods.get("offset", 0)
        return self.s64(sbase) + self.vsrc32(voff, lane) + mods.get("offset", 0)

    def _gload(self, a, m, n):
        regs = self.vrange(a[0])
        assert len(regs) == n, f"destination of a {n}-dword load: {a[0]}"
        res = {}
        for l in self.lanes():
            res[l] = struct.unpack(f"<{n}I", self.mem.read(self._gaddr(a[1], a[2], m, l), 4 * n))
        for l, vals in res.items():
            for r, x in zip(regs, vals):
                self.v[r][l] = x

    def _gstore(self, a, m, n):
        regs = self.vrange(a[1])
        assert len(regs) == n, f"data of a {n}-dword store: {a[1]}"
        for l in self.lanes():
            self.mem.write(self._gaddr(a[0], a[2], m, l), struct.pack(f"<{n}I", *[self.v[r][l] for r in regs]))

    def op_global_load_dword(self, a, m, f):
        self._gload(a, m, 1)

    def op_global_load_dwordx2(self, a, m, f):
        self._gload(a, m, 2)

    def op_global_load_dwordx4(self, a, m, f):
        self._gload(a, m, 4)

    def op_global_store_dword(self, a, m, f):
        self._gstore(a, m, 1)

    def op_global_store_dwordx2(self, a, m, f):
        self._gstore(a, m, 2)

    def op_global_store_dwordx4(self, a, m, f):
        self._gstore(a, m, 4)

    def op_global_atomic_add(self, a, m, f):
        # returning form: vdst, voff, vdata, saddr sc0
        assert "sc0" in f
        d = int(a[0][1:])
        for l in self.lanes():
            addr = self._gaddr(a[1], a[3], m, l)
            old = struct.unpack("<I", self.mem.read(addr, 4))[0]
            self.mem.write(addr, struct.pack("<I", (old + self.vsrc32(a[2], l)) & M32))
            self.v[d][l] = old

    def op_global_atomic_add_x2(self, a, m, f):
        # non-returning form: voff, vdata(64), saddr
        assert "sc0" not in f
        regs = self.vrange(a[1])
        assert len(regs) == 2
        for l in self.lanes():
            addr = self._gaddr(a[0], a[2], m, l)
            old = struct.unpack("<Q", self.mem.read(addr, 8))[0]
            add = self.v[regs[0]][l] | (self.v[regs[1]][l] << 32)
            self.mem.write(addr, struct.pack("<Q", (old + add) & ((1 << 64) - 1)))

    def op_ds_read_b64(self, a, m, f):
        regs = self.vrange(a[0])
        assert len(regs) == 2
        res = {}
        for l in self.lanes():
            o = self.vsrc32(a[1], l) + m.get("offset", 0)
            assert o % 8 == 0
            res[l] = struct.unpack_from("<2I", self.lds, o)
        for l, vals in res.items():
            self.v[regs[0]][l], self.v[regs[1]][l] = vals

    def op_ds_read2_b64(self, a, m, f):
        regs = self.vrange(a[0])
        assert len(regs) == 4
        res = {}
        for l in self.lanes():
            base = self.vsrc32(a[1], l)
            assert base % 8 == 0
            o0, o1 = base + 8 * m.get("offset0", 0), base + 8 * m.get("offset1", 0)
            res[l] = struct.unpack_from("<2I", self.lds, o0) + struct.unpack_from("<2I", self.lds, o1)
        for l, vals in res.items():
            for r, x in zip(regs, vals):
                self.v[r][l] = x

    def _sload(self, a, n):
        m = re.fullmatch(r"s\[(\d+):(\d+)\]", a[0])
        base = int(m.group(1)) if m else int(a[0][1:])
        addr = self.s64(a[1]) + self.imm(a[2])
        vals = struct.unpack(f"<{n}I", self.mem.read(addr, 4 * n))
        for i, x in enumerate(vals):
            self.s[base + i] = x

    def op_s_load_dword(self, a, m, f):
        self._sload(a, 1)

    def op_s_load_dwordx2(self, a, m, f):
        self._sload(a, 2)

    def op_s_load_dwordx4(self, a, m, f):
        self._sload(a, 4)

    def op_s_load_dwordx8(self, a, m, f):
        self._sload(a, 8)

    def op_s_load_dwordx16(self, a, m, f):
        self._sload(a, 16)
