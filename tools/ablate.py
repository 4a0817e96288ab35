#!/usr/bin/env python3
"""Sensitivity map of the walk kernel: builds of the engine with ONE cost component removed.

The ablated builds compute WRONG results on purpose (a removed carry, a skipped inversion, a missing
store): they only answer "how much faster would the kernel be if this component were free?", i.e. the
ceiling of any optimisation of that component, before the optimisation is written.  Nothing here is
product code: the patched sources live under build/abl/<name>/ (git-ignored) and are only ever loaded
through KNG_LIB_PATH by tools/ablate_run.sh.

usage: python tools/ablate.py            # build every variant
       python tools/ablate.py no_s_traffic no_inversion
"""
import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kangaroo_amd", "csrc")
OUT = os.path.join(ROOT, "build", "abl")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"]


def sub1(text, old, new, count=None):
    n = text.count(old)
    assert n >= 1, f"pattern not found: {old[:60]!r}"
    if count is not None:
        assert n == count, f"expected {count} matches of {old[:60]!r}, found {n}"
    return text.replace(old, new)


def v_base(files):
    return files


def v_no_s_traffic(files):
    """prefix-product planes S01/S23 neither read nor written in the jump loop (pass 0 keeps them)"""
    t = files["kng_engine.hip"]
    body = t[t.index("template <int SHARE, bool DSPLIT>\nKNG_DEV void walk_body"):t.index("__global__ void __launch_bounds__(256) kng_walk_kernel")]
    nb = body
    nb = sub1(nb, "fe nb = (G > 1) ? ld_prod(a.s01, a.s23, slot(1)) : fe_one();", "fe nb = fe{{cx.v[1], cy.v[0], cx.v[3], cy.v[2]}};", 1)
    nb = sub1(nb, "if (k + 2 < G) nnb = ld_prod(a.s01, a.s23, slot(k + 2));", "if (k + 2 < G) nnb = fe{{nx.v[1], ny.v[0], nx.v[3], ny.v[2]}};", 1)
    nb = sub1(nb, "                acc = k ? fe_mul(acc, dx2) : dx2;\n                st_prod(a.s01, a.s23, idx, acc);", "                acc = k ? fe_mul(acc, dx2) : dx2;", 1)
    files["kng_engine.hip"] = t.replace(body, nb)
    return files


def v_no_inversion(files):
    t = files["kng_engine.hip"]
    t = sub1(t, "fe i = fe_inv(pre[SHARE - 1]);", "fe i = pre[SHARE - 1];", 1)
    files["kng_engine.hip"] = t
    return files


def v_no_comba_carry(files):
    """Comba columns without the v_addc that collects each MAD's carry-out (62 per product)"""
    t = files["kng_mul32.h"]
    t = re.sub(r"\\n\\tv_addc_co_u32 %1, %\d, 0, (0|%1), %\d", "", t)
    t = t.replace('"=&v"(hi)', '"+v"(hi)')
    t = t.replace("    uint32_t hi;\n", "    uint32_t hi = 0;\n")
    files["kng_mul32.h"] = t
    return files


def v_no_fold(files):
    """512 -> 256 reduction replaced by a xor of the halves (ceiling for a cheaper fold)"""
    t = files["kng_field.h"]
    t = sub1(t, "KNG_DEV fe fe_fold32(const uint32_t w[16], uint64_t &rare, unsigned &lane) {", "KNG_DEV fe fe_fold32(const uint32_t w[16], uint64_t &rare, unsigned &lane) {\n"
             "    lane = 0;\n"
             "    return fe{{(uint64_t)(w[0] ^ w[8]) | ((uint64_t)(w[1] ^ w[9]) << 32), (uint64_t)(w[2] ^ w[10]) | ((uint64_t)(w[3] ^ w[11]) << 32),\n"
             "               (uint64_t)(w[4] ^ w[12]) | ((uint64_t)(w[5] ^ w[13]) << 32), (uint64_t)(w[6] ^ w[14]) | ((uint64_t)(w[7] ^ w[15]) << 32)}};\n"
             "}\nKNG_DEV fe fe_fold32_unused(const uint32_t w[16], uint64_t &rare, unsigned &lane) {", 1)
    files["kng_field.h"] = t
    return files


def v_no_state_store(files):
    """x, y, d of the jumped kangaroo not written back"""
    t = files["kng_engine.hip"]
    t = sub1(t, "            st_fe(a.x01, a.x23, idx, rx);\n            st_fe(a.y01, a.y23, idx, ry);\n            st_stream64(dlo + idx, cd.x);\n            if (!DSPLIT) st_stream64(dhi + idx, cd.y);\n",
             "            asm volatile(\"\" ::\"v\"(ry.v[0]), \"v\"(ry.v[1]), \"v\"(ry.v[2]), \"v\"(ry.v[3]), \"v\"(cd.x));\n", 1)
    files["kng_engine.hip"] = t
    return files


def v_no_memory(files):
    files = v_no_s_traffic(files)
    files = v_no_state_store(files)
    t = files["kng_engine.hip"]
    body = t[t.index("template <int SHARE, bool DSPLIT>\nKNG_DEV void walk_body"):t.index("__global__ void __launch_bounds__(256) kng_walk_kernel")]
    nb = body
    nb = sub1(nb, "                nx = ld_fe(a.x01, a.x23, nidx);\n                ny = ld_fe(a.y01, a.y23, nidx);\n                nd = DSPLIT ? make_ulonglong2(ld_stream64(dlo + nidx), 0) : ld_d(a.d, a.n_kang, nidx);",
              "                nx = fe{{rxp.v[0] + nidx, rxp.v[1], rxp.v[2], rxp.v[3]}};\n                ny = fe{{rxp.v[1], rxp.v[2] ^ nidx, rxp.v[0], rxp.v[3]}};\n                nd = make_ulonglong2(nidx, 0);", 1)
    nb = sub1(nb, "        for (uint32_t k = 0; k < G; k++) {\n            // ---- prefetch", "        fe rxp = cx;\n        for (uint32_t k = 0; k < G; k++) {\n            // ---- prefetch", 1)
    nb = sub1(nb, "            cx = nx;\n            cy = ny;", "            rxp = rx;\n            cx = nx;\n            cy = ny;", 1)
    files["kng_engine.hip"] = t.replace(body, nb)
    return files


def v_s_nt(files):
    """product planes with the non-temporal hint like the rest of the state (the round-1 kernel)"""
    t = files["kng_engine.hip"]
    body = t[t.index("template <int SHARE, bool DSPLIT>\nKNG_DEV void walk_body"):t.index("__global__ void __launch_bounds__(256) kng_walk_kernel")]
    nb = body.replace("ld_prod(a.s01, a.s23,", "ld_fe(a.s01, a.s23,").replace("st_prod(a.s01, a.s23,", "st_fe(a.s01, a.s23,")
    files["kng_engine.hip"] = t.replace(body, nb)
    return files


def v_s_ld_plain_st_nt(files):
    """product planes: plain loads, non-temporal stores"""
    t = files["kng_engine.hip"]
    body = t[t.index("template <int SHARE, bool DSPLIT>\nKNG_DEV void walk_body"):t.index("__global__ void __launch_bounds__(256) kng_walk_kernel")]
    nb = body.replace("st_prod(a.s01, a.s23,", "st_fe(a.s01, a.s23,")
    files["kng_engine.hip"] = t.replace(body, nb)
    return files


def v_s_ld_nt_st_plain(files):
    """product planes: non-temporal loads, plain stores"""
    t = files["kng_engine.hip"]
    body = t[t.index("template <int SHARE, bool DSPLIT>\nKNG_DEV void walk_body"):t.index("__global__ void __launch_bounds__(256) kng_walk_kernel")]
    nb = body.replace("ld_prod(a.s01, a.s23,", "ld_fe(a.s01, a.s23,")
    files["kng_engine.hip"] = t.replace(body, nb)
    return files


def v_xy_st_plain(files):
    """x, y: non-temporal loads, plain stores"""
    t = files["kng_engine.hip"]
    t = sub1(t, "#define KNG_NT_STORE 1", "#define KNG_NT_STORE 0", 1)
    files["kng_engine.hip"] = t
    return files


def v_stagger(files):
    """workgroups start up to 7/8 of a step apart, so that the chip is never all-walking or all-inverting at once"""
    t = files["kng_engine.hip"]
    t = sub1(t, "    // pass 0: prefix products of dx in ascending order (GPUCompute.h:52-61 + GPUMath.h:1173-1177)\n",
             "    if (SHARE > 1) for (uint32_t w = 0; w < 12u * (blockIdx.x & 7u); w++) __builtin_amdgcn_s_sleep(127);\n    // pass 0: prefix products of dx in ascending order (GPUCompute.h:52-61 + GPUMath.h:1173-1177)\n", 1)
    files["kng_engine.hip"] = t
    return files


def v_setprio(files):
    """waves 4..7 of the 512-thread block (the younger half, which loses VALU arbitration) run at priority 1"""
    t = files["kng_engine.hip"]
    t = sub1(t, "    // pass 0: prefix products of dx in ascending order (GPUCompute.h:52-61 + GPUMath.h:1173-1177)\n",
             "    if (SHARE > 1 && (threadIdx.x >> 8)) __builtin_amdgcn_s_setprio(1);\n    // pass 0: prefix products of dx in ascending order (GPUCompute.h:52-61 + GPUMath.h:1173-1177)\n", 1)
    files["kng_engine.hip"] = t
    return files


def v_lds_b64(files):
    """jump-table words read one ds_read_b64 at a time (hipcc pairs them into ds_read2_b64, whose banking is 32 x 4 B)"""
    t = files["kng_engine.hip"]
    t = sub1(t, "    return fe{{tab[base + j], tab[base + 32 + j], tab[base + 64 + j], tab[base + 96 + j]}};",
             "    typedef __attribute__((address_space(3))) const uint64_t lds_u64;\n"
             "    const uint32_t addr = (uint32_t)(uintptr_t)(lds_u64 *)tab + 8u * (uint32_t)(base + j);\n"
             "    uint64_t w0, w1, w2, w3;\n"
             "    asm volatile(\"ds_read_b64 %0, %4\\n\\tds_read_b64 %1, %4 offset:256\\n\\tds_read_b64 %2, %4 offset:512\\n\\tds_read_b64 %3, %4 offset:768\\n\\ts_waitcnt lgkmcnt(0)\"\n"
             "                 : \"=&v\"(w0), \"=&v\"(w1), \"=&v\"(w2), \"=&v\"(w3) : \"v\"(addr) : \"memory\");\n"
             "    return fe{{w0, w1, w2, w3}};", 1)
    files["kng_engine.hip"] = t
    return files


VARIANTS = {
    "base": v_base,
    "no_s_traffic": v_no_s_traffic,
    "no_inversion": v_no_inversion,
    "no_comba_carry": v_no_comba_carry,
    "no_fold": v_no_fold,
    "no_state_store": v_no_state_store,
    "no_memory": v_no_memory,
    "s_nt": v_s_nt,
    "s_ld_plain_st_nt": v_s_ld_plain_st_nt,
    "s_ld_nt_st_plain": v_s_ld_nt_st_plain,
    "xy_st_plain": v_xy_st_plain,
    "setprio": v_setprio,
    "stagger": v_stagger,
    "lds_b64": v_lds_b64,
}


def build(name):
    d = os.path.join(OUT, name)
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(os.path.join(d, "kangaroo_amd", "csrc"))
    os.makedirs(os.path.join(d, "include"))
    shutil.copy(os.path.join(ROOT, "include", "kangaroo_hip.h"), os.path.join(d, "include"))
    files = {f: open(os.path.join(CSRC, f)).read() for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))}
    files = VARIANTS[name](files)
    for f, t in files.items():
        with open(os.path.join(d, "kangaroo_amd", "csrc", f), "w") as fh:
            fh.write(t)
    lib = os.path.join(d, "libkangaroo_hip.so")
    srcs = [os.path.join(d, "kangaroo_amd", "csrc", f) for f in files if f.endswith(".hip")]
    subprocess.check_call(["hipcc", *FLAGS, "-o", lib, *srcs], stderr=subprocess.DEVNULL)
    return lib


def main():
    names = sys.argv[1:] or list(VARIANTS)
    with ThreadPoolExecutor(max_workers=4) as ex:
        for n, lib in zip(names, ex.map(build, names)):
            print(n, "->", os.path.relpath(lib, ROOT))


if __name__ == "__main__":
    main()
