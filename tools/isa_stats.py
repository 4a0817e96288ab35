#!/usr/bin/env python3
"""Opcode histogram of the walk kernel's hot loop (the per-kangaroo k-loop of walk_body), from `hipcc -S`.

usage: python tools/isa_stats.py [--kernel SUBSTR] [--src kangaroo_amd/csrc/kng_engine.hip] [--dump FILE] [-D...]

The hot loop is found structurally: inside the chosen kernel, the backward branch with the longest span that
contains no call (s_swappc_b64: the step loop contains the fe_inv call) -- that is the k-loop.  Blocks the
compiler moved out of line (the KNG_RARE_PATH bodies sit behind the loop) are not counted: the histogram is the
straight-line fast path a wave executes once per kangaroo-jump.  Classes follow profiles/r01_instr_throughput_gfx950.txt:
  slow  = 4 SIMD cycles per wave64 (carry ops, VOP3, 64-bit, v_mad_u64_u32)
  fast  = 2 SIMD cycles (plain VOP2: v_mov_b32, v_add_u32, v_and/or/xor, 32-bit shifts)
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FAST = {"v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32",
        "v_lshlrev_b32", "v_ashrrev_i32", "v_not_b32", "v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_mov_b32_e32",
        "v_min_u32", "v_max_u32"}


def strip(op):
    for suf in ("_e32", "_e64", "_dpp", "_sdwa"):
        if op.endswith(suf):
            op = op[: -len(suf)]
    return op


def compile_s(src, defines):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Wno-unused-command-line-argument",
           "-o", out, src] + [f"-D{d}" for d in defines]
    subprocess.check_call(cmd)
    return out


def kernel_lines(path, substr):
    lines = open(path).read().split("\n")
    start = end = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+|kng_\w+):", l)
        if m and substr in m.group(1) and start is None:
            start = i
        elif start is not None and l.startswith("\t.end_amdhsa_kernel") or (start is not None and re.match(r"^\.Lfunc_end", l)):
            end = i
            break
    if start is None:
        raise SystemExit(f"kernel containing '{substr}' not found")
    return lines[start:end]


def hot_loop(lines):
    """Fast path of the deepest loop: LLVM annotates every block with its loop ("in Loop: Header=BBx_y", the header
    itself with "This Inner Loop Header: Depth=2").  Blocks that start behind a KNG_RARE_PATH()/"cold path" marker, and
    blocks only reachable from such blocks, are dropped: what remains is what a wave executes per kangaroo-jump."""
    header = None
    for i, l in enumerate(lines):
        if "This Inner Loop Header: Depth=2" in l:
            m = re.match(r"^\.L(BB\d+_\d+):", lines[i - 1])
            if m:
                header = m.group(1)
    if header is None:
        raise SystemExit("no depth-2 inner loop found")
    # basic blocks: split at labels and behind branches
    blocks, cur, name, in_loop = [], [], None, False
    anon = 0

    def flush():
        nonlocal cur
        if cur:
            blocks.append({"name": name, "lines": cur, "loop": in_loop})
        cur = []

    for l in lines:
        m = re.match(r"^\.L(BB\d+_\d+):(.*)", l)
        if m:
            flush()
            name = m.group(1)
            in_loop = (name == header) or (f"Header={header} " in l + " ")
            continue
        if l.lstrip().startswith("; %bb."):
            continue
        cur.append(l)
        if re.match(r"^\s+s_c?branch", l) or re.match(r"^\s+s_endpgm", l):
            flush()
            anon += 1
            name = f"{name}+{anon}"
    flush()
    # an unlabeled continuation of a block inherits its loop membership (set above through `name`/`in_loop`)
    idx = {b["name"]: k for k, b in enumerate(blocks)}
    succ = {k: set() for k in range(len(blocks))}
    for k, b in enumerate(blocks):
        last = b["lines"][-1] if b["lines"] else ""
        m = re.match(r"^\s+(s_c?branch\w*)\s+\.L(BB\d+_\d+)", last)
        if m:
            if m.group(2) in idx:
                succ[k].add(idx[m.group(2)])
            if m.group(1) != "s_branch" and k + 1 < len(blocks):
                succ[k].add(k + 1)
        elif k + 1 < len(blocks) and "s_endpgm" not in last:
            succ[k].add(k + 1)
    pred = {k: set() for k in range(len(blocks))}
    for k, ss in succ.items():
        for t in ss:
            pred[t].add(k)
    rare = {k for k, b in enumerate(blocks) if any(("; rare path" in l or "; cold path" in l) for l in b["lines"])}
    changed = True
    while changed:
        changed = False
        for k, b in enumerate(blocks):
            if k in rare or not b["loop"] or b["name"] == header:
                continue
            ps = [q for q in pred[k] if blocks[q]["loop"]]
            if ps and all(q in rare for q in ps):
                rare.add(k)
                changed = True
    body = []
    for k, b in enumerate(blocks):
        if b["loop"] and k not in rare:
            body.append(f"; ---- block {b['name']}")
            body.extend(b["lines"])
    return body


def histogram(body):
    h = collections.Counter()
    for l in body:
        m = re.match(r"^\s+([a-z_0-9]+)", l)
        if not m or l.lstrip().startswith((";", ".")):
            continue
        op = strip(m.group(1))
        if op == "s_nop":
            n = int(l.split()[1]) + 1
            h["s_nop"] += 1
            h["(nop wait states)"] += n
            continue
        h[op] += 1
    return h


def summarise(h):
    tot = sum(v for k, v in h.items() if not k.startswith("("))
    valu = {k: v for k, v in h.items() if k.startswith("v_")}
    fast = sum(v for k, v in valu.items() if k in FAST)
    slow = sum(valu.values()) - fast
    salu = sum(v for k, v in h.items() if k.startswith("s_") and k not in ("s_nop", "s_waitcnt"))
    vmem = sum(v for k, v in h.items() if k.startswith(("global_", "buffer_", "scratch_", "flat_")))
    lds = sum(v for k, v in h.items() if k.startswith("ds_"))
    mad = h.get("v_mad_u64_u32", 0)
    carry = sum(v for k, v in valu.items() if k.startswith(("v_addc", "v_subb", "v_add_co", "v_sub_co", "v_subrev_co", "v_subbrev")))
    return {"instructions": tot, "valu": sum(valu.values()), "valu_slow": slow, "valu_fast": fast, "v_mad_u64_u32": mad, "carry_ops": carry,
            "v_mov_b32": h.get("v_mov_b32", 0), "s_nop": h.get("s_nop", 0), "nop_wait_states": h.get("(nop wait states)", 0), "salu": salu,
            "s_waitcnt": h.get("s_waitcnt", 0), "vmem": vmem, "lds": lds,
            "simd_cycles_model": 4 * slow + 2 * fast}


def resources(path):
    """Per walk kernel of a `hipcc -S` listing: what the code object's descriptor says (VGPRs, LDS, scratch bytes per lane) and
    where its scratch instructions are -- inside the generated asm statement (the per-kangaroo loop: must be none) or in the
    compiler's code around it (the inversion tree, the entry pass, the exact path).  VERDICT r5 weak 1: the headline kernel
    sits at the 2-waves-per-SIMD ceiling (256 VGPRs) WITH spills; they must stay out of the loop."""
    txt = open(path).read()
    desc = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
        g = lambda k: int((re.search(r"\.amdhsa_" + k + r"\s+(\S+)", m.group(2)) or [None, "0"])[1], 0)  # noqa: E731
        desc[m.group(1)] = {"vgprs": g("next_free_vgpr"), "sgprs": g("next_free_sgpr"), "lds_bytes": g("group_segment_fixed_size"),
                            "scratch_bytes_per_lane": g("private_segment_fixed_size")}
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", txt):
        if m.group(1) in desc:
            desc[m.group(1)]["vgpr_spill_count"] = int(m.group(2))
    out, cur, inasm, blk = {}, None, False, 0
    for l in txt.split("\n"):
        m = re.match(r"^(_Z\w+|kng_\w+):", l)
        if m:
            cur = m.group(1)
            out[cur] = dict(desc.get(cur, {}), scratch_ops_in_asm_statements=0, scratch_ops_outside=0, asm_statements_over_500_lines=0)
        if cur is None:
            continue
        if ";;#ASMSTART" in l:
            inasm, blk = True, 0
            continue
        if ";;#ASMEND" in l:
            inasm = False
            out[cur]["asm_statements_over_500_lines"] += blk > 500
            continue
        blk += inasm
        if re.search(r"\bscratch_(load|store)", l):
            out[cur]["scratch_ops_in_asm_statements" if inasm else "scratch_ops_outside"] += 1
    return {k: v for k, v in out.items() if "walk" in k}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--resources", action="store_true", help="registers / LDS / scratch of every walk kernel and where its scratch instructions are")
    ap.add_argument("--kernel", default="kng_walk_share_kernelILi8ELb1ELb0",
                    help="mangled-name substring; <SHARE, DSPLIT, ASM> = ILi8ELb1ELb0 is the compiler-scheduled loop of the default geometry")
    ap.add_argument("--src", default=os.path.join(ROOT, "kangaroo_amd", "csrc", "kng_engine.hip"))
    ap.add_argument("--asm", help="use this .s instead of compiling")
    ap.add_argument("--dump", help="write the loop body here")
    ap.add_argument("-D", action="append", default=[])
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    if a.resources:
        path = a.asm or compile_s(a.src, a.D)
        for k, v in resources(path).items():
            print(k, v)
        if not a.asm:
            os.unlink(path)
        return
    if re.search(r"kng_walk_share_kernelILi\dELb[01]ELb1", a.kernel):
        # ASM = true: the per-kangaroo loop is ONE generated asm statement (kng_walk_asm.h) that LLVM's loop annotations do not
        # see -- the loop this tool would find is the compiler-scheduled one of a launch's last step.  The generator prints the
        # scheduled loop's own instruction mix (tools/gen_walk_asm.py; profiles/r03_isa_stats_after.txt).
        print("the scheduled asm loop is counted by its generator: python tools/gen_walk_asm.py (see profiles/r03_isa_stats_after.txt)")
        return
    path = a.asm or compile_s(a.src, a.D)
    body = hot_loop(kernel_lines(path, a.kernel))
    if a.dump:
        open(a.dump, "w").write("\n".join(body) + "\n")
    h = histogram(body)
    s = summarise(h)
    print(f"kernel *{a.kernel}*: k-loop of {s['instructions']} instructions")
    for k, v in s.items():
        print(f"  {k:22s} {v}")
    print("  -- opcodes --")
    for k, v in h.most_common(a.top):
        print(f"  {k:28s} {v}")
    if not a.asm:
        os.unlink(path)


if __name__ == "__main__":
    main()
