#!/usr/bin/env python3
"""tests/golden/in40_25keys.txt: range + the first 25 public keys of the reference's shipped multi-key known-answer input
(/root/reference/VC_CUDA8/in40_1000.txt, 40-bit range, 1000 keys; the reference solves them one after the other, creating
and destroying its GPUEngine once per key: Kangaroo.cpp:1021-1075, ctor :523, `delete gpu` :634).  Input DATA only; the
answers are checked by recomputing the public key from every printed private key.  Run where /root/reference exists."""
import os
import sys

SRC = "/root/reference/VC_CUDA8/in40_1000.txt"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 25
lines = [l.strip() for l in open(SRC) if l.strip()]
out = os.path.join(ROOT, "tests", "golden", f"in40_{n}keys.txt")
with open(out, "w") as f:
    f.write("\n".join(lines[: 2 + n]) + "\n")
print(out, len(lines) - 2, "keys in the source,", n, "kept")
