#!/usr/bin/env python3
"""Generate tests/golden/ref_vectors.json and ref_hashtable.json from the REFERENCE's own objects.

Runs in the build container only (needs /root/reference to build oracle/_ref/refprobe via
`make -C oracle ref`).  The GPU box and the test-suite only ever read the committed JSON.

usage: python tools/make_golden.py [seed]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main() -> None:
    seed = sys.argv[1] if len(sys.argv) > 1 else "0x5EED1234"
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    out = os.path.join(ROOT, "tests", "golden", "ref_vectors.json")
    subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "refprobe"), out, seed],
                          stdout=subprocess.DEVNULL)
    with open(out) as f:
        data = json.load(f)  # validates
    with open(out, "w") as f:
        json.dump(data, f, separators=(",", ":"))
        f.write("\n")
    print("wrote", out, os.path.getsize(out), "bytes;",
          {k: (len(v) if hasattr(v, "__len__") else v) for k, v in data.items()})
    # HashTable::Add sequence (statuses, collision read-back, final buckets) -- SURVEY 8(f) rows 1/4
    out = os.path.join(ROOT, "tests", "golden", "ref_hashtable.json")
    subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "refprobe"), "--hashtable", out],
                          stdout=subprocess.DEVNULL)
    with open(out) as f:
        data = json.load(f)
    with open(out, "w") as f:
        json.dump(data, f, separators=(",", ":"))
        f.write("\n")
    print("wrote", out, os.path.getsize(out), "bytes;", len(data["adds"]), "adds,", len(data["buckets"]), "buckets")


if __name__ == "__main__":
    main()
