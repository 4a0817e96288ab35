"""`class HashTable` of the reference re-implemented for link-time replacement (kangaroo_amd/host/HashTable_kng.cpp, SURVEY 8 f4,
VERDICT r4 item 1).  oracle/Makefile links the probe oracle/refprobe.cpp twice: `refprobe` with the reference's HashTable.o,
`refprobe_kng` with HashTable_kng.o (compiled against the reference's own HashTable.h) in its place.  Whatever the class lets a
caller observe -- Add statuses, kDist / kType after a collision, bucket contents, maxItem words, SaveTable / LoadTable / MergeH
bytes -- must be the same from both.  The binaries are built where /root/reference exists and travel with the tree."""
import filecmp
import json
import os
import subprocess

import pytest

from helpers import ref_binary

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(exe, *args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=600, env=e)
    assert out.returncode == 0, out.stdout[-800:] + out.stderr[-800:]
    return [ln for ln in out.stdout.splitlines() if ln and not ln.startswith(("Jump Avg", "DP size", "Range width"))]


def test_golden_add_sequence_through_the_replacement_class(tmp_path):
    """The committed golden sequence (tests/golden/ref_hashtable.json, written by the REFERENCE's HashTable::Add: 900 adds
    with repeats, collisions, negative distances, crowded buckets) replayed through HashTable_kng.o by the same probe."""
    exe = ref_binary("refprobe_kng")
    out = tmp_path / "ht.json"
    _run(exe, "--hashtable", out)
    with open(out) as f, open(os.path.join(ROOT, "tests", "golden", "ref_hashtable.json")) as g:
        got, want = json.load(f), json.load(g)
    assert got["adds"] == want["adds"]          # statuses and collision read-backs
    assert got["count"] == want["count"]
    assert got["buckets"] == want["buckets"]    # nbItem, maxItem, entries in order, for every bucket touched


@pytest.mark.parametrize("adds,seed,buckets,tail", [
    (120000, 1, 48, None),      # ~2500 entries per bucket: every size class, many folds
    (120000, 7, 3, None),       # 40 000 per bucket
    (60000, 9, 4096, None),     # small buckets only
    (80000, 5, 3, 0),           # KNG_HT_TAIL=0: folded after every insertion (the reference's invariant at all times)
    (80000, 5, 3, 1),
    (80000, 5, 3, 4096),
])
def test_stress_sequence_equals_the_reference_class(tmp_path, adds, seed, buckets, tail):
    """A long sequence through all three Add overloads, a SaveTable / LoadTable round trip half way, MergeH of two overlapping
    tables and LoadTable of the merge: one stdout line (status hash, kDist / kType hash, counts) and two files, both programs."""
    ref, kng = ref_binary("refprobe"), ref_binary("refprobe_kng")
    a, b = tmp_path / "ref.tbl", tmp_path / "kng.tbl"
    la = _run(ref, "--hashtable-stress", a, adds, seed, buckets)
    lb = _run(kng, "--hashtable-stress", b, adds, seed, buckets, env=None if tail is None else {"KNG_HT_TAIL": str(tail)})
    assert la == lb and len(la) == 1 and " coll " in la[0]
    assert filecmp.cmp(a, b, shallow=False)
    assert filecmp.cmp(str(a) + ".merge", str(b) + ".merge", shallow=False)


@pytest.mark.parametrize("n,seed,buckets,threads", [(150000, 1, 4096, 4), (150000, 2, 7, 8), (100000, 3, 1, 3)])
def test_batch_ingest_equals_the_per_point_path(tmp_path, n, seed, buckets, threads):
    """kng_ht_ingest (what SolveKeyGPU_kng.cpp feeds the table with) against GPUEngine::Launch's conversion + HashTable::Add per
    point: same statuses, same stored distance for every collision, same SaveTable bytes -- with the reference's object doing
    the per-point path in one program and ours in the other -- and, from several threads at once, the same set of x with every
    point either stored or reported."""
    ref, kng = ref_binary("refprobe"), ref_binary("refprobe_kng")
    a, b = tmp_path / "ref.tbl", tmp_path / "kng.tbl"
    la = _run(ref, "--hashtable-ingest", a, n, seed, buckets, threads)
    lb = _run(kng, "--hashtable-ingest", b, n, seed, buckets, threads)
    assert la[0] == lb[0] and filecmp.cmp(a, b, shallow=False)
    assert "identical, table identical" in lb[1], lb
    assert lb[2].endswith("x set identical") and f"of {n}" in lb[2], lb


@pytest.mark.parametrize("pushes,threads,cap", [(30, 3, 4), (20, 1, 2), (25, 4, 2000)])
def test_gpu_thread_to_table_thread_queue_without_a_gpu(pushes, threads, cap):
    """kng_ingest.h (the queue inside SolveKeyGPU_kng.cpp): two producers -- two GPU threads -- each with its own table threads
    feed ONE table through bounded queues; entries + events must equal the records pushed, the queue must never hold more than
    its capacity, a small capacity must hold the producer back (back-pressure instead of loss), and an Ingest destroyed with work
    still queued must return."""
    exe = ref_binary("ingestprobe")
    lines = _run(exe, pushes, threads, cap)
    assert lines[0].endswith("CONSISTENT") and "INCONSISTENT" not in lines[0], lines
    assert lines[1] == "shutdown with queued work: returned"
    # round 6: hold for a work-file save -- table frozen, queue beyond its normal bound, producer blocked at the hold bound, the
    # table threads release themselves when the generation is finished, events carry the tag of their push
    assert lines[2].startswith("hold: table frozen") and lines[2].endswith(" CONSISTENT"), lines[2]
    assert lines[3] == "tables: 2 x 20 alive at once, released and deleted CONSISTENT"  # (16 registry slots aborted here in round 5)
    import re

    m = re.search(r"high water (\d+) / (\d+) of (\d+) blocked ([0-9.]+) s", lines[0])
    bound = max(cap, 2 * threads)  # a push is cut into one chunk per table thread: the bound is at least two chunks per thread
    assert int(m.group(3)) == bound and int(m.group(1)) <= bound and int(m.group(2)) <= bound
    if cap <= 4:
        assert float(m.group(4)) > 0.0      # ~20 000 points per push against 1-3 table threads: the producer had to wait


def test_link_time_replacements_are_really_linked_in():
    """What each program binary is made of, from its symbol table: `kangaroo_hip` carries the reference's HashTable and
    SolveKeyGPU (no kng_ht_* symbols); `kangaroo_hip_ht` our class HashTable but the reference's loop; `kangaroo_mi355x` both
    replacements -- the batch interface, the table-thread queue (kng_ingest) and ONE definition of Kangaroo::SolveKeyGPU, the one
    that references kng_drain_view.  A silent failure of `objcopy -W` or of the link order would show here, without a GPU."""
    def syms(name):
        out = subprocess.run(["nm", "-C", ref_binary(name)], capture_output=True, text=True, timeout=60)
        assert out.returncode == 0, out.stderr
        return out.stdout

    hip, ht, full = syms("kangaroo_hip"), syms("kangaroo_hip_ht"), syms("kangaroo_mi355x")
    assert "kng_ht_ingest" not in hip and "kng_ingest::" not in hip
    assert " T kng_ht_ingest" in ht and "kng_ingest::" not in ht
    assert " T kng_ht_ingest" in full and "kng_ingest::Ingest" in full and "kng_drain_view" in full
    def defs_of(text, member):  # definitions of the member itself (not its .cold part, not the lambdas inside it)
        return [ln.split()[1] for ln in text.splitlines() if ln.endswith(member) and len(ln.split()) >= 3]

    for text in (hip, ht, full):
        assert len(defs_of(text, "Kangaroo::SolveKeyGPU(TH_PARAM*)")) == 1
    assert defs_of(full, "Kangaroo::SolveKeyGPU(TH_PARAM*)") == ["T"]
    # round 6: the work-file members (Backup_kng.cpp), and the reference's own definitions kept under their second names, to
    # which everything not improved on is delegated (client mode, saves without kangaroos, server-kept kangaroos)
    for member in ("Kangaroo::SaveWork(unsigned long, double, TH_PARAM*, int)", "Kangaroo::FectchKangaroos(TH_PARAM*)"):
        assert defs_of(hip, member) == ["T"] and defs_of(ht, member) == ["T"] and defs_of(full, member) == ["T"], member
    for alias in ("kng_ref_SolveKeyGPU", "kng_ref_SaveWork", "kng_ref_FectchKangaroos"):
        assert f" T {alias}" in full and alias not in hip and alias not in ht, alias
    assert "kng_snapshot_read" in full and "kng_snapshot_read" not in hip
    assert "kng_drain_view" not in hip   # the reference's loop goes through GPUEngine::Launch (kng_drain)


def test_pool_of_owner_partitioned_table_threads_end_to_end():
    """oracle/poolbench.cpp (the measurement tool behind INTEGRATION.md's `-d` table) on a small count: two producers route
    300 000 uniform points to three table threads by bucket ownership; every point ends in the table."""
    lines = _run(ref_binary("poolbench"), 300000, 100000, 3, 2, 20000)
    assert lines[0].startswith("# HashTable_kng.o behind kng_ingest.h: 3 owner-partitioned table threads, 2 producers")
    assert lines[-1].startswith("# 300000 entries in "), lines[-1]
