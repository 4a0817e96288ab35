"""`Kangaroo::SaveWork` / `Kangaroo::FectchKangaroos` of the reference program replaced at link time
(kangaroo_amd/host/Backup_kng.cpp).  oracle/Makefile links the probe oracle/saveprobe.cpp twice:
`saveprobe_ref` with the reference's Backup.o, `saveprobe_kng` with the two symbols of that object weakened and Backup_kng.o
bound over them.  Same state in -> the same work-file bytes out; the same file in -> the same kangaroos handed to the threads.
(GPU threads' records come from device snapshots: tests/test_gpu_snapshot.py and the KNG_SAVE_VERIFY run of
tests/test_gpu_reference_program.py.)  The binaries are built where /root/reference exists and travel with the tree."""
import filecmp
import os
import re
import subprocess

import pytest

from helpers import ref_binary

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(exe, *args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=600, env=e)
    assert out.returncode == 0, out.stdout[-800:] + out.stderr[-800:]
    return out.stdout


@pytest.mark.parametrize("seed,threads,adds", [(7, 3, 5000), (8, 1, 0), (9, 9, 70000), (10, 0, 300)])
def test_save_writes_the_bytes_the_reference_writes(tmp_path, seed, threads, adds):
    """header + HashTable::SaveTable + kangaroo section of `threads` parked CPU threads (Backup.cpp:449-563)"""
    a, b = tmp_path / "ref.work", tmp_path / "kng.work"
    oa = _run(ref_binary("saveprobe_ref"), "save", a, seed, threads, adds)
    ob = _run(ref_binary("saveprobe_kng"), "save", b, seed, threads, adds)
    assert filecmp.cmp(a, b, shallow=False)
    assert os.path.getsize(a) == 156 + 8 * (1 << 18) + 32 * adds + 8 + 96 * 1024 * threads
    # what a caller can observe besides the file: the request flag is down again, the "done [.. MB]" line is there
    for o in (oa, ob):
        assert "saveRequest 0" in o and re.search(r"done \[[0-9.]+ MB\]", o), o


def test_split_save_resets_the_table_like_the_reference(tmp_path):
    """-wsplit: file name + time stamp, HashTable::Reset after the save (Backup.cpp:470-472, :550-551)"""
    outs = []
    for v in ("ref", "kng"):
        d = tmp_path / v
        d.mkdir()
        o = _run(ref_binary(f"saveprobe_{v}"), "save", d / "w", 11, 2, 4000, "split")
        files = sorted(os.listdir(d))
        assert len(files) == 1 and re.fullmatch(r"w_\d{2}\w{3}\d{2}_\d{6}", files[0]), files
        assert "table 0" in o  # emptied
        outs.append(d / files[0])
    assert filecmp.cmp(outs[0], outs[1], shallow=False)


@pytest.mark.parametrize("cpu_threads,gpu_kangaroos", [(3, 0), (2, 0), (4, 0), (1, 2048), (0, 3072), (2, 2048), (3, 512)])
def test_fetch_hands_out_the_kangaroos_the_reference_hands_out(tmp_path, cpu_threads, gpu_kangaroos):
    """LoadWork + FectchKangaroos (Backup.cpp:149-208, :286-364) of one file (3 x 1024 kangaroos) into more, fewer and as many
    threads as saved it; with a GPU thread behind the CPU threads the reference fills 3 x N Int for it, ours leaves the place of
    its records in the file -- the probe prints the kangaroos either would put on the device: the same ones, and the CPU
    threads get the same ones, and the same number is left over / created."""
    w = tmp_path / "in.work"
    _run(ref_binary("saveprobe_ref"), "save", w, 21, 3, 2500)

    def norm(text):
        # the kangaroos (CPU threads: "t g x y d", GPU thread: "G g x y d"), what is left over, what the summary line says;
        # the progress lines differ by design (the reference announces a FetchWalks per thread, ours only the CPU threads')
        return [ln for ln in text.splitlines() if re.match(r"(G |\d+ \d+ |nbLoadedWalk |FectchKangaroos: )", ln)]

    la = _run(ref_binary("saveprobe_ref"), "load", w, cpu_threads, gpu_kangaroos)
    lb = _run(ref_binary("saveprobe_kng"), "load", w, cpu_threads, gpu_kangaroos)
    na, nb = norm(la), norm(lb)
    file_has = 3 * 1024
    from_file_gpu = max(0, min(gpu_kangaroos, file_has - 1024 * cpu_threads))
    if gpu_kangaroos and from_file_gpu < gpu_kangaroos:
        # the tail the file does not have is CREATED (random): the reference creates it on the host, ours on the device later --
        # compare only what comes from the file
        cut = lambda lines: [ln for ln in lines if not (ln.startswith("G ") and int(ln.split()[1]) >= from_file_gpu)]  # noqa: E731
        na, nb = cut(na), cut(nb)
    assert na == nb
    if gpu_kangaroos:
        assert f"gpu thread: plan {from_file_gpu} records" in lb or from_file_gpu == 0
        assert sum(ln.startswith("G ") for ln in nb) == len(range(0, from_file_gpu, 41))
