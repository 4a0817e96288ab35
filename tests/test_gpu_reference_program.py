"""The reference PROGRAM at the rate of the kernel (SURVEY 8 f1 / f4 brought to the CLI).

oracle/_ref/kangaroo_hip    = the reference's host code, unmodified, + our `class GPUEngine` (the drop-in boundary);
oracle/_ref/kangaroo_mi355x = the same objects with HashTable.o and the Kangaroo::SolveKeyGPU symbol replaced at LINK time by
                              kangaroo_amd/host/HashTable_kng.cpp and SolveKeyGPU_kng.cpp (no reference source edited).
Both are built by oracle/Makefile where /root/reference exists and travel to the GPU box with the tree."""
import os
import re
import shutil
import signal
import subprocess
import time

import numpy as np
import pytest

from helpers import ref_binary

pytestmark = pytest.mark.gpu

# BASELINE configs[2]: 80-bit range.  The public key is NOT in the range (it is the key of in64.txt): the search cannot end,
# the herd and the table behave as in any unsolved run.
IN80 = ("B60E83280258A40F9CDF1649744D730D6E939DE92A2B00000000000000000000\n"
        "B60E83280258A40F9CDF1649744D730D6E939DE92A2BFFFFFFFFFFFFFFFFFFFF\n"
        "03BB113592002132E6EF387C3AEBC04667670D4CD40B2103C7D0EE4969E9FF56E4\n")
# elapsed time as Kangaroo::GetTimeStr prints it (Thread.cpp:124-162): "42s", "01:10", "01:02:03"
STATUS = re.compile(r"\[([0-9.]+) MK/s\]\[GPU [0-9.]+ MK/s\]\[Count 2\^([0-9.]+)\]\[Dead (\d+)\]\[([0-9:]+s?) \(Avg [^)]*\)\]\[([0-9.]+)/([0-9.]+)(MB|GB)\]")


def _seconds(t):
    if t.endswith("s"):
        return int(t[:-1])
    v = 0
    for part in t.split(":"):
        v = v * 60 + int(part)
    return v


def _run(cmd, seconds, env=None, until=None):
    """Run the program unbuffered for `seconds` after its walk has started (or until `until` appears), stop it, return output."""
    e = dict(os.environ)
    e.update(env or {})
    if shutil.which("stdbuf"):
        cmd = ["stdbuf", "-o0", "-e0"] + cmd
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=e, start_new_session=True)
    os.set_blocking(proc.stdout.fileno(), False)
    buf, started, t0 = b"", None, time.time()
    while time.time() - t0 < seconds + 240 and proc.poll() is None:
        time.sleep(0.25)
        try:
            chunk = proc.stdout.read()
        except BlockingIOError:
            chunk = None
        if chunk:
            buf += chunk
        if started is None and b"kangaroos [" in buf:
            started = time.time()
        if until and until.encode() in buf:
            break
        if started is not None and time.time() - started > seconds:
            break
    if proc.poll() is None:
        os.killpg(proc.pid, signal.SIGKILL)
    proc.wait()
    try:
        rest = proc.stdout.read()
        if rest:
            buf += rest
    except Exception:
        pass
    return buf.decode(errors="replace").replace("\r", "\n")


def _status_lines(text):
    out = []
    for m in STATUS.finditer(text):
        used = float(m.group(5)) * (1024.0 if m.group(7) == "GB" else 1.0)
        out.append({"mks": float(m.group(1)), "count": 2.0 ** float(m.group(2)), "t": _seconds(m.group(4)), "used_mb": used})
    return out


def _kernel_rate_gks(kng):
    """jumps per second of the walk kernel alone at the reference program's default grid (HIP events, 12 launches)."""
    import kangaroo_amd.hostlib as hl

    gx, gy = kng.default_grid(0)
    jd, jx, jy, _ = hl.jump_table(80)
    with kng.GPUEngine(gx, gy, 0, 1 << 17) as eng:
        eng.SetParams(hl.dp_mask(14), jd, jx, jy)
        eng.CreateHerdOnDevice(80, seed=5)
        ms = []
        for _ in range(15):
            eng.callKernel()
            eng.wait()
            eng.drain(raw=True)
            ms.append(eng.last_kernel_ms())
        return eng.nbKangaroo * 64 / (np.median(ms[3:]) * 1e-3) / 1e9


def test_reference_program_runs_at_the_kernel_rate_at_its_own_dp(kng, tmp_path):
    """`kangaroo_mi355x -t 0 -gpu in80.txt`: default grid (2^23 kangaroos), the DP size the program suggests itself (14), no
    other option.  The unmodified program on the same engine does 19 GK/s in its first minute and 12 GK/s over three (its
    HashTable and its lock-step loop, profiles/r03_*); with the two link-time replacements the GPU thread never waits for the
    host: >= 0.95 of the kernel-only rate, measured over ~50 s here (and over three minutes in profiles/r05_*).  The exact
    figure is the program's own launch count over its own clock (KNG_STATS line); the status line's Count column, which has a
    resolution of 0.7 % in count and 1 s in time, must agree within 6 %."""
    exe = ref_binary("kangaroo_mi355x")
    kernel = _kernel_rate_gks(kng)
    cfg = tmp_path / "in80.txt"
    cfg.write_text(IN80)
    text = _run([exe, "-t", "0", "-gpu", "-m", "0.32", str(cfg)], 150, env={"KNG_STATS": "1"}, until="SolveKeyGPU_kng GPU#0: ")
    assert "Suggested DP: 14" in text and "items lost" not in text, text[-1500:]
    m = re.search(r"SolveKeyGPU_kng GPU#0: (\d+) launches in ([0-9.]+) s = ([0-9.]+) MK/s; points (\d+) \(lost (\d+)\), events (\d+) \(\+\d+ stale\); "
                  r"GPU thread waited ([0-9.]+) s for kernels, ([0-9.]+) s for queue room", text)
    assert m, text[-2500:]
    launches, wall, mks, points, lost = int(m.group(1)), float(m.group(2)), float(m.group(3)), int(m.group(4)), int(m.group(5))
    assert launches > 1100 and lost == 0
    assert abs(points / launches - 32768) < 600                      # 2^29 jumps / 2^14 per launch, every one delivered
    assert mks / 1e3 >= 0.95 * kernel, (mks, kernel, text[-600:])
    st = [s for s in _status_lines(text) if s["t"] >= 8]
    assert len(st) >= 10
    by_count = (st[-1]["count"] - st[0]["count"]) / (st[-1]["t"] - st[0]["t"]) / 1e9
    assert abs(by_count / (mks / 1e3) - 1) < 0.06, (by_count, mks)
    # the table really holds them: 32 bytes per point in the "used" figure of the status line
    assert abs(st[-1]["used_mb"] * 1048576 / 32 / (st[-1]["count"] / 2 ** 14) - 1) < 0.05


@pytest.mark.parametrize("program", ["kangaroo_hip", "kangaroo_mi355x"])
def test_maxfound_of_the_program_is_a_floor_not_a_ceiling(tmp_path, program):
    """`-g 512,128 -d 11` (the DP size the program suggests for eight MI355X): a launch yields 262 144 points, twice the
    `maxFound = 65536*2` the program hard-codes (Kangaroo.cpp:523).  The shim sizes the engine from herd and mask in SetParams:
    nothing is lost, no warning, and the table grows by one point per 2^11 jumps."""
    exe = ref_binary(program)
    cfg = tmp_path / "in80.txt"
    cfg.write_text(IN80)
    text = _run([exe, "-t", "0", "-gpu", "-g", "512,128", "-d", "11", str(cfg)], 14, env={"KNG_STATS": "1"})
    assert "DP size: 11" in text and "items lost" not in text and "Warning" not in text, text[-1500:]
    st = _status_lines(text)
    assert len(st) >= 3, text[-1500:]
    last = st[-1]
    stored = last["used_mb"] * 1048576 / 32 - 2 ** 18 * 8 / 32   # "used" = 8 bytes per bucket + 32 per point (HashTable.cpp:328)
    # the status thread reads counters and table at slightly different moments, and points of the last launches are still on
    # their way: a band, but one that a table missing half of every launch (131 072 of 262 144) cannot reach
    assert 0.80 < stored / (last["count"] / 2 ** 11) < 1.05, (stored, last)


IN56 = "0\nFFFFFFFFFFFFFF\n02E9F43F810784FF1E91D8BC7C4FF06BFEE935DA71D7350734C3472FE305FEF82A\n"   # the reference's in.txt
IN56_ANSWER = "Priv: 0x378ABDEC51BC5D"


def test_cpu_walkers_and_the_gpu_feed_one_table(tmp_path):
    """`kangaroo_mi355x -t 2 -gpu`: the program's CPU threads (SolveKeyCPU, unmodified: AddToTable under ghMutex, per point) and
    the replaced SolveKeyGPU (table threads, kng_ht_ingest, no ghMutex) insert into the same HashTable at the same time -- the
    stripe locks of HashTable_kng.cpp are what makes that safe.  The shipped 56-bit key must come out."""
    exe = ref_binary("kangaroo_mi355x")
    cfg = tmp_path / "in.txt"
    cfg.write_text(IN56)
    out = subprocess.run([exe, "-t", "2", "-gpu", "-g", "16,128", str(cfg)], capture_output=True, text=True, timeout=300)
    assert IN56_ANSWER in out.stdout, out.stdout[-2000:] + out.stderr[-500:]
    assert "Number of CPU thread: 2" in out.stdout


def test_client_mode_through_the_replaced_solvekeygpu(tmp_path):
    """Client / server mode is outside the hot-path scope, but the replaced SolveKeyGPU still has to carry it: in
    client mode the points go to SendToServer in the reference's lock-step shape, not to the table.  A server (the unmodified
    CPU program, `-s`) and one GPU client (`kangaroo_mi355x -c`) on the loopback interface solve the 56-bit key."""
    import socket

    server_exe, client_exe = ref_binary("kangaroo_cpu"), ref_binary("kangaroo_mi355x")
    cfg = tmp_path / "in.txt"
    cfg.write_text(IN56)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    wrap = ["stdbuf", "-o0", "-e0"] if shutil.which("stdbuf") else []
    server = subprocess.Popen(wrap + [server_exe, "-s", "-sp", str(port), "-d", "8", str(cfg)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              start_new_session=True)
    client = None
    try:
        time.sleep(2.0)
        client = subprocess.Popen(wrap + [client_exe, "-c", "127.0.0.1", "-sp", str(port), "-t", "0", "-gpu", "-g", "16,128"], stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT, start_new_session=True)
        os.set_blocking(server.stdout.fileno(), False)
        buf, t0 = b"", time.time()
        while time.time() - t0 < 240 and IN56_ANSWER.encode() not in buf and server.poll() is None:
            time.sleep(0.5)
            try:
                chunk = server.stdout.read()
            except BlockingIOError:
                chunk = None
            if chunk:
                buf += chunk
        text = buf.decode(errors="replace").replace("\r", "\n")
        assert IN56_ANSWER in text, text[-2500:]
    finally:
        for p_ in (client, server):
            if p_ is not None and p_.poll() is None:
                os.killpg(p_.pid, signal.SIGKILL)
            if p_ is not None:
                p_.wait()


STATS = re.compile(r"SolveKeyGPU_kng GPU#0: (\d+) launches in ([0-9.]+) s = ([0-9.]+) MK/s; points (\d+) \(lost (\d+)\), events (\d+) \(\+(\d+) stale\); "
                   r"GPU thread waited ([0-9.]+) s for kernels, ([0-9.]+) s for queue room, ([0-9.]+) s at (\d+) save points")


def test_work_file_saves_do_not_park_the_gpu(kng, tmp_path, orc):
    """SURVEY 8 f3 in the reference program: `kangaroo_mi355x -t 0 -gpu -ws -w f -wi 6` at the default herd.
    The unmodified save path parks the GPU 1.5-2.7 s per save at this herd (profiles/r06_save_cost_before.txt: GetKangaroos into
    3 x 2^23 Int, 25 M fwrite calls).  With Backup_kng.cpp + the device snapshot the walk goes on: the run WITH saves delivers
    >= 0.97 of the kernel rate over its whole wall time (saves included), and KNG_SAVE_VERIFY=1 compares every streamed 96-byte
    record with GPUEngine::GetKangaroos -- the reference's own conversion, Int::ModSubK1order per wild kangaroo -- taken at the
    same launch boundary: 0 differ.  The file is accepted by the UNMODIFIED program (-winfo, -wcheck) and its kangaroos obey
    (x, y) = d*G / K + d*G (oracle, sampled)."""
    exe = ref_binary("kangaroo_mi355x")
    kernel = _kernel_rate_gks(kng)
    cfg = tmp_path / "in80.txt"
    cfg.write_text(IN80)
    w = tmp_path / "save.work"
    # -m 0.045: stops by itself after ~2^41.11 * 0.045 = 2^36.6 jumps ~ 4 s ... use the clock instead: -m large, killed after the KNG_STATS line
    text = _run([exe, "-t", "0", "-gpu", "-d", "18", "-ws", "-w", str(w), "-wi", "6", "-m", "0.30", str(cfg)], 120,
                env={"KNG_STATS": "1", "KNG_SAVE_VERIFY": "1"}, until="SolveKeyGPU_kng GPU#0: ")
    m = STATS.search(text)
    assert m, text[-3000:]
    launches, wall, mks, lost, saves, save_s = int(m.group(1)), float(m.group(2)), float(m.group(3)), int(m.group(5)), int(m.group(11)), float(m.group(10))
    ver = re.findall(r"SaveWork_kng verify: (\d+) streamed kangaroos, (\d+) differ", text)
    assert saves >= 3 and len(ver) >= 3, text[-3000:]
    assert all(int(a) == 1 << 23 and int(b) == 0 for a, b in ver), ver
    assert lost == 0
    # in verify mode every save point also runs the reference's GetKangaroos (about a second of host conversion with the GPU idle):
    # take it out of the wall time -- the program prints how long its save points took
    assert launches * (1 << 29) / (wall - save_s) / 1e9 >= 0.97 * kernel, (launches, wall, save_s, kernel)
    # the unmodified program reads the file
    ref = ref_binary("kangaroo_hip")
    info = subprocess.run([ref, "-winfo", str(w)], capture_output=True, text=True, timeout=300).stdout
    assert re.search(r"Kangaroos\s*:\s*8388608\b", info) and re.search(r"DP bits\s*:\s*18", info), info
    chk = subprocess.run([ref, "-t", "16", "-wcheck", str(w)], capture_output=True, text=True, timeout=900).stdout
    assert "100.000% OK" in chk, chk[-500:]
    # the kangaroo section itself: sampled records against the oracle's scalar multiplication
    nk = 1 << 23
    rec = np.fromfile(w, dtype=np.uint64, offset=os.path.getsize(w) - 96 * nk).reshape(nk, 12)
    from helpers import N_ORDER, array_to_ints
    kx, ky = 0xBB113592002132E6EF387C3AEBC04667670D4CD40B2103C7D0EE4969E9FF56E4, None
    start = int(IN80.split()[0], 16)
    for i in list(range(0, 40)) + list(range(12345, nk, nk // 60)):
        x, y, d = (array_to_ints(rec[i:i + 1, a:a + 4])[0] for a in (0, 4, 8))
        if i % 2 == 0:  # tame: (start + d) * G   (the program searches the key shifted by -start: Kangaroo.cpp:870-905)
            _, px, py = orc.pubkey(d % N_ORDER)
            assert (px, py) == (x, y), i
        else:
            assert d < N_ORDER and (d < (1 << 80) or d > N_ORDER - (1 << 80)), (i, hex(d))  # |true distance| < 2^79 around 0 mod n


def test_work_file_saves_without_verify_cost_nothing_and_restore_streams(kng, tmp_path):
    """The same run without the verify hook: the rate over the WHOLE wall time, three saves inside, is >= 0.97 of the kernel
    rate; `-i` of that file: FectchKangaroos leaves the GPU thread's records in the file, SolveKeyGPU uploads the bytes and
    unpacks them on the device; the restored run continues the count and saves again; the unmodified program restores OUR
    file and our program restores ITS file."""
    exe, ref = ref_binary("kangaroo_mi355x"), ref_binary("kangaroo_hip")
    kernel = _kernel_rate_gks(kng)
    cfg = tmp_path / "in80.txt"
    cfg.write_text(IN80)
    w1, w2, w3, w4 = (tmp_path / n for n in ("a.work", "b.work", "c.work", "d.work"))
    text = _run([exe, "-t", "0", "-gpu", "-d", "18", "-ws", "-w", str(w1), "-wi", "6", "-m", "0.26", str(cfg)], 120,
                env={"KNG_STATS": "1"}, until="SolveKeyGPU_kng GPU#0: ")
    m = STATS.search(text)
    assert m, text[-3000:]
    saves = int(m.group(11))
    assert saves >= 3 and text.count("done [") >= 3
    assert float(m.group(3)) / 1e3 >= 0.97 * kernel, (m.group(0), kernel)
    sw = re.findall(r"SaveWork_kng: threads at the save point after ([0-9.]+) s, header \+ table ([0-9.]+) s \(table threads released\), (\d+) kangaroos \((\d+) streamed", text)
    assert len(sw) >= 3 and all(int(a) == int(b) == 1 << 23 for _, _, a, b in sw), sw
    c1 = int(re.search(r"Count\s*:\s*(\d+)", subprocess.run([ref, "-winfo", str(w1)], capture_output=True, text=True, timeout=300).stdout).group(1))

    def restored(program, src, dst):
        t = _run([program, "-t", "0", "-gpu", "-d", "18", "-i", str(src), "-ws", "-w", str(dst), "-wi", "6", str(cfg)], 40, env={"KNG_STATS": "1"}, until="done [")
        assert "done [" in t and "2^23.00 kangaroos loaded" in t and "[0 created]" in t, t[-2000:]
        info = subprocess.run([ref, "-winfo", str(dst)], capture_output=True, text=True, timeout=300).stdout
        assert re.search(r"Kangaroos\s*:\s*8388608\b", info), info
        return t, int(re.search(r"Count\s*:\s*(\d+)", info).group(1))

    t2, c2 = restored(exe, w1, w2)       # ours restores ours
    assert "creating kangaroos" not in t2 and "Fetch kangaroos" not in t2  # no FetchWalks for the GPU thread, nothing created
    t3, c3 = restored(ref, w2, w3)       # the unmodified program restores ours
    t4, c4 = restored(exe, w3, w4)       # ours restores the unmodified program's
    assert c1 < c2 < c3 < c4, (c1, c2, c3, c4)
    chk = subprocess.run([ref, "-t", "16", "-wcheck", str(w4)], capture_output=True, text=True, timeout=900).stdout
    assert "100.000% OK" in chk, chk[-500:]


IN64 = IN80  # (name kept by the tests below) the 80-bit interval with a key OUTSIDE it: no test run can end by solving it


def _winfo(ref, path):
    info = subprocess.run([ref, "-winfo", str(path)], capture_output=True, text=True, timeout=300).stdout
    return info, int(re.search(r"Kangaroos\s*:\s*(\d+)", info).group(1)), int(re.search(r"Count\s*:\s*(\d+)", info).group(1))


def test_work_file_of_two_gpu_threads_and_two_cpu_threads(tmp_path):
    """Thread order of the kangaroo section (Backup.cpp:525-536: CPU threads first, then GPU threads): `-t 2 -gpu -gpuId 0,0` -- two
    parked CPU threads written from their arrays, two GPU threads streamed from their snapshots while they walk on.  The
    unmodified program reads the file, restores it into the same thread layout and continues; ours restores the unmodified
    program's file of that layout."""
    exe, ref = ref_binary("kangaroo_mi355x"), ref_binary("kangaroo_hip")
    cfg = tmp_path / "in80b.txt"
    cfg.write_text(IN64)
    base = ["-t", "2", "-gpu", "-gpuId", "0,0", "-g", "32,128,48,128", "-d", "12"]
    nk = 2 * 1024 + (32 + 48) * 128 * 128
    w1, w2, w3 = tmp_path / "a.work", tmp_path / "b.work", tmp_path / "c.work"
    t1 = _run([exe] + base + ["-ws", "-w", str(w1), "-wi", "4", str(cfg)], 30, env={"KNG_STATS": "1", "KNG_SAVE_VERIFY": "1"}, until="done [")
    assert "done [" in t1, t1[-2000:]
    ver = re.findall(r"SaveWork_kng verify: (\d+) streamed kangaroos, (\d+) differ", t1)
    assert ver and int(ver[0][0]) == (32 + 48) * 128 * 128 and int(ver[0][1]) == 0, (ver, t1[-1500:])
    _, k1, c1 = _winfo(ref, w1)
    assert k1 == nk
    chk = subprocess.run([ref, "-t", "8", "-wcheck", str(w1)], capture_output=True, text=True, timeout=600).stdout
    assert "100.000% OK" in chk, chk[-500:]
    t2 = _run([ref] + base + ["-i", str(w1), "-ws", "-w", str(w2), "-wi", "4", str(cfg)], 40, until="done [")
    assert "done [" in t2 and "[0 created]" in t2, t2[-2000:]
    _, k2, c2 = _winfo(ref, w2)
    t3 = _run([exe] + base + ["-i", str(w2), "-ws", "-w", str(w3), "-wi", "4", str(cfg)], 40, env={"KNG_STATS": "1"}, until="done [")
    assert "done [" in t3 and "[0 created]" in t3, t3[-2000:]
    _, k3, c3 = _winfo(ref, w3)
    assert k2 == k3 == nk and c1 < c2 < c3


def test_split_work_files_and_saves_without_kangaroos(tmp_path):
    """`-wsplit`: every save goes to its own file and empties the table (Backup.cpp:470-472, :550-551) -- with the GPU walking on
    and its points of the meantime going into the emptied table; `-w f -wi N` without `-ws`: header + table only, through the
    reference's own SaveWork (delegated), the GPU thread only pausing its table threads for it; KNG_REF_SAVE=1: the reference's own
    save code with the GPU thread filling `Int` arrays as it used to -- all three readable by the unmodified program."""
    exe, ref = ref_binary("kangaroo_mi355x"), ref_binary("kangaroo_hip")
    cfg = tmp_path / "in80b.txt"
    cfg.write_text(IN64)
    base = [exe, "-t", "0", "-gpu", "-g", "64,128", "-d", "12"]
    nk = 64 * 128 * 128
    d = tmp_path / "split"
    d.mkdir()
    t = _run(base + ["-ws", "-wsplit", "-w", str(d / "w"), "-wi", "3", str(cfg)], 14, env={"KNG_STATS": "1"})
    files = sorted(os.listdir(d))
    assert len(files) >= 2 and t.count("done [") >= 2, (files, t[-1500:])
    counts = []
    for f in files[:3]:
        info, k, c = _winfo(ref, d / f)
        assert k == nk
        counts.append(int(re.search(r"DP Count\s*:\s*(\d+)", info).group(1)))
    # each file holds only the points since the save before it: about interval x rate / 2^dp each, not a growing total
    assert max(counts[1:]) < 2.5 * min(counts[1:]) + 1000, counts
    # no -ws: no kangaroos in the file
    w = tmp_path / "table_only.work"
    t = _run(base + ["-w", str(w), "-wi", "3", str(cfg)], 9, env={"KNG_STATS": "1"}, until="done [")
    assert "done [" in t, t[-1500:]
    _, k, _ = _winfo(ref, w)
    assert k == 0
    # the reference's own save path on request
    w = tmp_path / "refsave.work"
    t = _run(base + ["-ws", "-w", str(w), "-wi", "3", str(cfg)], 12, env={"KNG_STATS": "1", "KNG_REF_SAVE": "1"}, until="done [")
    assert "done [" in t and "SaveWork_kng" not in t, t[-1500:]
    _, k, _ = _winfo(ref, w)
    assert k == nk
    chk = subprocess.run([ref, "-t", "8", "-wcheck", str(w)], capture_output=True, text=True, timeout=600).stdout
    assert "100.000% OK" in chk, chk[-500:]
    t = _run(base + ["-i", str(w), "-ws", "-w", str(tmp_path / "again.work"), "-wi", "3", str(cfg)], 20, env={"KNG_REF_SAVE": "1"}, until="done [")
    assert "done [" in t and "Fetch kangaroos" in t, t[-1500:]  # the reference's FetchWalks ran (delegated)


def test_a_refused_snapshot_buffer_falls_back_to_the_arrays(tmp_path):
    """The second device buffer (96 B x herd) can be refused on a device that is nearly full.  The GPU thread then fills the
    reference-shaped `Int` arrays at the same launch boundary and SaveWork_kng writes that thread's section from them: the file
    is complete and the unmodified program checks it (KNG_TEST_FAIL_SNAPSHOT=1 refuses every snapshot)."""
    exe, ref = ref_binary("kangaroo_mi355x"), ref_binary("kangaroo_hip")
    cfg = tmp_path / "in80b.txt"
    cfg.write_text(IN64)
    w = tmp_path / "fallback.work"
    t = _run([exe, "-t", "0", "-gpu", "-g", "64,128", "-d", "12", "-ws", "-w", str(w), "-wi", "3", str(cfg)], 12,
             env={"KNG_STATS": "1", "KNG_TEST_FAIL_SNAPSHOT": "1"}, until="done [")
    assert "done [" in t and "saving through GetKangaroos" in t and "FAILED" not in t, t[-2000:]
    assert re.search(r"SaveWork_kng: .* 1048576 kangaroos \(0 streamed from device snapshots\)", t), t[-1500:]
    _, k, _ = _winfo(ref, w)
    assert k == 64 * 128 * 128
    chk = subprocess.run([ref, "-t", "8", "-wcheck", str(w)], capture_output=True, text=True, timeout=600).stdout
    assert "100.000% OK" in chk, chk[-500:]


def test_key_found_across_three_interrupted_runs(tmp_path):
    """What a work file is FOR: a 76-bit key (known answer) searched by `kangaroo_mi355x -ws -wi 4`, the process KILLED after a
    save, restarted with `-i` (the file's table through LoadTable, the GPU thread's records uploaded and unpacked on the device),
    killed and restarted again, then left to finish.  The key comes out right -- table and kangaroos of every file were those of
    one launch boundary, the restored herd walks on from the saved distances -- and the total count of the last file continues
    the first's.  (The alternating restarts go through both programs: the unmodified one restores ours and vice versa.)"""
    import kangaroo_amd.hostlib as hl

    bits = 76
    start = 0x5A << 100
    key = start + 0x9C3F5A7E2D1B4C6 * 0x31 % (1 << bits) | (1 << (bits - 1))
    _, kx, ky = hl.pubkey(key)
    cfg = tmp_path / "in76.txt"
    cfg.write_text(f"{start:064X}\n{start + (1 << bits) - 1:064X}\n{'02' if ky % 2 == 0 else '03'}{kx:064X}\n")
    exe, ref = ref_binary("kangaroo_mi355x"), ref_binary("kangaroo_hip")
    w = tmp_path / "run.work"
    counts, found = [], None
    # default herd at 76 bits: expected 2^39.1 jumps = ~25 s at the kernel rate; three runs of ~6-7 s each are cut short
    plan = [(exe, None), (ref, w), (exe, w), (exe, w)]
    for n, (program, src) in enumerate(plan):
        last = n == len(plan) - 1
        cmd = [program, "-t", "0", "-gpu", "-d", "16"] + (["-i", str(src)] if src else []) + ["-ws", "-w", str(w), "-wi", "4", str(cfg)]
        t = _run(cmd, 240 if last else 30, env={"KNG_STATS": "1"}, until="Priv: 0x" if last else "done [")
        m = re.search(r"Priv: 0x([0-9A-F]+)", t)
        if m:
            found = int(m.group(1), 16)
            break
        assert "done [" in t, t[-2000:]
        _, k, c = _winfo(ref, w)
        assert k == 1 << 23
        counts.append(c)
    assert found == key, (hex(found) if found else None, hex(key))
    assert counts == sorted(counts) and (len(counts) < 2 or counts[-1] > counts[0]), counts
