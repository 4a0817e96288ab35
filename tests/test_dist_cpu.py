"""N>1 path on CPU: world size 2 over gloo.  The data path has no collective (independent herds),
so what must be right is the plumbing: rank -> device/seed, barrier-bracketed timing, max over ranks,
whole-job aggregation, auto-DP from the total kangaroo count."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    from kangaroo_amd.dist import Ranks, timed_steps, whole_job_rate
    from kangaroo_amd import hostlib as hl
    r = Ranks(backend="gloo")
    # a fake engine step: rank 1 is twice as slow as rank 0
    per_step = 0.02 * (r.rank + 1)
    def step(i): time.sleep(per_step)
    elapsed = timed_steps(r, step, lambda: None, steps=5)
    n = 512 * 128 * 128
    out = {"rank": r.rank, "device": r.device, "world": r.world, "seed": r.herd_seed(0xBEEF), "elapsed": elapsed,
           "rate": whole_job_rate(r, n * 64, 5, elapsed), "dp": hl.suggest_dp(80, r.total_kangaroos(n))}
    # N > 1 bench protocol: rank 0 runs the whole job (all GPUs, one host table), the others bracket it with barriers
    from kangaroo_amd.dist import timed_on_rank0
    ran = []
    out["job_elapsed"] = timed_on_rank0(r, (lambda: (time.sleep(0.15), ran.append(1))) if r.rank == 0 else None)
    out["ran"] = len(ran)
    print("RESULT " + json.dumps(out), flush=True)
    r.close()
""") % ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_gloo(tmp_path):
    from kangaroo_amd.build import build_all

    build_all()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    import json

    res = []
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0, err[-2000:]
        line = [l for l in out.splitlines() if l.startswith("RESULT ")][0]
        res.append(json.loads(line[7:]))
    res.sort(key=lambda r: r["rank"])
    assert [r["device"] for r in res] == [0, 1] and all(r["world"] == 2 for r in res)
    assert res[0]["seed"] != res[1]["seed"]
    # both ranks report the SAME elapsed: the max over ranks (rank 1 needs >= 5 * 40 ms)
    assert abs(res[0]["elapsed"] - res[1]["elapsed"]) < 1e-9 and res[0]["elapsed"] >= 0.19
    n = 512 * 128 * 128
    assert abs(res[0]["rate"] - 2 * n * 64 * 5 / res[0]["elapsed"]) < 1e-3
    # auto DP from the TOTAL kangaroo count (Kangaroo.cpp:980-988): 2 x 2^23 on 80 bits -> 13
    assert res[0]["dp"] == res[1]["dp"] == 13
    # the single-process job of rank 0 is seen with the same duration by every rank
    assert [r["ran"] for r in res] == [1, 0]
    assert abs(res[0]["job_elapsed"] - res[1]["job_elapsed"]) < 1e-9 and 0.15 <= res[0]["job_elapsed"] < 1.0


def test_single_rank_needs_no_torch_distributed():
    from kangaroo_amd.dist import Ranks, timed_steps, whole_job_rate

    env_backup = {k: os.environ.pop(k, None) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    try:
        r = Ranks(backend="gloo")
        assert (r.rank, r.world, r.device) == (0, 1, 0) and r.dist is None
        calls = []
        el = timed_steps(r, calls.append, lambda: calls.append("fin"), steps=3)
        assert calls == [0, 1, 2, "fin"] and el > 0
        assert whole_job_rate(r, 10, 3, 2.0) == 15.0
    finally:
        for k, v in env_backup.items():
            if v is not None:
                os.environ[k] = v


FAIL_WORKER = textwrap.dedent("""
    import os, sys, time
    sys.path.insert(0, %r)
    from kangaroo_amd.dist import Ranks, RankFailure, timed_on_rank0
    r = Ranks(backend="gloo")
    mode = os.environ["KNG_FAIL_MODE"]
    if mode == "prepare":
        # bench_multi's preparation protocol: rank 0 fails BEFORE the timed region, says so through all_ok(), everyone leaves
        err = RuntimeError("rank 0 sees 1 HIP device, needs 2") if r.rank == 0 else None
        if not r.all_ok(err is None):
            r.abort()
            sys.exit(3)
        sys.exit(0)
    def job():
        time.sleep(0.05)
        raise RuntimeError("engine failed in launch 3")
    try:
        timed_on_rank0(r, job if r.rank == 0 else None)
    except RankFailure as e:
        print("RANKFAILURE", r.rank, e, flush=True)
        r.abort()
        sys.exit(3)
    sys.exit(0)
""") % ROOT


def _run_two_ranks(script, extra_env, timeout, args=()):
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), **extra_env)
        procs.append(subprocess.Popen([sys.executable, str(script), *args], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            out, err = p.communicate(timeout=timeout)
            outs.append((p.returncode, out, err))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return outs


def test_rank0_failure_ends_every_rank(tmp_path):
    """when rank 0's job (the only rank that works) fails, BOTH ranks must leave non-zero promptly --
    nobody may sit in a barrier until the launcher's timeout.  Covers the timed region and the preparation before it."""
    import time

    script = tmp_path / "fail_worker.py"
    script.write_text(FAIL_WORKER)
    for mode in ("job", "prepare"):
        t0 = time.time()
        outs = _run_two_ranks(script, {"KNG_FAIL_MODE": mode}, timeout=120)
        took = time.time() - t0
        assert [rc for rc, _, _ in outs] == [3, 3], (mode, outs)
        assert took < 90, f"{mode}: {took:.0f} s"
        if mode == "job":
            assert all("RANKFAILURE" in out for _, out, _ in outs), outs
            assert "engine failed in launch 3" in outs[0][1]  # rank 0 knows why; rank 1 knows that


def test_bench_two_ranks_without_devices_ends_promptly():
    """`bench.py --gpus 2` as the driver launches it (one rank per GPU, gloo control plane) on a box where NO rank has a
    device: the visibility vote sends both ranks to the per-rank form, the device check fails collectively, and both leave
    with status 1 -- no rank is left in a barrier."""
    import time

    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("needs a box without GPUs")
    t0 = time.time()
    outs = _run_two_ranks(os.path.join(ROOT, "bench.py"), {}, timeout=240, args=("--gpus", "2", "--steps", "2", "--warmup", "1"))
    assert [rc for rc, _, _ in outs] == [1, 1], outs
    assert time.time() - t0 < 200
    assert "sees 0 of 2 devices" in outs[0][2] or "sees 0 HIP device" in outs[0][2], outs[0][2][-600:]
