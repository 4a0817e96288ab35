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
