"""Soak run of the host pipeline: two engines on one device, a save every two seconds for 50 s (every fifth with the
herds), each snapshot re-read and checked.  usage: python tools/solver_soak.py"""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from kangaroo_amd import hostlib as hl, solver as sv
start = 0x9900000000000000000000
kxy = hl.pubkey(start + 0x3F5A5A5A5A5A5A5A5A5A5)[1:]
s = sv.Solver(start, start + (1 << 86) - 1, kxy, gpus=(0, 0), grid=(128, 128), dp=13, seed=77)
s.start()
t0 = time.time(); prev = 0; i = 0
while time.time() - t0 < 50:
    rc = s.wait(2.0)
    assert rc == 0, rc
    s.save("/tmp/soak.work", with_kangaroos=(i % 5 == 4))
    t = sv.DpTable(); h, nk, _ = sv.read_workfile("/tmp/soak.work", t, with_kangaroos=False)
    assert h["count"] > prev and h["count"] % (128 * 128 * 128 * 64) == 0
    prev = h["count"]; i += 1; t.close()
s.stop(); st = s.stats()
assert st["table_items"] == st["dps"] - st["same_herd"] and st["dps_lost"] == 0 and st["wrong_collisions"] == 0
print(f"soak ok: {i} saves, {st['launches']} launches, 2^{__import__('math').log2(st['jumps']):.2f} jumps, {st['dps']} DPs, {st['jumps']/st['seconds']/1e9:.2f} GK/s incl. pauses")
s.close()
