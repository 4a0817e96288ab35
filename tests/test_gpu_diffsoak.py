"""Launch-by-launch differential between independent walk code paths on the device (tools/diff_soak.py):
every DP record of every launch and, periodically, every (x, y, d) of the herd.  The deep runs (2^40 jumps per engine) are
under profiles/r05_diffsoak_*.json; here 2^36 jumps per engine at the bench herd."""
import argparse
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


def _args(**kw):
    base = dict(variant="asm", launches=128, grid=(512, 128), range_power=80, dp=14, state_every=64, seed=0x50AC, device=0, out=None)
    base.update(kw)
    return argparse.Namespace(**base)


def test_scheduled_loop_equals_compiler_loop_for_2_36_jumps(kng):
    """asm 1 against asm 0, 2^23 kangaroos, 128 launches = 2^36 jumps each: 4.2 million DP records compared one launch at a
    time (a dropped or duplicated record, a wrong kidx or distance shows up in the launch it happens in), the whole herd three
    times."""
    import diff_soak

    r = diff_soak.run(_args())
    assert r["clean"], r
    assert r["jumps_per_engine"] == 128 * (1 << 23) * 64 and r["dp_records_compared"] > 4_000_000
    assert r["state_compares"] >= 3 and r["exact_exits"][1] == 0   # only the scheduled loop has an exact path to leave to


def test_low_word_distance_streaming_equals_full_distances_at_109_bits(kng):
    """dsplit 1 against dsplit 0 at 109 bits (jump distances ~2^55: a carry out of the low word every ~500 jumps per kangaroo),
    2^21 kangaroos x 256 launches: the L2-atomic carry path against plain 128-bit adds."""
    import diff_soak

    r = diff_soak.run(_args(variant="dsplit", launches=256, grid=(128, 128), range_power=109, dp=12, state_every=128))
    assert r["clean"], r
    assert r["dsplit_in_effect"] == [1, 0]


def test_one_level_inversion_tree_equals_two_level_tree(kng):
    """share 4 (256-thread blocks, new in round 5 for herds too small to fill the chip) against share 8 on the same herd of
    2^18 kangaroos (group 2: 131 072 lanes, a size both forms walk), 512 launches = 2^33 jumps each, DP 8: 33 million records
    per engine compared launch by launch, the herd nine times."""
    import diff_soak

    r = diff_soak.run(_args(variant="share", launches=512, grid=(512, 4), range_power=80, dp=8, state_every=128))
    assert r["clean"], r
    assert r["share_in_effect"] == [4, 8] and r["dp_records_compared"] > 30_000_000
