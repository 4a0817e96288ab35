"""First contact with an N-GPU node, as far as it can be tested without one: where GPU threads, pinned DP
rings and table threads are placed (sysfs NUMA topology, read through KNG_SYSFS_ROOT so that a made-up tree can stand in),
what happens when the kernel refuses an affinity mask, and how HIP device indices are tied to rocm_smi's (PCI address)."""
import ctypes as C
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tree(tmp_path, nodes, pci=None, siblings=None):
    """nodes: {id: cpulist}; pci: {bdf: numa_node text}; siblings: {cpu: list text}"""
    root = tmp_path / "sys"
    for nid, cpus in nodes.items():
        d = root / "devices" / "system" / "node" / f"node{nid}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus + "\n")
    for bdf, node in (pci or {}).items():
        d = root / "bus" / "pci" / "devices" / bdf
        d.mkdir(parents=True)
        (d / "numa_node").write_text(node + "\n")
    for cpu, sib in (siblings or {}).items():
        d = root / "devices" / "system" / "cpu" / f"cpu{cpu}" / "topology"
        d.mkdir(parents=True)
        (d / "thread_siblings_list").write_text(sib + "\n")
    return str(root)


def _host():
    from kangaroo_amd import hostlib

    L = hostlib.load()
    L.kngs_plan_placement.argtypes = [C.c_int, C.c_int, C.c_uint, C.c_char_p, C.c_size_t]
    L.kngs_gpu_thread_cpus.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
    L.kngs_try_pin.argtypes = [C.c_char_p]
    L.kngs_pin_failures.restype = C.c_uint64
    return L


def _plan(L, n, per_core=0, salt=0):
    buf = C.create_string_buffer(1 << 16)
    n_ids = L.kngs_plan_placement(n, per_core, salt, buf, len(buf))
    return n_ids, buf.value.decode().splitlines()


def _gpu_cpus(L, node):
    buf = C.create_string_buffer(4096)
    rc = L.kngs_gpu_thread_cpus(node, buf, len(buf))
    return rc, buf.value.decode()


@pytest.fixture
def allowed():
    cpus = sorted(os.sched_getaffinity(0))
    if len(cpus) < 4:
        pytest.skip("needs 4 CPUs in the affinity mask")
    return cpus


def _cpulist(cpus):
    return ",".join(str(c) for c in cpus)


def test_pci_address_to_numa_node(tmp_path, monkeypatch):
    import kangaroo_amd

    lib = kangaroo_amd.load_library()
    lib.kng_numa_node_of_bdf.argtypes = [C.c_char_p]
    root = _tree(tmp_path, {0: "0"}, pci={"0000:c3:00.0": "2", "0000:05:00.0": "-1", "0001:0a:00.0": "garbage"})
    monkeypatch.setenv("KNG_SYSFS_ROOT", root)
    assert lib.kng_numa_node_of_bdf(b"0000:C3:00.0") == 2      # HIP spells hex digits in upper case, sysfs in lower
    assert lib.kng_numa_node_of_bdf(b"0000:c3:00.0") == 2
    assert lib.kng_numa_node_of_bdf(b"0000:05:00.0") == -1     # "-1": the platform does not say
    assert lib.kng_numa_node_of_bdf(b"0001:0a:00.0") == -1     # unreadable content
    assert lib.kng_numa_node_of_bdf(b"0000:ff:00.0") == -1     # no such device
    assert lib.kng_numa_node_of_bdf(b"") == -1 and lib.kng_numa_node_of_bdf(None) == -1
    assert lib.kng_numa_node_of_bdf(b"x" * 200) == -1


def test_nodes_are_looked_up_by_their_real_id(tmp_path, monkeypatch, allowed):
    """Sparse ids (0 and 2), a memory-only node (3), CPUs outside the affinity mask: the CPU set of node 2 is at index 2,
    a GPU on node 2 is confined to node 2's CPUs, a GPU on a node without usable CPUs is not confined at all."""
    lo, hi = allowed[: len(allowed) // 2], allowed[len(allowed) // 2:]
    root = _tree(tmp_path, {0: _cpulist(lo), 2: _cpulist(hi) + ",4000-4007", 3: ""})
    monkeypatch.setenv("KNG_SYSFS_ROOT", root)
    monkeypatch.delenv("KNGS_PIN", raising=False)
    L = _host()
    n_ids, lines = _plan(L, 4)
    assert n_ids == 4
    node_lines = [ln for ln in lines if ln.startswith("node ")]
    assert node_lines[1] == "node 1: -" and node_lines[3] == "node 3: -"
    assert node_lines[0].startswith("node 0: ") and node_lines[2].startswith("node 2: ")
    set0, set2 = node_lines[0].split(": ")[1], node_lines[2].split(": ")[1]
    cons = [ln for ln in lines if ln.startswith("consumer ")]
    assert cons == [f"consumer 0: node 0 cpus {set0}", f"consumer 1: node 0 cpus {set0}",
                    f"consumer 2: node 2 cpus {set2}", f"consumer 3: node 2 cpus {set2}"]
    assert "4000" not in set2                                    # CPUs the process may not use never enter a mask
    assert _gpu_cpus(L, 2) == (1, set2) and _gpu_cpus(L, 0) == (1, set0)
    for node in (1, 3, 7, -1):
        assert _gpu_cpus(L, node) == (0, "unconfined")


def test_single_node_machine_is_left_alone_by_default(tmp_path, monkeypatch, allowed):
    root = _tree(tmp_path, {0: _cpulist(allowed)})
    monkeypatch.setenv("KNG_SYSFS_ROOT", root)
    L = _host()
    _, lines = _plan(L, 3)
    assert [ln for ln in lines if ln.startswith("consumer ")] == [f"consumer {c}: node 0 unconfined" for c in range(3)]
    assert _gpu_cpus(L, 0) == (0, "unconfined")


def test_affinity_mask_that_excludes_a_node(tmp_path, allowed):
    """Under `taskset` to the first node's CPUs the second node has nothing this process may use: one usable node, nobody is
    confined, and a GPU that hangs off the excluded node is not pinned to an empty set."""
    lo, hi = allowed[: len(allowed) // 2], allowed[len(allowed) // 2:]
    root = _tree(tmp_path, {0: _cpulist(lo), 1: _cpulist(hi)})
    code = (
        "import os, sys, ctypes as C\n"
        f"sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})\n"
        f"os.sched_setaffinity(0, {set(lo)!r})\n"
        "import test_placement_cpu as t\n"
        "L = t._host()\n"
        "print(t._plan(L, 2)); print(t._gpu_cpus(L, 1)); print(t._gpu_cpus(L, 0))\n")
    env = dict(os.environ, KNG_SYSFS_ROOT=root)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr[-1500:]
    lines = out.stdout.splitlines()
    assert "'node 1: -'" in lines[0] and "consumer 0: node 0 unconfined" in lines[0] and "consumer 1: node 0 unconfined" in lines[0]
    assert lines[1] == "(0, 'unconfined')" and lines[2] == "(0, 'unconfined')"


def test_one_core_per_table_thread_is_opt_in_and_rotates(tmp_path, monkeypatch, allowed):
    """KNGS_PIN=core (per_core=1): one physical core per thread, distinct within a node, hyper-thread siblings skipped, and the
    choice depends on the salt (solver instance + pid) so that two solvers of one host do not stack on the same cores."""
    if len(allowed) < 8:
        pytest.skip("needs 8 CPUs")
    cpus = allowed[:8]
    sib = {}
    for a, b in zip(cpus[0::2], cpus[1::2]):     # pairs of hardware threads
        sib[a] = sib[b] = f"{a},{b}"
    root = _tree(tmp_path, {0: _cpulist(cpus[:4]), 1: _cpulist(cpus[4:])}, siblings=sib)
    monkeypatch.setenv("KNG_SYSFS_ROOT", root)
    L = _host()
    _, lines = _plan(L, 2, per_core=1, salt=0)
    cons = [ln for ln in lines if ln.startswith("consumer ")]
    assert cons == [f"consumer 0: node 0 cpus {cpus[0]}", f"consumer 1: node 1 cpus {cpus[4]}"]
    _, lines1 = _plan(L, 2, per_core=1, salt=1)
    cons1 = [ln for ln in lines1 if ln.startswith("consumer ")]
    assert cons1 == [f"consumer 0: node 0 cpus {cpus[2]}", f"consumer 1: node 1 cpus {cpus[6]}"]   # the other core of each node
    # more threads than cores on a node: the node's whole set, no single-CPU masks
    _, lines4 = _plan(L, 6, per_core=1, salt=0)
    assert all("cpus " in ln and "," in ln.split("cpus ")[1] or "-" in ln.split("cpus ")[1] for ln in lines4 if ln.startswith("consumer "))


def test_a_refused_affinity_mask_is_counted_not_fatal():
    L = _host()
    before = L.kngs_pin_failures()
    mask0 = os.sched_getaffinity(0)
    assert L.kngs_try_pin(b"900") == 0            # a CPU this machine does not have: sched_setaffinity says EINVAL
    assert L.kngs_try_pin(b"") == 0               # an empty set is never handed to the kernel
    assert L.kngs_pin_failures() == before + 2
    assert os.sched_getaffinity(0) == mask0       # the thread keeps running where it was
    assert L.kngs_try_pin(str(min(mask0)).encode()) == 1
    assert os.sched_getaffinity(0) == mask0


def test_hip_index_to_rocm_smi_index_goes_through_the_pci_address():
    from kangaroo_amd import telemetry as t

    assert t.bdf_of_smi_id((0 << 32) | (0xC3 << 8) | (0 << 3) | 0) == "0000:c3:00.0"
    assert t.bdf_of_smi_id((1 << 32) | (2 << 28) | (0x0A << 8) | (3 << 3) | 1) == "0001:0a:03.1"   # partition bits ignored
    smi = ["0000:05:00.0", "0000:26:00.0", "0000:c3:00.0", "0000:e3:00.0"]
    # HIP_VISIBLE_DEVICES=2,0: HIP 0 is the third card, HIP 1 the first
    assert t.map_devices({0: "0000:c3:00.0", 1: "0000:05:00.0"}, smi) == {0: (2, "pci"), 1: (0, "pci")}
    assert t.map_devices({0: None}, smi) == {0: (0, "index (assumed)")}                  # address unreadable: say it is a guess
    assert t.map_devices({5: None}, smi)[5][0] is None
    assert t.map_devices({0: "0000:aa:00.0"}, smi)[0][0] is None                          # a card rocm_smi does not list
    assert t.map_devices({0: "0000:c3:00.0"}, [None, None]) == {0: (0, "index (assumed)")}


def test_sampler_survives_a_library_without_the_power_symbols():
    """a librocm_smi64 that lacks a symbol must give None, not kill the sampler thread."""
    from kangaroo_amd import telemetry as t

    class Bare:  # no rsmi_* attributes at all
        pass

    assert t._power_w(Bare(), 0) is None and t._sclk_mhz(Bare(), 0) is None and t._energy_j(Bare(), 0) is None
    assert t.smi_bdfs(Bare()) == []


def test_table_threads_of_the_reference_program_are_spread_over_the_nodes(tmp_path, allowed):
    """Round 6: the pool of owner-partitioned table threads behind SolveKeyGPU_kng.cpp (kng_ingest.h) uses the same placement
    as the repo's solver: on a made-up two-node tree (this process's CPUs cut in two) the pool says two nodes, on a one-node
    tree and with KNG_TABLE_PIN=0 it confines nothing -- and the points all arrive either way."""
    from helpers import ref_binary

    exe = ref_binary("poolbench")
    half = len(allowed) // 2
    two = _tree(tmp_path / "two", {0: _cpulist(allowed[:half]), 1: _cpulist(allowed[half:])})
    one = _tree(tmp_path / "one", {0: _cpulist(allowed)})

    def run(root, **env):
        e = dict(os.environ, KNG_SYSFS_ROOT=root, **env)
        out = subprocess.run([exe, "200000", "100000", "4", "2", "20000"], capture_output=True, text=True, timeout=300, env=e)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "# 200000 entries in " in out.stdout, out.stdout
        return out.stdout

    assert "table threads spread over 2 NUMA node(s)" in run(two)
    assert "table threads spread over 0 NUMA node(s) (not confined)" in run(one)
    assert "table threads spread over 0 NUMA node(s) (not confined)" in run(two, KNG_TABLE_PIN="0")
    assert "table threads spread over 2 NUMA node(s)" in run(two, KNG_TABLE_PIN="core")
