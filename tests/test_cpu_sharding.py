"""Read sharding over ranks: partition properties, and a world_size-2 gloo run of the counter
all-reduce with per-rank work done by the oracle standing in for the GPU."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def test_shard_tasks_partition_and_balance():
    from mm2chain import sharding
    rng = np.random.default_rng(0)
    sizes = rng.integers(100, 9000, 1000)
    for ws in (1, 2, 4, 8):
        parts = [sharding.shard_tasks(sizes, ws, r) for r in range(ws)]
        allt = np.sort(np.concatenate(parts))
        assert np.array_equal(allt, np.arange(1000))
        loads = np.array([sizes[p].sum() for p in parts])
        assert loads.max() - loads.min() <= sizes.max()
    eq = np.full(1000, 5000)
    parts = [sharding.shard_tasks(eq, 8, r) for r in range(8)]
    assert all(len(p) == 125 for p in parts)
    off = np.concatenate(([0], np.cumsum(sizes)))
    new_off, starts, ns = sharding.sub_batch(off, parts[3][:10])
    assert new_off[-1] == ns.sum() and np.array_equal(starts, off[:-1][parts[3][:10]])


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "minimap2-fpga_amd")); sys.path.insert(0, os.path.join(root, "tests"))
    from mm2chain import sharding, synth, params
    import oracle_binding as ob
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    off, a = synth.make_stream("mixed", 24, (200, 900), seed=3)     # same stream on every rank (deterministic)
    off = off.numpy(); a = a.numpy().view(np.uint64)
    mine = sharding.shard_tasks(off[1:] - off[:-1], world, rank)
    new_off, starts, ns = sharding.sub_batch(off, mine)
    idx = np.concatenate([np.arange(s, s + n) for s, n in zip(starts, ns)]) if len(mine) else np.zeros(0, np.int64)
    f, p, secs = ob.chain_batch(params.map_ont(), new_off, a[idx], 1)
    tot, _, mx = sharding.allreduce_counters(int(ns.sum()), 0, int(secs * 1e9))
    q.put((rank, mine.tolist(), int(f.astype(np.int64).sum()), tot, mx))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_shards_cover_the_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs: p.join(60)
    assert all(p.exitcode == 0 for p in procs)
    res.sort()
    from mm2chain import synth, params
    import oracle_binding as ob
    off, a = synth.make_stream("mixed", 24, (200, 900), seed=3)
    f, _, _ = ob.chain_batch(params.map_ont(), off.numpy(), a.numpy().view(np.uint64), 1)
    assert sorted(res[0][1] + res[1][1]) == list(range(24))
    assert res[0][3] == res[1][3] == int(off[-1])                      # all-reduced anchor count = whole batch
    assert res[0][2] + res[1][2] == int(f.astype(np.int64).sum())      # the two shards together reproduce the whole
    assert res[0][4] == res[1][4] > 0


def test_bench_gpus_n_starts_n_ranks():
    """`python bench.py --gpus 2` with no launcher around it must become two ranks (launch_ranks: fresh child processes, RANK / WORLD_SIZE /
    MASTER_* set, rendezvous on 127.0.0.1) and print n_gpus 2.  No GPU here: MM2C_BENCH_REHEARSE_NO_GPU makes each rank do the rendezvous, the
    counter all-reduce and the per-rank gather only; the same flow with real work runs on the GPU box (test_gpu_parity.py)."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MM2C_BENCH_REHEARSE_NO_GPU"] = "1"
    for n in (2, 1):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout
        out = json.loads(lines[0])
        assert out["n_gpus"] == n and out["world_size_seen"] == n and out["requested_gpus"] == n
        assert out["counters_sum"] == 1000 * n * (n + 1) // 2 and out["per_rank_ns"] == [1000 + k for k in range(n)] and out["max_ns"] == 1000 + n - 1


def _bench_env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_bench_refuses_a_run_whose_ranks_are_not_the_gpus_asked_for(tmp_path):
    """The multi-GPU line must not carry a figure when (a) the process group is not --gpus ranks, (b) two ranks report the same PCI bus id
    (they share a card), or (c) the node shows fewer GPUs than asked for -- counted from the KFD topology in sysfs, so the launcher parent never
    starts a GPU runtime (cf. the reference's device enumeration before it makes its queues, chain_hardware.cpp:278-330).  No GPU here: the
    rehearsal knob makes each rank do the rendezvous and the gathers only, with made-up bus ids."""
    import json, subprocess, sys
    from mm2chain import sharding
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ids = lambda *b: [{"rank": k, "ordinal": k, "pci_bus_id": x, "arch": "gfx950"} for k, x in enumerate(b)]
    assert sharding.placement_problems(2, 2, ids("0000:05:00.0", "0000:15:00.0")) == []
    assert any("both run on" in m for m in sharding.placement_problems(2, 2, ids("0000:05:00.0", "0000:05:00.0")))
    assert sharding.placement_problems(2, 2, ids("0000:05:00.0", "0000:05:00.0"), shared_device_ok=True) == []
    assert any("--gpus 8" in m for m in sharding.placement_problems(8, 2, ids("a", "b")))
    assert any("identities" in m for m in sharding.placement_problems(2, 2, ids("a")))
    bench = [sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"]
    # (b) two ranks on one card: no line, code 3; allowed only under the one-device rehearsal knob
    r = subprocess.run(bench + ["--gpus", "2"], env=_bench_env(MM2C_BENCH_REHEARSE_NO_GPU="1", MM2C_BENCH_FAKE_BUS_IDS="0000:05:00.0,0000:05:00.0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 3 and "both run on" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")], (r.returncode, r.stderr[-1500:])
    r = subprocess.run(bench + ["--gpus", "2"], env=_bench_env(MM2C_BENCH_REHEARSE_NO_GPU="1", MM2C_BENCH_FAKE_BUS_IDS="0000:05:00.0,0000:05:00.0", MM2C_BENCH_ONE_DEVICE="1"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert [d["rank"] for d in out["devices"]] == [0, 1] and out["devices"][0]["pci_bus_id"] == "0000:05:00.0"
    # (a) a launcher that started one rank for --gpus 2
    r = subprocess.run(bench + ["--gpus", "2"], env=_bench_env(MM2C_BENCH_REHEARSE_NO_GPU="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 3 and "--gpus 2" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")], (r.returncode, r.stderr[-1500:])
    # (c) the node count comes from the KFD topology: a made-up tree with one CPU node and two GPU nodes
    for k, simd in enumerate((0, 256, 256)):
        d = tmp_path / "nodes" / str(k); d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count {16 if simd == 0 else 0}\nsimd_count {simd}\nmem_banks_count 1\n")
    sys.path.insert(0, root)
    import bench as bench_mod
    assert bench_mod.count_gpu_nodes(str(tmp_path / "nodes")) == 2 and bench_mod.count_gpu_nodes(str(tmp_path / "absent")) is None
    r = subprocess.run(bench + ["--gpus", "3"], env=_bench_env(MM2C_BENCH_KFD_NODES=str(tmp_path / "nodes")), capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "shows 2 GPU" in r.stderr, (r.returncode, r.stderr[-1500:])


def test_in_process_device_split_covers_and_balances():
    """mm2c_split_tasks (the split the host-batch entries use when mm2c_init_devices configured several devices; cf. the reference's
    per-kernel queue scaffolding chain_hardware.cpp:9-23): with a fake device count, every task lands in exactly one contiguous range, in
    order, and no range exceeds its share of the anchors by more than its last task.  Pure host logic: no GPU is touched."""
    import mm2chain
    rng = np.random.default_rng(3)
    for n_tasks, n_parts in [(1000, 8), (7, 8), (1, 3), (0, 4), (5000, 2), (64, 64)]:
        sizes = rng.integers(0, 9000, n_tasks)
        if n_tasks > 10:
            sizes[rng.integers(0, n_tasks, 3)] = 300000          # a few very long reads
        off = np.concatenate([[17], 17 + np.cumsum(sizes)]).astype(np.int64)      # offsets need not start at 0
        b = mm2chain.split_tasks(off, n_parts)
        assert b[0] == 0 and b[-1] == n_tasks and np.all(np.diff(b) >= 0)
        total = int(off[-1] - off[0])
        for s in range(n_parts):
            part = int(off[b[s + 1]] - off[b[s]])
            last = int(sizes[b[s + 1] - 1]) if b[s + 1] > b[s] else 0
            assert part <= total / n_parts + last + 1, (n_tasks, n_parts, s, part)
        # ranges end at the first task boundary at or beyond their share
        for s in range(1, n_parts):
            assert int(off[b[s]] - off[0]) >= total * s // n_parts or b[s] == n_tasks


def test_per_read_calls_are_routed_to_the_least_loaded_device_slot():
    """mm2c_route_slot, the rule by which a per-read call (run_chaining_on_hw / mm_chain_dp, one blocking call per read) picks a device's call combiner -- the
    reference picks one of its kernels per call (chain_hardware.cpp:58-72).  Fake device counts, no GPU: the least loaded slot wins, idle slots are taken in turn
    from tid % n, and a simulated stream of calls (each stays for a time proportional to its anchors) spreads its anchors evenly."""
    import ctypes as C
    from mm2chain import _native as N
    lib = N.load()

    def route(out, tid):
        a = np.ascontiguousarray(out, dtype=np.int64)
        return lib.mm2c_route_slot(a.size, a.ctypes.data_as(C.c_void_p), tid)

    assert route([0], 5) == 0 and route([7, 7, 7, 7], 6) == 2 and route([7, 7, 7, 7], -3) == 0
    assert route([5, 0, 9, 0], 0) == 1 and route([5, 0, 9, 0], 2) == 3 and route([5, 0, 9, 0], 3) == 3
    assert route([1, 2, 3, 0, 4, 5, 6, 7], 11) == 3
    assert lib.mm2c_route_slot(1, None, 4) == 0 and lib.mm2c_route_slot(8, None, 4) == 0
    rng = np.random.default_rng(8)
    for n_slots in (2, 4, 8):
        out = np.zeros(n_slots, np.int64)
        served = np.zeros(n_slots, np.int64)
        inside = []                                                 # (leaves at, slot, anchors)
        now = 0
        for call in range(20000):
            n = int(rng.integers(200, 20000))
            now += int(rng.integers(0, 3000))
            for t, s, m in [c for c in inside if c[0] <= now]:
                out[s] -= m
            inside = [c for c in inside if c[0] > now]
            s = route(out, call % 16)
            out[s] += n; served[s] += n
            inside.append((now + n, s, n))
        assert served.min() > 0.8 * served.mean(), (n_slots, served)


def test_device_to_cpu_set_mapping_reads_the_numa_node_of_the_pci_device(tmp_path):
    """Host feed of several GPUs from one process: the worker thread of a device slot is pinned to the CPUs of the NUMA node its device hangs off
    (mm2c_numa_cpulist: <sysfs>/bus/pci/devices/<bus id>/numa_node -> <sysfs>/devices/system/node/node<N>/cpulist).  A made-up sysfs tree: two sockets,
    a device on each, one device without NUMA information (-1, what a single-socket box or a container shows) and one that is not there."""
    import ctypes as C
    from mm2chain import _native as N
    lib = N.load()
    root = tmp_path / "sys"
    for bus, node in (("0000:05:00.0", 0), ("0000:85:00.0", 1), ("0000:f4:00.0", -1)):
        d = root / "bus" / "pci" / "devices" / bus
        d.mkdir(parents=True)
        (d / "numa_node").write_text(f"{node}\n")
    for node, cpus in ((0, "0-47,96-143"), (1, "48-95,144-191")):
        d = root / "devices" / "system" / "node" / f"node{node}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus + "\n")

    def lookup(bus):
        buf = C.create_string_buffer(256)
        return lib.mm2c_numa_cpulist(str(root).encode(), bus.encode(), buf, 256), buf.value.decode()

    assert lookup("0000:05:00.0") == (0, "0-47,96-143")
    assert lookup("0000:85:00.0") == (1, "48-95,144-191")
    assert lookup("0000:f4:00.0")[0] == -1                       # no NUMA information: the worker stays where the scheduler puts it
    assert lookup("0000:99:00.0")[0] == -1                       # no such device
    (root / "bus" / "pci" / "devices" / "0000:05:00.0" / "numa_node").write_text("7\n")
    assert lookup("0000:05:00.0")[0] == -1                       # a node without a cpulist file
    assert lib.mm2c_numa_cpulist(None, b"x", C.create_string_buffer(8), 8) == -1
    assert lib.mm2c_slot_worker_node(0) == -1 and lib.mm2c_slot_worker_node(99) == -1     # no worker has been started in this process


def test_route_of_a_batch_between_the_two_dp_kernels():
    """Round 6: which DP kernel a batch takes under the library's default ("coop_plans" 2) -- pure arithmetic (mm2c_route_pieces = coop_pays, csrc/chain_kernel.h), the same
    rule on the host (plans whose tasks run as they are, host passes and chunks) and on the device (chain_route, after long tasks have been cut).  Few long pieces get
    sixteen waves each (eight when there are more pieces than CUs), the analogue of the reference's one deep pipeline per task (device/minimap2_opencl.cl:49,71); anything else one wave each."""
    from mm2chain import _native as N
    lib = N.load()
    r = lib.mm2c_route_pieces
    assert r(1, 5000, 5000) == 16                                   # a lone per-read call
    assert r(255, 1000000, 255000000) == 16 and r(256, 10**6, 256 * 10**6) == 16      # at most one piece per CU: sixteen waves, one workgroup per CU
    assert r(1020, 300000, 306000000) == 8 and r(510, 500000, 255000000) == 8         # more pieces than CUs: eight waves, two workgroups per CU (129.3 -> 111.3 ms)
    assert r(2048, 100000, 204800000) == 8 and r(1533, 150000, 1533 * 150000) == 8    # long pieces: up to 2 048 equal ones (80.1 / 77.4 ms, 117.8 / 85.1)
    assert r(2048, 3000, 2048 * 3000) == 1 and r(2048, 8191, 2048 * 8191) == 1        # ... short ones only under the round's first rule (3.2 / 5.8 ms)
    assert r(1024, 1000, 1024000) == 8                              # (that rule: 1450 * longest > total)
    assert r(65536, 5000, 327680000) == 1 and r(2049, 10**6, 2049 * 10**6) == 1       # never above 2 048 pieces: the GPU is full of waves anyway
    assert r(2000, 300, 600000) == 1 and r(100, 300, 30000) == 16   # short pieces: only when there are few of them
    assert r(0, 0, 0) == 1
    assert r(40, 2500, 100000) == 16                                # tests/test_gpu_long_reads.py: 8 reads x 5 loci cut on the device
