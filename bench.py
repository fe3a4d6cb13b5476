#!/usr/bin/env python3
"""bench.py -- anchors/s chained on synthetic ONT anchor streams (BASELINE.json config 2), HBM-resident.

One "step" = one pass of the chaining DP (f[], p[] for every anchor) over one CSR batch of synthetic reads that is
already resident in HBM.  N GPUs = N processes (torch.distributed / RCCL), reads sharded with no data-path
collective (weak scaling: the per-GPU batch is fixed); one all-reduce of counters at the end.  The N ranks come from a launcher
(`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` sets RANK / WORLD_SIZE) or, for the bare
`python bench.py --gpus N`, from this script itself (launch_ranks: N child processes started before any GPU call).

Prints ONE JSON line on rank 0 (see the keys below).  `roofline` prices the dominant kernel against HBM with
24 algorithmic bytes per anchor (16 B mm128_t read + 4 B f + 4 B p written); `cpu_baseline` times the
CPU oracle (a port of chain.c's loop) on the host cores over a bounded sample of the same batch.
"""
import argparse
import json
import os
import sys
import time

# The host-buffer pipelines of the library keep an upload stream and three compute streams busy side by side; the runtime spreads a process's
# streams over GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order, and two streams on one queue run one after the other (measured:
# 23.0 ms per 4.1e7 anchors with the default in this process, 16.8 ms with 16 queues).  Read when the runtime starts, so set before any GPU call;
# mm2c_init does the same for hosts that reach the runtime through the library first.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "minimap2-fpga_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ALGO_BYTES_PER_ANCHOR = 24          # 16 B read + 4 B f + 4 B p
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def host_cores():
    """cores this process may really use: min(affinity mask, cgroup cpu quota)"""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    cores = min(cores, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    cores = min(cores, max(1, q // per))
        except Exception:
            pass
    return cores


def measured_traffic(profile, total_anchors, preset="map-ont"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary of this same command
    (profiles/traffic.json: FETCH_SIZE + 8 B per anchor for the half-counted dwordx4 anchor loads + WRITE_SIZE, separate
    --pmc passes).  PMC counters cannot be read from inside this process, so the value is the profiled one, scaled by
    anchors when the batch size differs; None when no summary exists for the profile."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        import hashlib
        rec = json.load(open(path))[profile if preset == "map-ont" else f"{preset}:{profile}"]   # keyed by stream profile, other presets as preset:profile
        h = hashlib.sha256()
        for fn in ("chain_dp_tile.h", "chain_wave.h", "chain_kernel.hip", "chain_kernel.h"):
            h.update(open(os.path.join(ROOT, "minimap2-fpga_amd", "csrc", fn), "rb").read())
        if rec.get("kernel_source_sha") != h.hexdigest()[:16]:
            return None                                        # the kernel changed since it was profiled: no stale number
        return rec["hbm_bytes_per_launch"] * (total_anchors / rec["anchors_per_launch"])
    except Exception:
        return None


def e2e_map_ont(threads, reads, genome_mb, mini_batch, budget_s=120.0, device_list=None):
    """BASELINE.json's second metric, end-to-end map-ont wall-clock (the reference's pipeline: main.c:406-410 -> mm_map_file, map.c:526-620), on a
    config-3 stand-in (hg38 is not available offline): tools/make_synth_genome.py writes a synthetic genome with planted repeats and simulated ONT
    reads, and three hosts built over the reference's own non-path objects (oracle/ref_host/Makefile -> oracle/_ref/) map them with the same -t / -K:
    mm2_refhost chains on the CPU threads (the stated baseline), mm2_batchhost is INTEGRATION.md path C (matches in, chains out, one library call per
    mini-batch), mm2_gpuhost is path B (one synchronous library call per read through mm_chain_dp).  Wall seconds around each process, the PAF
    of all three must be byte-identical.  Bounded: each run under `timeout`, the whole leg skipped once `budget_s` is used up.  Outside the timed region, on rank 0.
    device_list (N > 1: "0,1,...", the ranks' devices, after the ranks have ended): the two GPU hosts run as ONE process each over all of them (MM2C_DEVICES, the
    in-process form of DESIGN 5: mini-batches split across the devices in path C, every device its own call combiner in path B)."""
    import hashlib
    import re
    import subprocess
    import tempfile
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    exes = {"cpu_chaining": "mm2_refhost", "batched_gpu": "mm2_batchhost", "per_read_gpu": "mm2_gpuhost"}
    if os.path.exists(os.path.join(ref_dir, "mm2_splithost")):
        # the caller's side of path A restated (oracle/ref_host/chain_shim_split.c): chain.c's HW/SW decision with the MI355X constants + the busy protocol, device branch = the product
        exes["split_gpu_cpu"] = "mm2_splithost"
    for e in exes.values():
        if not os.path.exists(os.path.join(ref_dir, e)):
            return {"skipped": f"oracle/_ref/{e} is not there: the hosts are built from the reference's own objects by __graft_entry__.build() where /root/reference exists"}
    t_all = time.perf_counter()
    with tempfile.TemporaryDirectory(prefix="mm2c_e2e_") as w:
        pre = os.path.join(w, "syn")
        t0 = time.perf_counter()
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_synth_genome.py"), pre, "--genome-mb", str(genome_mb), "--reads", str(reads)],
                              stdout=subprocess.DEVNULL)
        out = {"workload": f"map-ont, synthetic {genome_mb} Mb genome with planted repeats (4 sequences), {reads} simulated ONT reads (10 kb, 10 % error), "
                           f"-t {threads}, mini-batches of {mini_batch} bases (-K)", "stand_in_for": "BASELINE config 3 (hg38 + 100k ONT reads: not available offline)",
               "threads": threads, "reads": reads, "genome_mb": genome_mb, "generate_s": round(time.perf_counter() - t0, 2),
               "devices_of_the_gpu_hosts": device_list or "0", "hosts": {}}
        md5s = {}
        for name, exe in exes.items():
            if time.perf_counter() - t_all > budget_s:
                out["hosts"][name] = {"skipped": f"the leg's budget of {budget_s:.0f} s was used up"}
                continue
            env = dict(os.environ, MM2_MINI_BATCH=str(mini_batch), MM2C_QUIET="1")
            if device_list and name != "cpu_chaining":
                env["MM2C_DEVICES"] = device_list
            paf = os.path.join(w, name + ".paf")
            t0 = time.perf_counter()
            with open(paf, "wb") as fo:
                r = subprocess.run(["timeout", "-k", "10", "120", os.path.join(ref_dir, exe), "-t", str(threads), pre + ".ref.fa", pre + ".reads.fa"],
                                   stdout=fo, stderr=subprocess.PIPE, env=env)
            wall = time.perf_counter() - t0
            err = r.stderr.decode(errors="replace")
            h = hashlib.md5(open(paf, "rb").read()).hexdigest()
            rec = {"exe": "oracle/_ref/" + exe, "wall_s": round(wall, 3), "rc": r.returncode, "paf_md5": h, "paf_lines": sum(1 for _ in open(paf, "rb"))}
            m = re.search(r"stages \(summed over mini-batches, they overlap\): (.*)", err)
            if m:
                rec["host_stage_sums"] = m.group(1).strip()
            m = re.search(r"inside the library \(mm2c_get_stage_stats\): (.*)", err)
            if m:
                rec["library_stage_stats"] = m.group(1).strip()
            m = re.search(r"(\d+) reads, (\d+) anchors; ([0-9.]+) s in the batched GPU calls", err)
            if m:
                rec["anchors"] = int(m.group(2)); rec["gpu_calls_s"] = float(m.group(3))
            m = re.search(r"GPU chaining: (.*)", err)
            if m:
                rec["per_read_calls"] = m.group(1).strip()
            m = re.search(r"\[mm2_splithost\] (split model .*)", err)
            if m:
                rec["split"] = m.group(1).strip()
            m = re.search(r"per device slot: (.*)", err)
            if m:
                rec["per_device_slot"] = m.group(1).strip()
            if r.returncode != 0:
                rec["stderr_tail"] = err[-400:]
            out["hosts"][name] = rec
            md5s[name] = h if r.returncode == 0 else None
        done = [v for v in md5s.values() if v]
        out["paf_identical"] = bool(len(done) == len(exes) and len(set(done)) == 1)
        cpu = out["hosts"].get("cpu_chaining", {}).get("wall_s")
        for k in ("batched_gpu", "per_read_gpu", "split_gpu_cpu"):
            wk = out["hosts"].get(k, {}).get("wall_s")
            if cpu and wk and out["hosts"][k].get("rc") == 0:
                out[k + "_vs_cpu_chaining"] = round(wk / cpu, 3)          # < 1: faster than the CPU-chaining host
        out["what"] = ("wall seconds around each process (index + FASTA reading + seeding + chaining + alignment-free post-processing + PAF output; process and HIP start-up included), "
                       "same FASTA files, same -t and -K")
    return out


def plan_predict_totals(mm2chain, P, off1, a1):
    """chain.c:53-78 on the GPU for the distinct reads: (sum num_subparts, sum total_subparts, sum total_trip_count)"""
    pl = mm2chain.ChainPlan(P, off1.numpy())
    ns, ts, tt = pl.predict(a1)
    torch.cuda.synchronize()
    r = (int(ns.sum()), int(ts.sum()), int(tt.sum()))
    pl.close()
    return r


KFD_NODES = os.environ.get("MM2C_BENCH_KFD_NODES", "/sys/class/kfd/kfd/topology/nodes")      # (the variable: tests point it at a made-up tree)


def count_gpu_nodes(base=None):
    """GPUs of this node WITHOUT starting a GPU runtime in this process: the KFD topology nodes that have SIMDs (a CPU node has simd_count 0).
    None when there is no topology to read (no amdgpu driver, or sysfs not visible).  torch.cuda.device_count() is not used for this: without
    amdsmi it falls back to hipGetDeviceCount, and the launcher parent would then sit on a runtime context on every GPU while the ranks run."""
    base = base or KFD_NODES
    try:
        nodes = sorted(os.listdir(base))
    except OSError:
        return None
    n, readable = 0, 0
    for d in nodes:
        try:
            with open(os.path.join(base, d, "properties")) as fh:
                readable += 1
                for line in fh:
                    k, _, v = line.strip().partition(" ")
                    if k == "simd_count" and int(v) > 0:
                        n += 1
        except (OSError, ValueError):
            continue
    return n if readable else None


def launch_ranks(n):
    """What `python -m torch.distributed.run --nnodes=1 --nproc-per-node n bench.py ...` would do, for the bare `python bench.py --gpus n`:
    n child processes of this script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1.  Rank 0's JSON line goes to
    this process's stdout (inherited).  Returns the exit code: 0 only when every rank ended with 0; when one fails the others are ended."""
    import socket
    import subprocess
    if os.environ.get("MM2C_BENCH_ONE_DEVICE") != "1" and os.environ.get("MM2C_BENCH_REHEARSE_NO_GPU") != "1":
        have = count_gpu_nodes()                                # from the KFD topology in sysfs: this process never starts a GPU runtime
        if have is not None and have < n:
            print(f"bench.py: --gpus {n} but this node shows {have} GPU(s) ({KFD_NODES})", file=sys.stderr)
            return 2
        # (no topology to read: no pre-check -- a rank that cannot select its device ends with an error and takes the others with it)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc, live = 0, set(range(n))
    while live:
        for r in list(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f"bench.py: rank {r} ended with code {code}; ending the other ranks", file=sys.stderr)
                for o in live:
                    procs[o].terminate()                         # exactly the processes started above
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--profile", default="mixed", choices=["sparse", "mixed", "dense", "colinear"])
    ap.add_argument("--reads", type=int, default=65536, help="reads (tasks) per GPU per step")
    ap.add_argument("--distinct", type=int, default=8192, help="distinct synthetic reads generated; tiled up to --reads")
    ap.add_argument("--anchors-per-read", type=int, default=5000)
    ap.add_argument("--ragged", action="store_true", help="anchors per read ~ U[0.2, 1.8] x --anchors-per-read (SURVEY 8d ragged variant)")
    ap.add_argument("--seed", type=int, default=20240)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the CPU baseline leg (0 = skip)")
    ap.add_argument("--preset", default="map-ont", choices=["map-ont", "asm20", "ava-ont"],
                    help="chaining scalars + stream shape: map-ont (BASELINE config 2, default), asm20 (config 4 stand-in: span 19, "
                         "7500 anchors/read), ava-ont (config 5 stand-in: bw 2000, max_gap 10000, 20000 anchors/read)")
    ap.add_argument("--ring-class", type=int, default=None)
    ap.add_argument("--general", action="store_true", help="run the general kernel variant (segment ids / cDNA branches, chain.c:206,211-217) on the same stream")
    ap.add_argument("--gap-scale", type=float, default=None, help="chain_gap_scale other than 1 (the f64 path of chain.c:219)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: ONE fixed batch of --reads reads (the same on every rank), its tasks dealt to the ranks by "
                         "sharding.shard_tasks (longest first, on anchors per task) and gathered into a rank-local CSR batch; "
                         "value = anchors of the whole batch / max-over-ranks time")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary figures (extra launches); used when profiling")
    ap.add_argument("--no-long", action="store_true", help="skip the long-read leg (BASELINE config 5's regime: 1e5 .. 1e6 anchors per read, about half a minute)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end map-ont leg (three host processes over a synthetic genome, about half a minute)")
    ap.add_argument("--e2e-reads", type=int, default=120000)
    ap.add_argument("--e2e-genome-mb", type=float, default=50.0)
    ap.add_argument("--e2e-mini-batch", type=int, default=100_000_000, help="-K of the three hosts (bases per mini-batch)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks here, one process per GPU, BEFORE this process makes any GPU call
        # (fresh children, never an exec of a process that has touched the GPU); this process only waits for them
        sys.exit(launch_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knobs (never set by the driver): MM2C_BENCH_BACKEND=gloo and MM2C_BENCH_ONE_DEVICE=1 let several ranks share
    # one GPU on a 1-GPU box to exercise the multi-process flow; the real runs use nccl (= RCCL) with one GPU per rank
    backend = os.environ.get("MM2C_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("MM2C_BENCH_ONE_DEVICE") == "1" else local_rank
    if os.environ.get("MM2C_BENCH_REHEARSE_NO_GPU") == "1":
        # launcher rehearsal on a box without a GPU (tests/test_cpu_sharding.py): rendezvous, the counter all-reduce and the per-rank gather
        # with NO chaining work at all -- it exercises launch_ranks() and the rank plumbing, and its line says so (value null)
        from mm2chain import sharding
        if world > 1:
            dist.init_process_group("gloo")
        tot, _, mx = sharding.allreduce_counters(1000 * (rank + 1), 0, 1000 + rank)
        per = sharding.gather_elapsed_ns(1000 + rank)
        fake = os.environ.get("MM2C_BENCH_FAKE_BUS_IDS", "").split(",")          # tests: what each rank "sits on"
        me = {"rank": rank, "ordinal": dev_index, "pci_bus_id": fake[rank] if rank < len(fake) and fake[rank] else f"rehearsal:{rank}", "arch": "none"}
        devices = [json.loads(t) for t in sharding.gather_strings(json.dumps(me))]
        bad = sharding.placement_problems(args.gpus, dist.get_world_size() if world > 1 else 1, devices, shared_device_ok=os.environ.get("MM2C_BENCH_ONE_DEVICE") == "1")
        if rank == 0 and not bad:
            print(json.dumps({"metric": "anchors/sec chained", "value": None, "unit": "anchors/s", "n_gpus": world, "world_size_seen": world,
                              "rehearsal": "launcher and rank plumbing only: no GPU, no chaining work", "counters_sum": tot, "max_ns": mx,
                              "per_rank_ns": per, "requested_gpus": args.gpus, "devices": devices}))
        if world > 1:
            dist.destroy_process_group()
        if bad:
            if rank == 0:
                print("bench.py: " + "; ".join(bad), file=sys.stderr)
            sys.exit(3)
        return
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"

    import mm2chain
    from mm2chain import synth, params, sharding
    mm2chain.init(torch.cuda.current_device())
    if args.ring_class is not None:
        mm2chain.tune("ring_class", args.ring_class)
    # ---- who runs where: every rank's device (HIP ordinal, PCI bus id, architecture) on every rank; a run whose ranks are not the GPUs that were
    # asked for, or whose ranks share a card, prints no figure and ends with code 3 on every rank (the rehearsal knob for one shared card aside)
    ident = dict(mm2chain.device_identity(), rank=rank)
    devices = [json.loads(t) for t in sharding.gather_strings(json.dumps(ident))]
    bad = sharding.placement_problems(args.gpus, dist.get_world_size() if world > 1 else 1, devices, shared_device_ok=os.environ.get("MM2C_BENCH_ONE_DEVICE") == "1")
    if bad:
        if rank == 0:
            print("bench.py: " + "; ".join(bad), file=sys.stderr)
        if world > 1:
            dist.destroy_process_group()
        sys.exit(3)
    P = params.map_ont()                                       # max_iter = 5000, max_skip = 25 (options.c:29-30)
    q_span, locus = 15, None
    if args.preset == "asm20":                                 # options.c:113-122: k = 19; cleaner, longer chains
        P, q_span = params.asm20(), 19
        if args.anchors_per_read == 5000:
            args.anchors_per_read = 7500
    elif args.preset == "ava-ont":                             # options.c:83-86
        P = params.ava_ont()
        if args.anchors_per_read == 5000:
            args.anchors_per_read = 20000
        locus = 400000
        if args.reads == 65536:
            args.reads, args.distinct = 16384, min(args.distinct, 2048)

    if args.general:
        P.flags |= mm2chain.MM2C_F_FORCE_GENERAL
    if args.gap_scale is not None:
        P.gap_scale = args.gap_scale
    # ---- synthetic batch of this rank, generated on the device (deterministic: splitmix64 of seed + rank)
    distinct = min(args.distinct, args.reads)
    times = max(1, args.reads // distinct)
    n_per = (int(0.2 * args.anchors_per_read), int(1.8 * args.anchors_per_read)) if args.ragged else args.anchors_per_read
    off1, a1 = synth.make_stream(args.profile, distinct, n_per, seed=args.seed + (0 if args.strong else rank), q_span=q_span, device="cuda", locus=locus)
    distinct_chk = distinct
    if not args.strong:
        off, anchors = synth.replicate(off1, a1, times)
        global_total = int(off[-1])
    else:
        # ONE batch for all ranks: the `distinct` synthetic reads tiled `times` times (task t = distinct read t mod distinct, as
        # synth.replicate lays them out); this rank keeps the tasks shard_tasks deals it and builds its own CSR batch from them.
        sizes1 = (off1[1:] - off1[:-1]).numpy()
        sizes = np.tile(sizes1, times)
        global_total = int(sizes.sum())
        ids = sharding.shard_tasks(sizes, world, rank)
        d_ids = ids % distinct
        lens = sizes1[d_ids]
        new_off = np.zeros(ids.size + 1, dtype=np.int64); new_off[1:] = np.cumsum(lens)
        starts = off1.numpy()[:-1][d_ids]
        n_loc_rows = int(new_off[-1])
        anchors = torch.empty((n_loc_rows, 2), dtype=torch.int64, device="cuda")
        rel = torch.from_numpy(starts - new_off[:-1]).cuda()              # source row = destination row + rel[task]
        lens_t = torch.from_numpy(lens).cuda()
        CH = 1 << 24                                                      # rows per gather: torch's indexing kernels go wrong beyond 2^31 bytes
        t_lo = 0
        while t_lo < ids.size:                                            # whole tasks per chunk
            t_hi = int(np.searchsorted(new_off, new_off[t_lo] + CH, side="right")) - 1
            t_hi = max(t_hi, t_lo + 1)
            r0, r1 = int(new_off[t_lo]), int(new_off[t_hi])
            src = torch.repeat_interleave(rel[t_lo:t_hi], lens_t[t_lo:t_hi]) + torch.arange(r0, r1, device="cuda")
            anchors[r0:r1] = a1.index_select(0, src)
            t_lo = t_hi
        for t in (0, ids.size - 1):                                       # the gather is part of the harness: check it
            assert torch.equal(anchors[int(new_off[t]):int(new_off[t + 1])], a1[int(starts[t]):int(starts[t]) + int(lens[t])])
        off = torch.from_numpy(new_off)
        # the reads the oracle check below looks at: the first ones of THIS rank's batch
        n_loc = min(64, ids.size)
        off1 = off[: n_loc + 1].clone(); a1 = anchors[: int(off1[-1])].clone()
        distinct_chk = n_loc
        del src, lens_t, rel
    n_tasks = off.numel() - 1
    total = int(off[-1])
    d_f = torch.empty(total, dtype=torch.int32, device="cuda")
    d_p = torch.empty_like(d_f)
    plan = mm2chain.ChainPlan(P, off.numpy())
    torch.cuda.synchronize()

    def step():
        plan.run(anchors, d_f, d_p)                             # on torch's current stream

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    kernel_ms, prepass_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kernel_ms.append(plan.last_kernel_ms())                 # HIP events on the launch stream, recorded by the library
        prepass_ms.append(plan.last_prepass_ms())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    tot_anchors, _, max_ns = sharding.allreduce_counters(total * args.steps, 0, int(elapsed * 1e9))
    per_rank_ns = sharding.gather_elapsed_ns(int(elapsed * 1e9))
    wall = max_ns / 1e9

    # ---- correctness of what was just timed (outside the timed region): f / p of EVERY distinct read of this rank's batch against the oracle -- the replicas behind
    # them are copies of the same reads, compared with the first replica on the device.  The oracle pass doubles as the first pass of the cpu_baseline leg.
    import oracle_binding as ob
    n_check = distinct_chk
    end = int(off1[n_check])
    chk_threads = max(1, host_cores() // max(world, 1))
    a1_np = a1[:end].cpu().numpy().view(np.uint64)
    f_ref, p_ref, chk_s = ob.chain_batch(P, off1[: n_check + 1].numpy(), a1_np, chk_threads)
    verified = bool(np.array_equal(d_f[:end].cpu().numpy(), f_ref) and np.array_equal(d_p[:end].cpu().numpy(), p_ref))
    del f_ref, p_ref
    if not args.strong:
        last = total - int(off1[-1])                            # the last replica must equal the first one
        verified = verified and bool(torch.equal(d_f[last:], d_f[: int(off1[-1])])) and bool(torch.equal(d_p[last:], d_p[: int(off1[-1])]))
        if times > 2:                                            # ... and one in the middle
            mid = (times // 2) * int(off1[-1])
            verified = verified and bool(torch.equal(d_f[mid: mid + int(off1[-1])], d_f[: int(off1[-1])])) and bool(torch.equal(d_p[mid: mid + int(off1[-1])], d_p[: int(off1[-1])]))
    if world > 1:                                               # every rank's check counts
        vt = torch.tensor([1 if verified else 0], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(vt, op=dist.ReduceOp.MIN)
        verified = bool(int(vt[0]))

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    k_avg_ms = float(np.mean(kernel_ms))
    achieved = total * ALGO_BYTES_PER_ANCHOR / (k_avg_ms * 1e-3) / 1e9
    out = {
        "metric": "anchors/sec chained", "value": tot_anchors / wall, "unit": "anchors/s",
        "n_gpus": world, "world_size_seen": (dist.get_world_size() if world > 1 else 1), "backend": (backend if world > 1 else None), "devices": devices,
        "per_rank_ms_per_step": [ns / 1e6 / args.steps for ns in per_rank_ns], "slowest_rank": int(np.argmax(per_rank_ns)), "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall * 1e3 / args.steps,
        "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": f"synthetic ONT anchor stream ({args.profile}), {args.anchors_per_read} anchors/read, "
                               f"{args.preset} chaining params (max_dist={P.max_dist_x}, bw={P.bw}, max_iter={P.max_iter}, max_skip={P.max_skip}), "
                               f"HBM-resident",
                   "reads_per_gpu_per_step": n_tasks, "reads_whole_batch": (global_total and (args.reads if args.strong else n_tasks * world)), "distinct_reads": distinct, "anchors_per_read": args.anchors_per_read,
                   "profile": args.profile, "preset": args.preset, "ragged": bool(args.ragged),
                   "parallelism": f"read-sharded x{world}" + (" (one batch, tasks dealt longest-first)" if args.strong else "")},
        "verified_vs_oracle": verified, "verified_reads": n_check,
        "verified_what": "f[] and p[] of the step just timed, every distinct read of the batch element-wise against the CPU oracle; the first, middle and last replica equal on the device",
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": measured_traffic(args.profile, total, args.preset), "kernel": "chain_dp_tile", "kernel_ms_avg": k_avg_ms,
                     "prepass_kernel_ms_avg": float(np.mean(prepass_ms)),
                     "algorithmic_bytes_per_launch": total * ALGO_BYTES_PER_ANCHOR},
    }

    # ---- secondary figures (outside the timed region, rank 0)
    try:
        if args.no_secondary or world > 1 or args.strong:                                # N = 1 only: keep multi-GPU runs lean
            raise StopIteration
        _, _, tt = plan_predict_totals(mm2chain, P, off1, a1)
        out["secondary"] = {"nominal_cells_per_s": float(tt) * times / (k_avg_ms * 1e-3),
                            "nominal_cells_definition": "sum over anchors of min(i - st, 1024), the reference's total_trip_count (chain.c:69)",
                            "nominal_cells_per_anchor": float(tt) / int(off1[-1])}
        # whole mm_chain_dp: DP + the epilogue of chain.c:348-422 on the GPU, chains left in HBM
        min_cnt, min_sc = (3, 100) if args.preset == "ava-ont" else (3, 40)     # options.c:24-25,85
        u_off, u, b_off, b = plan.chains(anchors, d_f, d_p, min_cnt, min_sc)   # warm-up (allocates the scratch)
        torch.cuda.synchronize()
        del u_off, u, b_off, b                                                  # so that the timed call reuses these blocks instead of allocating
        tw = time.perf_counter()
        plan.run(anchors, d_f, d_p)
        u_off, u, b_off, b = plan.chains(anchors, d_f, d_p, min_cnt, min_sc)
        torch.cuda.synchronize()
        tw = time.perf_counter() - tw
        n_chk = min(16, distinct)
        uo, bo = u_off[: n_chk + 1].cpu().numpy(), b_off[: n_chk + 1].cpu().numpy()
        u_h, b_h = u[: int(uo[-1])].cpu().numpy().view(np.uint64), b[: int(bo[-1])].cpu().numpy().view(np.uint64)
        a_chk = a1[: int(off1[n_chk])].cpu().numpy().view(np.uint64)
        ok = True
        for k in range(n_chk):
            u_ref, b_ref = ob.mm_chain_dp(P, min_cnt, min_sc, a_chk[int(off1[k]):int(off1[k + 1])])
            ok = ok and np.array_equal(u_h[uo[k]:uo[k + 1]], u_ref) and np.array_equal(b_h[bo[k]:bo[k + 1]], b_ref)
        out["whole_mm_chain_dp"] = {"value": total / tw, "unit": "anchors/s", "epilogue_ms": plan.last_epilogue_ms(),
                                    "chains": int(u_off[-1]), "chained_anchors": int(b_off[-1]), "min_cnt": min_cnt, "min_sc": min_sc,
                                    "what": "prepass + DP + device epilogue (v[], chain ends, backtrack, filter, chain order), HBM-resident, 1 step",
                                    "verified_vs_oracle": bool(ok)}
        del u, b
        # seed hits -> sorted anchors on the GPU: matches derived from the first reads of this batch (their anchors,
        # regrouped by query position), tiled; the anchors that come out are checked against the oracle's collect_seed_hits
        n_s = min(distinct, 256)
        qlen_s = 1 << 20
        ms_, hs_, mo_, ao_ = [], [], [0], [0]
        a_s = a1[: int(off1[n_s])].cpu().numpy().view(np.uint64)
        for k in range(n_s):
            m_k, h_k = synth.matches_from_anchors(a_s[int(off1[k]):int(off1[k + 1])], qlen_s)
            m_k["cr_off"] += ao_[-1]
            ms_.append(m_k); hs_.append(h_k); mo_.append(mo_[-1] + m_k.size); ao_.append(ao_[-1] + h_k.size)
        rep = 32
        m1_, h1_ = np.concatenate(ms_), np.concatenate(hs_)
        mt_ = np.tile(m1_, rep); mt_["cr_off"] += np.repeat(np.arange(rep, dtype=np.int64) * h1_.size, m1_.size)
        mo_t = np.concatenate([[0], np.tile(np.diff(mo_), rep).cumsum()]).astype(np.int64)
        ao_t = np.concatenate([[0], np.tile(np.diff(ao_), rep).cumsum()]).astype(np.int64)
        sp = mm2chain.SeedPlan(mo_t, ao_t)
        d_m = torch.from_numpy(mt_.view(np.uint8)).cuda(); d_h = torch.from_numpy(np.tile(h1_, rep).view(np.int64)).cuda()
        d_q = torch.full((n_s * rep,), qlen_s, dtype=torch.int32, device="cuda")
        d_as = sp.run(d_m, d_h, d_q)
        d_as = sp.run(d_m, d_h, d_q, d_as)
        n_tie_reads = sp.check()
        got = d_as[: int(ao_[4])].cpu().numpy().view(np.uint64)
        ok_s = True
        for k in range(4):
            mk = ms_[k].copy(); mk["cr_off"] -= ao_[k]
            ok_s = ok_s and np.array_equal(got[ao_[k]:ao_[k + 1]], ob.collect_seed_hits(mk, hs_[k], qlen_s))
        out["seed_hits"] = {"value": int(ao_t[-1]) / (sp.last_ms() * 1e-3), "unit": "anchors/s", "ms": sp.last_ms(),
                            "reads": n_s * rep, "anchors": int(ao_t[-1]), "matches": int(mt_.size), "reads_with_equal_x": int(n_tie_reads),
                            "what": "matches -> anchors as collect_seed_hits leaves them (map.c:215-247, incl. radix_sort_128x's order among equal x), HBM-resident",
                            "verified_vs_oracle": bool(ok_s)}
        sp.close(); del d_m, d_h, d_as
        # one synchronous call for ONE read, the reference's call pattern (chain.c:103 -> run_chaining_on_hw, chain_hardware.cpp:27-197): the dispatch symbol itself (V2:
        # look-back 1024, no max-skip) and the extended entry (V1 = stock mm_chain_dp), upload / launches / download / synchronisation included, beside one CPU thread
        t_one = a1[: int(off1[1])].cpu().numpy().view(np.uint64)
        avg_one = ob.avg_qspan(t_one)
        P2 = params.make_params(P.max_dist_x, P.max_dist_y, P.bw, max_skip=2**31 - 1, max_iter=1024, q_span_override=q_span, flags=mm2chain.MM2C_F_IGNORE_SEG)
        f1_ref, p1_ref, _ = ob.chain_fpv(P, t_one, avg_one)
        f2_ref, p2_ref, _ = ob.chain_fpv(P2, t_one, avg_one)
        lone = {"task": f"the first read of the batch ({t_one.shape[0]} anchors)", "what": "best and median of 40 synchronous calls from one host thread after 200 such calls, idle GPU otherwise"}
        for key, call, ref in (("run_chaining_on_hw_ms", lambda: mm2chain.run_chaining_on_hw(t_one.shape[0], P.max_dist_x, P.max_dist_y, P.bw, q_span, avg_one, t_one)[1:], (f2_ref, p2_ref)),
                               ("mm2c_chain_task_host_ms", lambda: mm2chain.chain_task(P, t_one, avg_one), (f1_ref, p1_ref))):
            ts, same = [], True
            for k in range(240):                                             # 200 calls first: the leg follows host-side checks, the GPU clocks are down when it starts
                tl = time.perf_counter(); fo, po = call(); ts.append(time.perf_counter() - tl)
                same = same and (k > 0 or (np.array_equal(fo, ref[0]) and np.array_equal(po, ref[1])))
            ts = np.array(ts[200:]) * 1e3
            lone[key] = {"best": round(float(ts.min()), 4), "median": round(float(np.median(ts)), 4), "identical_to_oracle": bool(same)}
        lone["kernel"] = mm2chain.last_host_variant()
        tl = time.perf_counter()
        for _ in range(5):
            ob.chain_fpv(P, t_one, avg_one)
        lone["cpu_thread_ms"] = round((time.perf_counter() - tl) / 5 * 1e3, 4)
        out["lone_call"] = lone
        n_h = min(distinct, 8192)                                            # the distinct reads of the batch: 4.1e7 anchors at the default sizes
        a_host = a1[: int(off1[n_h])].cpu().numpy().view(np.uint64)
        off_host = off1[: n_h + 1].numpy()
        mm2chain.chain_batch_host(P, off_host, a_host)                       # warm up staging buffers
        th = time.perf_counter()
        fh, ph = mm2chain.chain_batch_host(P, off_host, a_host)
        th = time.perf_counter() - th
        out["host_streamed"] = {"value": int(off_host[-1]) / th, "unit": "anchors/s",
                                "sample": f"{n_h} reads ({int(off_host[-1])} anchors) from pageable host memory: H2D + prepass + DP + D2H + sync, 1 call",
                                "matches_resident": bool(np.array_equal(fh, d_f[: int(off_host[-1])].cpu().numpy()))}
        pa = mm2chain.PinnedArray(a_host.shape, np.uint64); pf = mm2chain.PinnedArray(fh.shape, np.int32); pp = mm2chain.PinnedArray(ph.shape, np.int32)
        pa.array[:] = a_host
        mm2chain.chain_batch_host_into(P, off_host, pa.array, pf.array, pp.array)
        ths = []
        for _ in range(10):                                                  # short bursts after host-side work: the first calls run before the GPU clocks are up
            th = time.perf_counter()
            mm2chain.chain_batch_host_into(P, off_host, pa.array, pf.array, pp.array)
            ths.append(time.perf_counter() - th)
        th = min(ths)
        out["host_streamed_pinned"] = {"value": int(off_host[-1]) / th, "unit": "anchors/s", "ms_of_10_calls": [round(t * 1e3, 2) for t in ths],
                                       "sample": "same call with anchors and outputs in page-locked host memory (mm2c_pinned_alloc): chunks uploaded back to back on one stream, "
                                                 "their kernels and downloads on three compute streams in turn",
                                       "matches_resident": bool(np.array_equal(pf.array, fh) and np.array_equal(pp.array, ph))}
        # ---- long reads (round 6; BASELINE config 5's regime, SURVEY 8 a1: 1e5 .. 1e6 anchors per task): batches with fewer tasks than the GPU has wave slots, at the
        # anchor density of the ava-ont stream, -x ava-ont scalars.  DP through a plan on the library's default route (which kernel took the pieces is reported) and the
        # seed-hit path; f / p and the sorted anchors of the first reads against the oracle.  tools/long_reads.py is the same measurement with every route.
        if not args.no_long and args.preset == "map-ont":
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import long_reads
            lr = {"what": "DP (prepass + kernel, HIP events) and seed hits -> sorted anchors on synthetic long reads: mixed profile at 50 anchors per kb (locus = 20 x anchors per read), "
                          "-x ava-ont scalars (max_dist 10000, bw 2000), HBM-resident, best of 3; a few distinct reads replicated into separate memory", "sizes": []}
            for reads_l, per_l in ((2048, 100000), (1024, 300000), (256, 1000000)):
                r_l = long_reads.measure(reads_l, per_l, routes=("auto",), reps=3, check=1)
                lr["sizes"].append({"reads": r_l["reads"], "anchors_per_read": per_l, "dp": r_l["dp"]["auto"], "seed_hits": r_l["seed_hits"]})
            lr["dp_min_value"] = min(s_["dp"]["value"] for s_ in lr["sizes"]); lr["seed_hits_min_value"] = min(s_["seed_hits"]["value"] for s_ in lr["sizes"])
            lr["verified_vs_oracle"] = bool(all(s_["dp"]["identical_to_oracle"] and s_["seed_hits"]["identical_to_oracle"] for s_ in lr["sizes"]))
            out["long_reads"] = lr
    except StopIteration:
        pass
    except Exception as e:                                                   # never let a secondary figure break the line
        out["secondary_error"] = repr(e)

    # ---- CPU baseline: the oracle (port of chain.c:184-238) on the host cores, bounded sample of the same batch
    if args.cpu_seconds > 0 and world == 1:                                  # rank 0 at N = 1 only
        cores = host_cores()
        off_np = off1.numpy()
        a_np = a1.cpu().numpy().view(np.uint64)
        cpu_model = "unknown"
        try:
            for line in open("/proc/cpuinfo"):
                if line.lower().startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
        except OSError:
            pass
        probe = min(distinct, 4 * cores)
        _, _, s = ob.chain_batch(P, off_np[: probe + 1], a_np[: int(off_np[probe])], cores)
        rate = int(off_np[probe]) / max(s, 1e-6)
        want = rate * args.cpu_seconds / args.anchors_per_read            # reads worth ~cpu_seconds of all-core work
        n_s = int(max(cores, min(distinct, want)))
        reps = int(max(1, min(64, round(want / n_s))))
        s_all = 0.0
        for _ in range(reps):
            s_all += ob.chain_batch(P, off_np[: n_s + 1], a_np[: int(off_np[n_s])], cores)[2]
        n1 = int(max(1, min(n_s, 2_000_000 // args.anchors_per_read)))
        _, _, s_one = ob.chain_batch(P, off_np[: n1 + 1], a_np[: int(off_np[n1])], 1)
        out["cpu_baseline"] = {"value": reps * int(off_np[n_s]) / s_all, "unit": "anchors/s", "cores": cores, "cpu_model": cpu_model, "kind": "port",
                               "verification_pass": {"reads": n_check, "anchors": end, "threads": chk_threads, "seconds": round(chk_s, 3)},
                               "sample": f"first {n_s} reads of the same batch x {reps} passes ({reps * int(off_np[n_s])} anchors), "
                                         f"{cores} threads, tasks round-robin, {s_all:.1f} s wall",
                               "value_1thread": int(off_np[n1]) / s_one}
    # ---- end to end (BASELINE.json's second metric): after everything else, the GPU memory of this process given back first
    # N > 1: the other ranks have ended by now (they return right after the verification all-reduce) and this rank leaves the process group first; the two
    # GPU hosts then run as one process each over the ranks' devices (MM2C_DEVICES)
    if world > 1:
        dist.destroy_process_group()
    if not args.strong and not args.no_e2e and (world > 1 or not args.no_secondary) and args.preset == "map-ont":
        try:
            del d_f, d_p, anchors
            plan.close(); plan = None
            mm2chain.tune("trim", 0)
            torch.cuda.empty_cache()
            dev_list = None
            if world > 1:
                mm2chain.shutdown()
                dev_list = ",".join("0" if os.environ.get("MM2C_BENCH_ONE_DEVICE") == "1" else str(d["ordinal"]) for d in devices)
                time.sleep(1.0)                                     # the other ranks' processes are on their way out
            out["e2e_map_ont"] = e2e_map_ont(host_cores(), args.e2e_reads, args.e2e_genome_mb, args.e2e_mini_batch, device_list=dev_list)
        except Exception as e:
            out["e2e_map_ont"] = {"error": repr(e)}
    print(json.dumps(out))
    if plan is not None:
        plan.close()


if __name__ == "__main__":
    main()
