"""Read sharding across the GPUs of one node: tasks are independent (chain.c:42-45 touches
only its own a/f/p), so each rank takes a subset, no data-path collective; one all-reduce of three counters at
the end (RCCL over xGMI on GPUs, gloo in CPU tests)."""
import numpy as np
import torch
import torch.distributed as dist


def shard_tasks(task_sizes, world_size, rank):
    """Greedy longest-processing-time partition on anchors per task; deterministic, identical on every rank.
    Returns the sorted task indices of `rank`."""
    sizes = np.asarray(task_sizes, dtype=np.int64)
    order = np.argsort(-sizes, kind="stable")
    load = np.zeros(world_size, dtype=np.int64)
    owner = np.empty(sizes.size, dtype=np.int32)
    if sizes.size and np.all(sizes == sizes[0]):
        owner[:] = np.arange(sizes.size) % world_size     # equal tasks: round robin (same result, O(n))
    else:
        for t in order:
            r = int(np.argmin(load))
            owner[t] = r
            load[r] += sizes[t]
    return np.nonzero(owner == rank)[0]


def sub_batch(offsets, task_ids):
    """CSR offsets of the chosen tasks plus the index ranges to gather their anchors."""
    offsets = np.asarray(offsets, dtype=np.int64)
    n = offsets[1:] - offsets[:-1]
    sel_n = n[task_ids]
    new_off = np.zeros(len(task_ids) + 1, dtype=np.int64)
    new_off[1:] = np.cumsum(sel_n)
    return new_off, offsets[:-1][task_ids], sel_n


def allreduce_counters(anchors, pairs, elapsed_ns, device=None):
    """Sum {anchors chained, pairs evaluated (0 if untracked), elapsed ns} over ranks; max of elapsed too.
    Returns (sum_anchors, sum_pairs, max_elapsed_ns)."""
    if not (dist.is_available() and dist.is_initialized()):
        return int(anchors), int(pairs), int(elapsed_ns)
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    t = torch.tensor([int(anchors), int(pairs)], dtype=torch.int64, device=dev)
    m = torch.tensor([int(elapsed_ns)], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return int(t[0]), int(t[1]), int(m[0])


def gather_elapsed_ns(elapsed_ns, device=None):
    """every rank's elapsed ns, in rank order, on every rank (the bench line names the slowest rank)"""
    if not (dist.is_available() and dist.is_initialized()):
        return [int(elapsed_ns)]
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    mine = torch.tensor([int(elapsed_ns)], dtype=torch.int64, device=dev)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [int(t[0]) for t in out]


def gather_strings(text, device=None, width=160):
    """every rank's short string (device identity: ordinal | PCI bus id | arch), in rank order, on every rank"""
    if not (dist.is_available() and dist.is_initialized()):
        return [str(text)]
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    raw = str(text).encode()[:width]
    mine = torch.zeros(width, dtype=torch.uint8)
    mine[: len(raw)] = torch.tensor(list(raw), dtype=torch.uint8)
    mine = mine.to(dev)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [bytes(t.cpu().tolist()).rstrip(b"\0").decode() for t in out]


def placement_problems(requested_gpus, world_size, identities, shared_device_ok=False):
    """What is wrong with a multi-GPU run's placement, as a list of sentences (empty: nothing): the ranks that came up are not the GPUs that were
    asked for, or two ranks report the same PCI bus id (they share a card: the figure would not be an N-GPU figure).  identities: one
    {"rank", "ordinal", "pci_bus_id", "arch"} per rank."""
    bad = []
    if int(world_size) != int(requested_gpus):
        bad.append(f"--gpus {requested_gpus} but the process group has {world_size} rank(s)")
    if len(identities) != int(world_size):
        bad.append(f"{len(identities)} device identities for {world_size} rank(s)")
    seen = {}
    for d in identities:
        b = d.get("pci_bus_id", "")
        if b in seen and not shared_device_ok:
            bad.append(f"ranks {seen[b]} and {d.get('rank')} both run on the device at PCI bus id {b!r}")
        seen.setdefault(b, d.get("rank"))
    return bad
