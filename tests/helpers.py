"""shared test helpers: task construction and GPU-vs-oracle comparison"""
import numpy as np
import torch

import oracle_binding as ob


def mk_anchor(strand, rid, rpos, qpos, span=15, seg=0, flags=0):
    x = (strand << 63) | (rid << 32) | rpos
    y = flags | (seg << 48) | (span << 32) | (qpos & 0xFFFFFFFF)
    return x, y


def pack(rows):
    """rows: list of (x, y) python ints -> uint64 [n,2], sorted by x (stable)"""
    a = np.array(rows, dtype=np.uint64).reshape(-1, 2)
    o = np.argsort(a[:, 0], kind="stable")
    return np.ascontiguousarray(a[o])


def oracle_batch(par, offsets, anchors):
    """per-task oracle f, p (avg computed per task as chain.c:48-49)"""
    f, p, _ = ob.chain_batch(par, offsets, anchors, n_threads=4)
    return f, p


# Which DP kernel a plan runs is decided per run since round 6 ("coop_plans" 2: few long pieces -> sixteen waves per piece, chain_dp_coop; anything else -> one wave per
# piece, chain_dp_tile / chain_dp_wave).  The parity inputs were written against the instantiations of the one-wave kernels, and many tests assert which of them ran --
# and nearly all of them are small batches, which the default now sends to the cooperative kernel.  So gpu_batch runs every input TWICE unless a test has pinned the
# route itself (the `knobs` fixture notes it here): with one wave per piece (what is returned, and what `variant` names) and with the library's default route, and
# the two results must be equal element for element.  ROUTE_LOG collects what the default chose (pieces, one-wave pieces, cooperative pieces).
PINNED_ROUTE = None
ROUTE_LOG = []


def _run_plan(par, offsets, d_a, d_avg, variant):
    import mm2chain
    total = d_a.shape[0]
    d_f = torch.full((total,), -77, dtype=torch.int32, device="cuda")
    d_p = torch.full((total,), -77, dtype=torch.int32, device="cuda")
    plan = mm2chain.ChainPlan(par, offsets)
    plan.run(d_a, d_f, d_p, d_avg)
    torch.cuda.synchronize()
    if variant is not None:
        variant.append(plan.last_variant())
    route = plan.last_route()
    plan.close()
    return d_f.cpu().numpy(), d_p.cpu().numpy(), route


def gpu_batch(par, offsets, anchors, avg=None, variant=None):
    """run the HIP path through the C ABI (plan, device-resident) and return f, p as numpy; `variant`: a list that receives the text of
    mm2c_plan_last_variant (which kernel instantiation ran)"""
    import mm2chain
    a_np = np.ascontiguousarray(anchors).view(np.int64).reshape(-1, 2)
    d_a = torch.from_numpy(a_np).cuda()
    d_avg = torch.from_numpy(np.asarray(avg, dtype=np.float32)).cuda() if avg is not None else None
    if PINNED_ROUTE is not None:
        f, p, _ = _run_plan(par, offsets, d_a, d_avg, variant)
        return f, p
    try:
        mm2chain.tune("coop_plans", 0)
        f, p, _ = _run_plan(par, offsets, d_a, d_avg, variant)
    finally:
        mm2chain.tune("coop_plans", 2)
    f2, p2, route = _run_plan(par, offsets, d_a, d_avg, None)
    ROUTE_LOG.append(route)
    bad = np.nonzero((f != f2) | (p != p2))[0]
    if bad.size:
        i = int(bad[0])
        raise AssertionError(f"the library's default route (pieces {route[0]}: {route[1]} with one wave, {route[2]} with sixteen) differs from one wave per piece at "
                             f"{bad.size} of {f.size} anchors; first at {i}: f {f2[i]} vs {f[i]}, p {p2[i]} vs {p[i]}")
    return f, p


def assert_same(f, p, f_ref, p_ref, offsets=None, what=""):
    bad = np.nonzero((f != f_ref) | (p != p_ref))[0]
    if bad.size:
        i = int(bad[0])
        task = int(np.searchsorted(np.asarray(offsets), i, side="right") - 1) if offsets is not None else -1
        raise AssertionError(f"{what}: {bad.size} of {f.size} anchors differ; first at {i} (task {task}): "
                             f"f {f[i]} vs {f_ref[i]}, p {p[i]} vs {p_ref[i]}")


def respan_q(rng, task, max_dq, mode):
    """A copy of one task (uint64 [n, 2]) with its query positions moved so that the compact x / q ring of the tile kernel (chain_dp_tile.h, Lds<>:
    differences of the low 16 bits) meets its corners; x and the order of the anchors stay.  The ring is exact while the task's q values span at
    most 65535 - max_dq; the prepass sends every other task to the 32-bit ring.
      1: every q shifted by one large constant (the span stays: still compact, the low halves wrap)
      2: multiples of 65536 added to random anchors (differences that alias mod 2^16: must be recognised as a wide task)
      3: one anchor moved so that the span is exactly 65535 - max_dq (the last compact value);   4: one more (the first wide one)
      5: q values spread over about 60 000 (compact for the usual max_dq, pairs with dq just below / above 2^16 - max_dq)
      6: multiples of 2^24 added to random anchors;   7: multiples of 65536 up to 2^24 (wide tasks of every size, as 2)"""
    t = np.array(task, dtype=np.uint64).reshape(-1, 2).copy()
    if t.shape[0] == 0 or mode == 0:
        return t
    q = (t[:, 1] & np.uint64(0xffffffff)).astype(np.int64)
    hi = t[:, 1] & np.uint64(0xffffffff00000000)
    bound = 65535 - max(int(max_dq), 0)
    if mode == 1:
        q = q + int(rng.integers(1 << 16, (1 << 31) - int(q.max()) - 1))
    elif mode == 2:
        q = q + 65536 * rng.integers(0, 3, q.shape[0]) * (rng.random(q.shape[0]) < 0.3)
    elif mode == 6:
        q = q + (1 << 24) * rng.integers(0, 3, q.shape[0]) * (rng.random(q.shape[0]) < 0.3)
    elif mode == 7:
        q = q + 65536 * rng.integers(0, 250, q.shape[0]) * (rng.random(q.shape[0]) < 0.5)
    elif mode in (3, 4):
        k = int(rng.integers(0, q.shape[0]))
        q = np.minimum(q, int(q.min()) + max(bound, 0))
        q[k] = int(q.min()) + max(bound, 0) + (1 if mode == 4 else 0)
        if q.shape[0] > 1:
            q[(k + 1) % q.shape[0]] = int(q.min())
    elif mode == 5:
        q = int(q.min()) + (q - int(q.min())) % 60000 + (rng.random(q.shape[0]) < 0.2) * rng.integers(0, 60000, q.shape[0])
        q = int(q.min()) + (q - int(q.min())) % 60000
    t[:, 1] = hi | (q.astype(np.uint64) & np.uint64(0xffffffff))
    return t


def fold_driver_tasks(rng, shapes=((3000, 4, 0.05, 0.15), (2600, 3, 0.0, 0.3), (1500, 12, 0.2, 0.15), (4000, 6, 0.1, 0.05), (900, 40, 0.0, 0.2))):
    """Tasks written for the rare exits of the fold (chain.c:226-233): branching chains (query positions that follow x with jumps), runs of equal x, per-anchor
    spans, and -- where the anchors are about 3 apart in x -- windows of well over 960 anchors, i.e. beyond a ring of 16 tiles.  shapes: (anchors, largest step
    in x, share of anchors with the x of their predecessor, share of query jumps).  Used with max_skip 1 / 3 / 25 by the GPU parity test of the same name's
    family and by the CPU test that keeps the model's label counters warm."""
    tasks = []
    for n, step_hi, dup, jump in shapes:
        step = np.where(rng.random(n) < dup, 0, rng.integers(1, step_hi + 1, n))
        pos = (1 << 22) + np.cumsum(step)
        q = 50 + np.cumsum(np.where(rng.random(n) < jump, rng.integers(-300, 300, n), rng.integers(0, 2 * step_hi, n)))
        span = np.where(rng.random(n) < 0.7, 15, rng.integers(8, 40, n))
        x = (np.uint64(1) << np.uint64(32)) | pos.astype(np.uint64)
        y = (span.astype(np.uint64) << np.uint64(32)) | (np.maximum(q, 1).astype(np.uint64) & np.uint64(0xffffffff))
        o = np.argsort(x, kind="stable")
        tasks.append(np.stack((x[o], y[o]), 1))
    return tasks
