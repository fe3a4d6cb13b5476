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


def gpu_batch(par, offsets, anchors, avg=None, variant=None):
    """run the HIP path through the C ABI (plan, device-resident) and return f, p as numpy; `variant`: a list that receives the text of
    mm2c_plan_last_variant (which kernel instantiation ran)"""
    import mm2chain
    a_np = np.ascontiguousarray(anchors).view(np.int64).reshape(-1, 2)
    d_a = torch.from_numpy(a_np).cuda()
    total = a_np.shape[0]
    d_f = torch.full((total,), -77, dtype=torch.int32, device="cuda")
    d_p = torch.full((total,), -77, dtype=torch.int32, device="cuda")
    d_avg = torch.from_numpy(np.asarray(avg, dtype=np.float32)).cuda() if avg is not None else None
    plan = mm2chain.ChainPlan(par, offsets)
    plan.run(d_a, d_f, d_p, d_avg)
    torch.cuda.synchronize()
    if variant is not None:
        variant.append(plan.last_variant())
    plan.close()
    return d_f.cpu().numpy(), d_p.cpu().numpy()


def assert_same(f, p, f_ref, p_ref, offsets=None, what=""):
    bad = np.nonzero((f != f_ref) | (p != p_ref))[0]
    if bad.size:
        i = int(bad[0])
        task = int(np.searchsorted(np.asarray(offsets), i, side="right") - 1) if offsets is not None else -1
        raise AssertionError(f"{what}: {bad.size} of {f.size} anchors differ; first at {i} (task {task}): "
                             f"f {f[i]} vs {f_ref[i]}, p {p[i]} vs {p_ref[i]}")
