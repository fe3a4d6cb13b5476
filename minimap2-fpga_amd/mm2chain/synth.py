"""Deterministic synthetic anchor streams (BASELINE.json configs 2, 4', 5').

Every random draw is splitmix64(seed, stream, global anchor index), evaluated with torch int64 arithmetic, so the
same (profile, seed, sizes) gives bit-identical anchors on CPU and on the GPU.  Anchors follow the reference's
encoding (minimap.h:53, map.c:232-241): x = strand<<63 | rid<<32 | rpos (rpos < 2^31), y = q_span<<32 | qpos,
segment id 0, each task sorted ascending by x (map.c:245).
"""
import torch

_M = (1 << 64) - 1


def _s64(v):
    """python int (mod 2^64) -> value representable in torch int64"""
    v &= _M
    return v - (1 << 64) if v >= (1 << 63) else v


def _lsr(x, s):
    return (x >> s) & ((1 << (64 - s)) - 1)


def splitmix64(idx, seed, stream):
    """idx: int64 tensor of counters.  Returns uniform int64 bit patterns (counter-based, wraps mod 2^64)."""
    z = idx * _s64(0x9E3779B97F4A7C15) + _s64(seed * 0xD1342543DE82EF95 + (stream + 1) * 0xA0761D6478BD642F)
    z = (z ^ _lsr(z, 30)) * _s64(0xBF58476D1CE4E5B9)
    z = (z ^ _lsr(z, 27)) * _s64(0x94D049BB133111EB)
    return z ^ _lsr(z, 31)


def _uniform(idx, seed, stream, lo, hi):
    """integers in [lo, hi)"""
    return lo + (_lsr(splitmix64(idx, seed, stream), 11) % (hi - lo))


PROFILES = ("sparse", "mixed", "dense", "colinear")


def make_stream(profile, n_reads, n_per_read=5000, seed=1, q_span=15, device="cpu", locus=None):
    """Returns (offsets int64 cpu [n_reads+1], anchors int64 [total, 2] on `device`).

    n_per_read: int (fixed) or (lo, hi) for n ~ U[lo, hi].  Profiles:
      sparse   -- anchors uniform over 24 references x 2^31 positions, both strands (windows mostly empty)
      mixed    -- per read one locus (~100 kb): 35 % a colinear chain (steps U[5,35], query jitter +-2), 15 % copies of
                  chain positions shifted by one of two per-read repeat offsets, 50 % uniform noise in the locus
      dense    -- as mixed without repeats, 50 % chain / 50 % noise, everything inside a 45 kb locus
      colinear -- the chain only
    """
    assert profile in PROFILES
    dev = torch.device(device)
    rd = torch.arange(n_reads, dtype=torch.int64, device=dev)
    if isinstance(n_per_read, int):
        n_r = torch.full((n_reads,), n_per_read, dtype=torch.int64, device=dev)
    else:
        n_r = _uniform(rd, seed, 0, int(n_per_read[0]), int(n_per_read[1]) + 1)
    offsets = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
    offsets[1:] = torch.cumsum(n_r, 0)
    total = int(offsets[-1])
    task = torch.repeat_interleave(rd, n_r, output_size=total)
    g = torch.arange(total, dtype=torch.int64, device=dev)       # global anchor counter

    if profile == "sparse":
        strand = _uniform(g, seed, 1, 0, 2)
        rid = _uniform(g, seed, 2, 0, 24)
        rpos = _uniform(g, seed, 3, 0, 1 << 31)
        qpos = _uniform(g, seed, 4, q_span, 10000)
    else:
        span_ref = {"mixed": 100000, "dense": 45000, "colinear": 100000}[profile] if locus is None else locus
        strand = _uniform(rd, seed, 1, 0, 2)[task]
        rid = _uniform(rd, seed, 2, 0, 24)[task]
        # (a locus longer than ~4 Mb -- the long-read streams of tools/long_reads.py -- must still end below 2^31: the bound moves down for those only)
        start_hi = (1 << 31) - (1 << 22) if span_ref + 40000 < (1 << 22) else (1 << 31) - span_ref - (1 << 20)
        start = _uniform(rd, seed, 3, 1 << 20, start_hi)[task]
        kind = _uniform(g, seed, 5, 0, 100)
        if profile == "mixed":
            is_chain, is_rep = kind < 35, (kind >= 35) & (kind < 50)
        elif profile == "dense":
            is_chain, is_rep = kind < 50, torch.zeros_like(kind, dtype=torch.bool)
        else:
            is_chain, is_rep = torch.ones_like(kind, dtype=torch.bool), torch.zeros_like(kind, dtype=torch.bool)
        step = torch.where(is_chain, _uniform(g, seed, 6, 5, 36), torch.zeros_like(g))
        jit = torch.where(is_chain, _uniform(g, seed, 7, -2, 3), torch.zeros_like(g))
        cs_r = torch.cumsum(step, 0)
        cs_q = torch.cumsum(step + jit, 0)
        base_r = (cs_r - step)[offsets[:-1].clamp(max=max(total - 1, 0))][task] if total else cs_r
        base_q = (cs_q - step - jit)[offsets[:-1].clamp(max=max(total - 1, 0))][task] if total else cs_q
        chain_r = cs_r - base_r                                   # chain coordinate reached at this anchor
        chain_q = cs_q - base_q
        # chains longer than the locus wrap around inside it (keeps dense loci dense)
        chain_len = span_ref - 1000
        rep_off = torch.where(_uniform(g, seed, 8, 0, 2) == 0, _uniform(rd, seed, 9, 2000, 20000)[task],
                              -_uniform(rd, seed, 10, 2000, 20000)[task])
        qmax = torch.clamp(chain_q.new_tensor(0) + (cs_q - base_q)[(offsets[1:] - 1).clamp(min=0)][task], min=2000) + q_span
        noise_r = _uniform(g, seed, 11, 0, span_ref)
        noise_q = q_span + _lsr(splitmix64(g, seed, 12), 11) % qmax
        rpos = torch.where(is_chain, 500 + chain_r % chain_len,
                           torch.where(is_rep, (500 + chain_r % chain_len + rep_off).clamp(min=0), noise_r))
        qpos = torch.where(is_chain | is_rep, q_span + chain_q, noise_q)
        rpos = (start + rpos).clamp(max=(1 << 31) - 1)
    x = (strand << 63) | (rid << 32) | rpos
    y = (torch.full_like(qpos, q_span) << 32) | (qpos & 0xFFFFFFFF)
    # sort inside each task by unsigned x: stable sort by x, then stable sort by task
    key = x ^ _s64(1 << 63)
    o1 = torch.sort(key, stable=True).indices
    o2 = torch.sort(task[o1], stable=True).indices
    perm = o1[o2]
    anchors = torch.stack((x[perm], y[perm]), dim=1).contiguous()
    return offsets.cpu(), anchors


def replicate(offsets, anchors, times):
    """Tile a batch `times` times (distinct memory, same content): used to reach 10^5-task batches quickly."""
    n = offsets[1:] - offsets[:-1]
    off = torch.zeros(n.numel() * times + 1, dtype=torch.int64)
    off[1:] = torch.cumsum(n.repeat(times), 0)
    return off, anchors.repeat(times, 1).contiguous()


def matches_from_anchors(a, qlen=1 << 20):
    """anchors of one read (uint64 [n, 2], any order) -> (matches, hits) that collect_seed_hits (map.c:215-247) expands back into
    exactly these anchors: one match per (query position, span), its hit list = the reference positions at that query position.
    Used to drive the seed-hit path with the bench's synthetic streams.  matches: numpy structured array with the fields of
    mm2c_match_t (cr_off relative to the returned hit array)."""
    import numpy as np
    dt = np.dtype([("cr_off", "<i8"), ("n", "<u4"), ("q_pos", "<u4"), ("q_span", "<u4"), ("seg_tandem", "<u4")])
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 2)
    x, y = a[:, 0], a[:, 1]
    rev = (x >> np.uint64(63)).astype(np.uint64)
    span = ((y >> np.uint64(32)) & np.uint64(0xff)).astype(np.int64)
    q = (y & np.uint64(0xffffffff)).astype(np.int64)
    qp = np.where(rev == 1, qlen - 1 - q - 1 + span, q)                       # map.c:237 inverted
    r = (x & np.uint64(0x7fffffff00000000)) | ((x & np.uint64(0xffffffff)) << np.uint64(1)) | rev   # strand bit != q_pos bit 0 <=> reverse
    key = qp * 256 + span
    order = np.lexsort((r, key))
    key, r = key[order], r[order]
    first = np.concatenate([[True], key[1:] != key[:-1]]) if key.size else np.zeros(0, bool)
    starts = np.nonzero(first)[0]
    m = np.zeros(starts.size, dt)
    m["cr_off"] = starts
    m["n"] = np.diff(np.concatenate([starts, [key.size]]))
    m["q_pos"] = (key[starts] // 256).astype(np.uint32) << 1
    m["q_span"] = key[starts] % 256
    return m, r
