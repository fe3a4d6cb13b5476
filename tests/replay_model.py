"""NumPy-free models of one pass of klib's radix sort (ksort.h:101-151, rs_sort) over a bucket whose records have the digits `dig`:
   literal(dig, K)  -- the reference's cycle-leader loop (ksort.h:117-131) on (id, digit) pairs, statement by statement;
   walk(dig, K)     -- the restatement csrc/radix_replay.h is built on: every bucket is a queue of its original occupants with one cursor,
                       the walk takes the occupant at the cursor of the bucket it stands at and goes where that record goes; it reads digits
                       only and returns moved[] (moved[destination] = source);
   asm_walk(dig, K) -- the same with the data layout and control flow of the hand-written loop (replay_walk: cells {address of the cursor's
                       digit, digit}, links to the next non-empty bucket, the head test on (cell, cursor) pairs).
Test infrastructure (tests/test_cpu_oracle.py); the GPU kernels are checked against the oracle's radix_sort_128x, not against this file."""


def _bounds(dig, K):
    cnt = [0] * K
    for d in dig:
        cnt[d] += 1
    lo = [0] * (K + 1)
    for d in range(K):
        lo[d + 1] = lo[d] + cnt[d]
    return cnt, lo


def literal(dig, K):
    n = len(dig)
    ids, dg = list(range(n)), list(dig)
    cnt, lo = _bounds(dig, K)
    cur, hi = lo[:K], lo[1:]
    k = 0
    while k < K:                                          # ksort.h:117
        if cur[k] != hi[k]:                               # :118
            l = dg[cur[k]]
            if l != k:                                    # :120
                tmp = (ids[cur[k]], dg[cur[k]])
                while True:                               # :122-126
                    swap = tmp
                    tmp = (ids[cur[l]], dg[cur[l]])
                    ids[cur[l]], dg[cur[l]] = swap
                    cur[l] += 1
                    l = tmp[1]
                    if l == k:
                        break
                ids[cur[k]], dg[cur[k]] = tmp             # :127
                cur[k] += 1
            else:
                cur[k] += 1                               # :128
        else:
            k += 1                                        # :129
    return ids


def walk(dig, K):
    n = len(dig)
    cnt, lo = _bounds(dig, K)
    cur = lo[:K]
    head = min(d for d in range(K) if cnt[d])
    c, moved = head, [None] * n
    for _ in range(n):
        p = cur[c]
        if c == head and p == lo[c + 1]:                  # the head's queue is exhausted: the next bucket that is not takes over
            head = min(d for d in range(K) if cur[d] != lo[d + 1])
            c, p = head, cur[head]
        assert p < lo[c + 1]                              # no other bucket can be exhausted on arrival
        d = dig[p]
        cur[c] = p + 1
        dst = cur[d] - (1 if d == head else 0)            # the head receives one place before its cursor
        assert moved[dst] is None
        moved[dst] = p
        c = d
    return moved


def asm_walk(dig, K, dg_addr=1000):
    n = len(dig)
    cnt, lo = _bounds(dig, K)
    mem = {dg_addr + i: dig[i] for i in range(n)}
    rd = lambda a: mem.get(a, 77 % K)                     # the byte behind the bucket: anything
    cell_p = [dg_addr + lo[d] for d in range(K)]
    cell_d = [rd(dg_addr + lo[d]) for d in range(K)]
    link = [0] * K
    for d in range(K):
        for e in range(d + 1, K):
            if cnt[e]:
                link[d] = e
                break
    h = min(d for d in range(K) if cnt[d])
    moved, steps, state = {}, 0, "load"
    while True:
        if state == "adv":                                # label 3 of the assembly
            h = link[h]
            if h == 0:
                break
            state = "load"
        if state == "load":                               # label 5
            hc, c, P, D, he = h, h, cell_p[h], cell_d[h], dg_addr + lo[h + 1]
            if P == he:
                state = "adv"
                continue
            state = "step"
        if (c, P) == (hc, he):                            # v_cmp_eq_u64 on the pairs
            state = "adv"
            continue
        d, nx, pd_p, pd_d = D, rd(P + 1), cell_p[D], cell_d[D]
        src, same = P - dg_addr, D == c
        cell_p[c], cell_d[c] = P + 1, nx
        n_p, n_d = (P + 1, nx) if same else (pd_p, pd_d)
        t = n_p - (1 if d == hc else 0)
        assert t - dg_addr not in moved
        moved[t - dg_addr] = src
        c, P, D = d, n_p, n_d
        steps += 1
        assert steps <= n
    return [moved[i] for i in range(n)]
