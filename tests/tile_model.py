"""NumPy model of the control flow of the round-2 HIP kernel (csrc/chain_dp_tile.h), lane for lane: tile-aligned chunks scanned
nearest-first (own tile from "registers", NX - 1 older tiles from the x / q ring, f / p of the NF nearest from the ring and deeper ones
from the task's own stores, anything older from "global" memory), the three-instruction filter, equal-x runs found per tile, 16-bit
stamps in a ring of 64 NX slots with 32-bit stamps beyond it, and the fold paths (A: no lane beats the running best, B0: the first surviving
lane is the only new maximum, B1: a single candidate, B2: prefix max + closed-form or max-plus skip counter).  It exists so that the formulation can be checked against the oracle
WITHOUT a GPU (-m "not gpu"), and so that tools/chunk_stats.py can count how often each path of the hand-written loop is taken
(profiles/r2_isa_budget.md); the GPU tests check the real kernel.  One segment, no cDNA (the variant the hand-written loop covers)."""
import numpy as np

INT_MIN = -(2**31)


def _score(P, avg, dr1, dq1, dd, span_i):
    """chain.c:207-219 for same-segment pairs; dr1 = dr - 1, dq1 = dq - 1"""
    s = np.minimum(np.minimum(dq1, dr1), span_i - 1) + 1
    lg = np.where(dd > 0, np.floor(np.log2(np.maximum(dd, 1))).astype(np.int64), 0)
    gap = (dd.astype(np.float32) * np.float32(avg)).astype(np.int64) + (lg >> 1)
    if np.float32(P.gap_scale) != np.float32(1.0):
        gap = (gap.astype(np.float64) * np.float64(np.float32(P.gap_scale)) + .499).astype(np.int64)
    return s - gap


LABEL_KEYS = ("lk", "own_pass", "lloop", "lold", "lfg", "lpart", "part_pass", "limp", "fold_b0", "lslow2", "lslow", "fold_b1", "lb1m", "lb2", "b2_no_skip_events", "lli",
              "li_closed", "lcfb", "lgen", "lbk", "break_in_fold_a", "break_in_b0", "lend", "far_chunks", "far_pass", "far_stamp", "lspec")


def chain_tile_model(P, anchors, avg, NX=8, NF=2, span_override=-1, stats=None, compact=False):
    """compact: the x / q ring holds the low 16 bits of x and q (Lds<..., C16>), differences taken mod 2^16 and zero-extended, as the SDWA subtractions of the
    compact instantiations do (the caller checks the task's q span, as the prepass does).  stats additionally receives the sub-paths of the fold by the names of
    the assembly's labels (LABEL_KEYS; tests/test_cpu_oracle.py keeps them warm, tests/test_gpu_labels.py counts the real ones)."""
    a = np.ascontiguousarray(anchors).view(np.uint64).reshape(-1, 2)
    n = a.shape[0]
    x64 = a[:, 0].astype(np.uint64)
    xlo = (a[:, 0] & 0xFFFFFFFF).astype(np.int64)
    q = (a[:, 1] & 0xFFFFFFFF).astype(np.uint32).view(np.int32).astype(np.int64)
    span = ((a[:, 1] >> 32) & 0xff).astype(np.int64)
    f = np.zeros(n, np.int64); p = np.full(n, -1, np.int64)
    SN = 64 * NX
    s_x = np.zeros(SN, np.int64); s_q = np.zeros(SN, np.int64)                    # x / q rings: anchor j at j mod SN
    s_f = np.zeros(64 * NF, np.int64); s_p = np.full(64 * NF, -1, np.int64)       # f / p rings: anchor j at j mod 64 NF
    s_t = np.zeros(SN, np.int64)                                                  # 16-bit stamps
    t_glob = np.zeros(n, np.int64)
    # prepass (chain_window_start): st[i] = max(first j with x_i <= x_j + max_dist_x, i - max_iter), 64-bit compare
    D = np.uint64(P.max_dist_x)
    st = np.maximum(np.searchsorted(x64, np.where(x64 >= D, x64 - D, np.uint64(0)), side="left"), np.arange(n) - P.max_iter)
    max_dq = min(P.max_dist_x, P.max_dist_y)
    fast_filter = P.bw >= 0 and max_dq - 1 >= P.bw
    if P.bw < 0:
        max_dq = 0                                                                # chain.c:205: dd >= 0 > bw, nothing passes
    lane = np.arange(64)
    S = dict(anchors=0, no_window=0, own_chunks=0, own_pass=0, ring_chunks=0, ring_pass=0, deep_fp=0, far_chunks=0, far_pass=0,
             fold_a=0, fold_b0=0, fold_b1=0, fold_b2_closed=0, fold_b2_scan=0, breaks=0, eq_run_anchors=0)
    S.update(dict.fromkeys(LABEL_KEYS, 0), fold_b0=0, fold_b1=0, own_pass=0, far_chunks=0, far_pass=0)
    S.update(skip_tested=0, skip_rejects=0, skip_wrong=0, skip_missed=0)
    CM = 0xffff
    for i0 in range(0, n, 64):
        cnt = min(64, n - i0)
        stamp_lo = i0 - 64 * (NX - 1)
        idx = i0 + 63 - lane                                                      # lane L holds anchor i0 + 63 - L
        m = idx < n
        s_x[idx[m] % SN] = xlo[idx[m]] & CM if compact else xlo[idx[m]]
        s_q[idx[m] % SN] = q[idx[m]] & CM if compact else q[idx[m]]
        s_t[:] = 0                                                                # one-byte stamps: the ring is wiped per tile
        for k in range(cnt):
            i = i0 + k
            sp_i = span_override if span_override >= 0 else int(span[i])
            lo = i if max_dq <= 0 else min(int(st[i]), i)
            best, best_j, n_skip = sp_i, -1, 0
            S["anchors"] += 1
            S["lk"] += 1
            if lo >= i:
                S["no_window"] += 1
                S["lspec"] += 1
                f[i], p[i] = best, best_j
                continue
            # equal-x run that ends at i (chain.c:202 `dr == 0`): those predecessors are dropped
            e = 0
            while i - 1 - e >= lo and x64[i - 1 - e] == x64[i]:
                e += 1
            if e:
                S["eq_run_anchors"] += 1
                if e > i - i0:
                    S["lspec"] += 1                                                # the run reaches into the tile before: the C++ path takes the anchor
            s16 = 1 + (i & 63)
            broke = False
            base = i0
            while base + 63 >= lo and not broke:
                j = base + 63 - lane                                              # ascending lane = descending j = the reference's order
                inwin = (j >= lo) & (j < i)
                if not inwin.any():
                    base -= 64; continue
                own = base == i0
                ring = (not own) and base >= stamp_lo
                part = bool((j < lo).any())
                if own:
                    S["own_chunks"] += 1
                elif ring:
                    S["ring_chunks"] += 1
                    S["lpart" if part else "lloop"] += 1
                else:
                    S["far_chunks"] += 1
                jj = np.clip(j, 0, n - 1)
                xj = xlo[jj] if not ring else s_x[j % SN]
                qj = q[jj] if not ring else s_q[j % SN]
                if compact:
                    # low halves, differences mod 2^16, zero-extended: 0xffff stands for -1 (dr == 0 / dq == 0) and fails the filter like any other big value
                    dr1 = (xlo[i] - 1 - xj) & CM
                    dq1 = (q[i] - 1 - qj) & CM
                else:
                    dr1 = ((xlo[i] - 1 - xj + 2**31) % 2**32) - 2**31
                    dq1 = ((q[i] - 1 - qj + 2**31) % 2**32) - 2**31
                dd = np.abs(dr1 - dq1)
                if fast_filter:
                    u = np.maximum(np.maximum((dq1 % 2**32) - (max_dq - 1 - P.bw), 0), dd)          # v_sub clamp, v_max_u32 (dq1 < 0 wraps to huge)
                    ok = u <= P.bw
                else:
                    ok = (dq1 >= 0) & (dq1 + 1 <= max_dq) & (dd <= P.bw)
                valid = ok & inwin & (j < i - e)
                if ring:
                    # (round 6, tools/chunk_stats.py --tile-skip: would a 256-bit occupancy bitmap of the tile's diagonal buckets ((x - q) >> 9 mod 256) have rejected this
                    # visit before its 13 instructions?  A pair passes only with |dr - dq| <= bw <= 511: the candidate's bucket is the anchor's or a neighbour)
                    bj = (((xlo[jj] - q[jj]) >> 9) & 255)[j >= max(base, 0)]
                    bi = ((int(xlo[i]) - int(q[i])) >> 9) & 255
                    hit = np.isin(bj, [(bi - 1) & 255, bi, (bi + 1) & 255]).any()
                    S["skip_tested"] += 1
                    if not hit:
                        S["skip_rejects"] += 1
                        if valid.any():
                            S["skip_wrong"] += 1                               # must stay 0: a rejected tile has no lane inside the band
                    elif not valid.any():
                        S["skip_missed"] += 1
                if not valid.any():
                    base -= 64; continue
                depth = (i0 - base) // 64
                if own:
                    S["own_pass"] += 1
                    fj, pj = f[jj], p[jj]
                elif ring:
                    S["ring_pass"] += 1
                    S["part_pass" if part else "lold"] += 1
                    if depth <= NF:
                        fj, pj = s_f[j % (64 * NF)], s_p[j % (64 * NF)]
                    else:
                        S["deep_fp"] += 1
                        S["lfg"] += 1
                        fj, pj = f[jj], p[jj]
                else:
                    S["far_pass"] += 1
                    fj, pj = f[jj], p[jj]
                # stamps (chain.c:226-233): scatter by p, gather by j; 16 bits inside the ring, 32 bits beyond
                do_mark = valid & (pj >= lo)
                for L in np.nonzero(do_mark)[0]:
                    if pj[L] >= stamp_lo:
                        s_t[pj[L] % SN] = s16
                    else:
                        t_glob[pj[L]] = i + 1
                        S["far_stamp"] += 1
                if own or ring:
                    marked = valid & (s_t[j % SN] == s16)
                else:
                    marked = valid & (t_glob[jj] == i + 1)
                sc = _score(P, avg, dr1, dq1, dd, sp_i) + fj
                scv = np.where(valid, sc, INT_MIN)
                cand = scv > best
                last = 63
                if not cand.any():                                                 # fold A
                    S["fold_a"] += 1
                    n_skip += int(marked.sum())
                    if marked.any() and n_skip > P.max_skip:                       # the break of chain.c:231; nothing before it changes the best
                        broke = True
                        S["break_in_fold_a"] += 1
                    base -= 64
                    if broke:
                        S["breaks"] += 1
                    continue
                S["limp"] += 1
                l0 = int(np.argmax(valid))
                if scv[l0] > best and (scv > scv[l0]).any():
                    S["lslow2"] += 1
                if not (scv[l0] > best and not (scv > scv[l0]).any()):
                    S["lslow"] += 1
                if scv[l0] > best and not (scv > scv[l0]).any():                   # fold B0: the first surviving lane is the only new maximum
                    S["fold_b0"] += 1
                    best, best_j = int(scv[l0]), base + 63 - l0
                    n_skip = max(n_skip - 1, 0)
                    se = marked.copy(); se[l0] = False                             # every marked lane behind it is a skip event
                    n_skip += int(se.sum())
                    if se.any() and n_skip > P.max_skip:                           # chain.c:231 is only reached by a skip event
                        broke = True
                        S["breaks"] += 1
                        S["break_in_b0"] += 1
                    base -= 64
                    continue
                incl = np.maximum.accumulate(scv)
                if not marked.any() and n_skip == 0 and int(cand.sum()) == 1:      # fold B1
                    S["fold_b1"] += 1
                    L = int(np.argmax(cand))
                    best, best_j = int(scv[L]), base + 63 - L
                    base -= 64
                    continue
                if not marked.any() and n_skip == 0:
                    S["lb1m"] += 1                                                 # several candidates, nothing to count: prefix max only
                else:
                    S["lb2"] += 1
                excl = np.concatenate(([INT_MIN], incl[:-1]))
                nm = valid & (sc > np.maximum(best, excl))
                se = marked & ~nm
                if not se.any():
                    S["fold_b2_closed"] += 1
                    if marked.any() or n_skip != 0:
                        S["b2_no_skip_events"] += 1
                    n_skip = max(n_skip - int(nm.sum()), 0)
                else:
                    last_nm = 63 - int(np.argmax(nm[::-1])) if nm.any() else -1
                    first_se = int(np.argmax(se))
                    closed = last_nm < first_se and max(n_skip - int(nm.sum()), 0) + int(se.sum()) <= P.max_skip
                    S["fold_b2_closed" if closed else "fold_b2_scan"] += 1
                    S["lli"] += 1
                    if last_nm < first_se:
                        S["li_closed" if closed else "lcfb"] += 1
                    else:
                        S["lgen"] += 1
                    D = np.cumsum(se.astype(np.int64) - nm.astype(np.int64))
                    nl = D + np.maximum(n_skip, np.maximum.accumulate(-D))
                    brk = se & (nl > P.max_skip)
                    if brk.any():
                        last = int(np.argmax(brk)) - 1
                        broke = True
                        S["breaks"] += 1
                        if last_nm >= first_se:
                            S["lbk"] += 1
                    else:
                        n_skip = int(nl[63])
                if last >= 0:
                    mc = int(incl[last])
                    if mc > best:
                        best = mc
                        best_j = base + 63 - int(np.argmax(valid & (sc == mc)))
                base -= 64
            if not broke and base + 63 < lo and i0 - lo > 0:
                S["lend"] += 1                                                     # the window ran out (in the assembly: its ring part, Lend)
            f[i], p[i] = best, best_j
        sl = slice(i0, i0 + cnt)
        s_f[np.arange(i0, i0 + cnt) % (64 * NF)] = f[sl]; s_p[np.arange(i0, i0 + cnt) % (64 * NF)] = p[sl]
    if stats is not None:
        stats.update(S)
    return f.astype(np.int32), p.astype(np.int32)
