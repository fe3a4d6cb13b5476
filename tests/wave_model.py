"""NumPy model of the HIP kernel's control flow (csrc/chain_kernel.hip), lane for lane: 64-wide chunks scanned
nearest-first, chunk 0 from a shifted register window, older chunks from an LDS ring of R anchors, look-back
beyond the ring from "global" arrays, stamps restricted to the window, prefix-max + Lindley scan for max_skip.
It exists so that the wave-parallel formulation can be checked against the oracle WITHOUT a GPU (-m "not gpu");
the GPU tests check the real kernel."""
import numpy as np

INT_MIN = -(2**31)


def _pair_score(P, avg, dr, dq, same, span_i):
    dd = np.where(dr > dq, dr - dq, dq - dr).astype(np.int64)
    ok = ~((same & (dr == 0)) | (dq <= 0))
    ok &= ~((same & (dq > P.max_dist_y)) | (dq > P.max_dist_x))
    ok &= ~(same & (dd > P.bw))
    if P.n_segs > 1 and not P.is_cdna:
        ok &= ~(same & (dr > P.max_dist_y))
    s = np.minimum(np.minimum(dq, dr), span_i).astype(np.int64)
    lg = np.where(dd > 0, np.floor(np.log2(np.maximum(dd, 1))).astype(np.int64), 0)
    lin = (dd.astype(np.float32) * np.float32(avg)).astype(np.int64)      # f32 multiply, truncation
    if P.is_cdna:
        gap = np.where(~same & (dr == 0), 0, np.where((dr > dq) | ~same, np.minimum(lin, lg), lin + (lg >> 1)))
        s = s + (~same & (dr == 0))
    else:
        gap_diff = np.where(dr == 0, 0, np.minimum(lin, lg))
        gap = np.where(same, lin + (lg >> 1), gap_diff)
        s = s + (~same & (dr == 0))
    pen = (gap.astype(np.float64) * np.float64(np.float32(P.gap_scale)) + .499).astype(np.int64)
    return ok, s - pen


def chain_wave_model(P, anchors, avg, R=256, span_override=-1, ignore_seg=False, stats=None):
    a = np.ascontiguousarray(anchors).view(np.uint64).reshape(-1, 2)
    n = a.shape[0]
    xlo = (a[:, 0] & 0xFFFFFFFF).astype(np.int64); xhi = (a[:, 0] >> 32).astype(np.int64)
    q = (a[:, 1] & 0xFFFFFFFF).astype(np.uint32).view(np.int32).astype(np.int64)
    span = ((a[:, 1] >> 32) & 0xff).astype(np.int64); seg = ((a[:, 1] >> 48) & 0xff).astype(np.int64)
    if ignore_seg:
        seg = np.zeros_like(seg)
    f = np.zeros(n, np.int64); p = np.full(n, -1, np.int64)
    far = P.max_iter + 64 > R
    s_x = np.zeros(R, np.int64); s_q = np.zeros(R, np.int64); s_f = np.zeros(R, np.int64); s_p = np.full(R, -1, np.int64)
    s_g = np.zeros(R, np.int64); s_t = np.zeros(R, np.int64)
    t_glob = np.zeros(n, np.int64)
    f_glob = np.zeros(n, np.int64); p_glob = np.full(n, -1, np.int64)     # what has been stored to HBM so far
    lane = np.arange(64)
    wx = np.zeros(64, np.int64); wq = np.zeros(64, np.int64); wf = np.zeros(64, np.int64); wp = np.full(64, -1, np.int64)
    wg = np.zeros(64, np.int64)
    D = P.max_dist_x
    run_hi, hs = 0, 0
    n_chunks = 0; n_far = 0
    for i0 in range(0, n, 64):
        cnt = min(64, n - i0)
        idx = i0 + lane
        m = idx < n
        s_x[idx[m] % R] = xlo[idx[m]]; s_q[idx[m] % R] = q[idx[m]]; s_g[idx[m] % R] = seg[idx[m]]
        lds_lo = i0 + 64 - R
        for k in range(cnt):
            i = i0 + k
            xi, qi, sgi = xlo[i], q[i], seg[i]
            sp_i = span_override if span_override >= 0 else span[i]
            if i == 0 or xhi[i] != run_hi:
                run_hi, hs = xhi[i], i
            lo = max(hs, max(i - P.max_iter, 0))
            best, best_j, n_skip = sp_i, -1, 0
            jtop = i - 1
            c = 0
            more, broke = True, False
            while jtop >= lo and more and not broke:
                j = jtop - lane
                if c == 0:
                    xj, qj, fj, pj, gj = wx, wq, wf, wp, wg
                else:
                    xj = np.zeros(64, np.int64); qj = xj.copy(); fj = xj.copy(); pj = np.full(64, -1, np.int64); gj = xj.copy()
                    near = (j >= lds_lo) if far else np.ones(64, bool)
                    sl = j % R
                    xj[near] = s_x[sl[near]]; qj[near] = s_q[sl[near]]; fj[near] = s_f[sl[near]]; pj[near] = s_p[sl[near]]
                    gj[near] = s_g[sl[near]]
                    fr = ~near & (j >= lo)
                    if fr.any():
                        n_far += 1
                        xj[fr] = xlo[j[fr]]; qj[fr] = q[j[fr]]; gj[fr] = seg[j[fr]]
                        fj[fr] = f_glob[j[fr]]; pj[fr] = p_glob[j[fr]]
                n_chunks += 1
                dr = (xi - xj) & 0xFFFFFFFF
                inwin = (j >= lo) & (dr <= D)
                more = bool(inwin.all())
                dq = ((qi - qj + 2**31) % 2**32) - 2**31
                ok, sc = _pair_score(P, avg, dr, dq, gj == sgi, sp_i)
                valid = ok & inwin
                sc = sc + fj
                scv = np.where(valid, sc, INT_MIN)
                incl = np.maximum.accumulate(scv)
                last = 63
                if P.max_skip < P.max_iter:
                    stamp = i + 1
                    do_mark = valid & (pj >= lo)
                    for L in np.nonzero(do_mark)[0]:
                        if (not far) or pj[L] >= lds_lo:
                            s_t[pj[L] % R] = stamp
                        else:
                            t_glob[pj[L]] = stamp
                    tj = np.zeros(64, np.int64)
                    nearj = (j >= lds_lo) if far else np.ones(64, bool)
                    tj[nearj] = s_t[j[nearj] % R]
                    fr = ~nearj & inwin
                    tj[fr] = t_glob[j[fr]]
                    marked = tj == stamp
                    excl = np.concatenate(([INT_MIN], incl[:-1]))
                    nm = valid & (sc > np.maximum(best, excl))
                    se = valid & ~nm & marked
                    if se.any():
                        S = np.cumsum(se.astype(np.int64) - nm.astype(np.int64))
                        nl = S + np.maximum(n_skip, np.maximum.accumulate(-S))
                        brk = se & (nl > P.max_skip)
                        if brk.any():
                            last = int(np.argmax(brk)) - 1
                            broke = True
                        else:
                            n_skip = int(nl[63])
                    else:
                        n_skip = max(n_skip - int(nm.sum()), 0)
                if last >= 0:
                    mc = int(incl[last])
                    if mc > best:
                        best = mc
                        best_j = jtop - int(np.argmax(valid & (sc == mc)))
                jtop -= 64
                c += 1
            f[i], p[i] = best, best_j
            s_f[i % R], s_p[i % R] = best, best_j
            wx = np.concatenate(([xi], wx[:-1])); wq = np.concatenate(([qi], wq[:-1]))
            wf = np.concatenate(([best], wf[:-1])); wp = np.concatenate(([best_j], wp[:-1])); wg = np.concatenate(([sgi], wg[:-1]))
        f_glob[i0:i0 + cnt] = f[i0:i0 + cnt]; p_glob[i0:i0 + cnt] = p[i0:i0 + cnt]
    if stats is not None:
        stats["chunks"] = n_chunks; stats["far_chunks"] = n_far
    return f.astype(np.int32), p.astype(np.int32)
