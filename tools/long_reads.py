#!/usr/bin/env python3
"""Long reads (BASELINE config 5's regime, SURVEY 8 a1: n = 1e5 .. 1e6 anchors per task, chain_hardware.h:62-64 admits 5 187 500): the chaining DP and the
seed-hit path on batches that have FEWER tasks than the GPU has wave slots, at the anchor density of the bench's ava-ont stream (20 000 anchors in a
400 kb locus = 50 per kb: locus = 20 x anchors per read; -x ava-ont scalars, options.c:83-86).

usage: python tools/long_reads.py [--sizes 2048x100000,1024x300000,256x1000000] [--routes auto,one-wave,coop16,coop8] [--no-seed] [--distinct N] [--profile mixed]
Every route's f / p of the first distinct reads are compared with the CPU oracle; one line per (size, route).  bench.py imports measure() for its `long_reads` leg."""
import argparse
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "minimap2-fpga_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np   # noqa: E402
import torch         # noqa: E402
import mm2chain      # noqa: E402
from mm2chain import params, synth   # noqa: E402
import oracle_binding as ob          # noqa: E402

ROUTES = {   # tuning knobs of a route (the library's defaults are route 'auto': chosen per run, on the device when long tasks are cut first)
    "auto": {"coop_plans": 2},
    "one-wave": {"coop_plans": 0},
    "coop": {"coop_plans": 1, "coop_max_tasks": 1 << 30},                             # several waves per piece always; sixteen or eight by the number of pieces
    "coop16": {"coop_plans": 1, "coop_max_tasks": 1 << 30, "coop_w8_above": 1 << 30},
    "coop8": {"coop_plans": 1, "coop_max_tasks": 1 << 30, "coop_w8_above": 0},
}
QLEN = 1 << 26


def set_route(name):
    mm2chain.tune("coop_max_tasks", 1024); mm2chain.tune("coop_w8_above", 256)
    for k, v in ROUTES[name].items():
        mm2chain.tune(k, v)


def measure(reads, per, routes=("auto",), dp=True, seed=True, reps=3, check=2, profile="mixed", distinct=0, say=None):
    """one size: `reads` reads of `per` anchors (a few distinct ones, replicated into separate memory).  Returns a dict: per route the DP kernel + prepass ms, anchors/s,
    what ran and whether f / p of the first `check` reads equal the oracle; the seed-hit path's ms, anchors/s and whether its anchors equal the oracle's."""
    say = say or (lambda *a: None)
    P = params.ava_ont()
    distinct = distinct or max(2, min(reads, 3_200_000 // per))
    times = max(1, reads // distinct)
    off1, a1 = synth.make_stream(profile, distinct, per, seed=11, device="cuda", locus=20 * per)
    off, a = synth.replicate(off1, a1, times)
    total = int(off[-1]); n_tasks = off.numel() - 1
    a1_h = a1.cpu().numpy().view(np.uint64); off1_h = off1.numpy()
    n_chk = min(check, distinct)
    end = int(off1_h[n_chk])
    out = {"reads": n_tasks, "anchors_per_read": per, "anchors": total, "distinct_reads": distinct, "verified_reads": n_chk}
    t0 = time.perf_counter()
    f_ref, p_ref, _ = ob.chain_batch(P, off1_h[: n_chk + 1], a1_h[:end], min(8, os.cpu_count() or 1))
    t_cpu = time.perf_counter() - t0
    say(f"== {n_tasks} reads x {per} anchors = {total} anchors ({distinct} distinct); oracle on {n_chk} reads: {end / t_cpu / 1e6:.2f} M anchors/s on {min(8, n_chk)} thread(s)")
    if dp:
        d_f = torch.empty(total, dtype=torch.int32, device="cuda"); d_p = torch.empty_like(d_f)
        out["dp"] = {}
        for route in routes:
            set_route(route)
            plan = mm2chain.ChainPlan(P, off.numpy())
            ms = []
            for _ in range(reps):
                d_f.fill_(-7); d_p.fill_(-7)
                plan.run(a, d_f, d_p)
                torch.cuda.synchronize()
                ms.append((plan.last_kernel_ms(), plan.last_prepass_ms()))
            k_ms, pre_ms = min(ms, key=sum)                                    # the best run as a whole (a cold first run has a slow prepass)
            ok = bool(np.array_equal(d_f[:end].cpu().numpy(), f_ref) and np.array_equal(d_p[:end].cpu().numpy(), p_ref))
            ok = ok and bool(torch.equal(d_f[total - int(off1[-1]):], d_f[: int(off1[-1])])) and bool(torch.equal(d_p[total - int(off1[-1]):], d_p[: int(off1[-1])]))
            pieces, one_wave, coop = plan.last_route()
            out["dp"][route] = {"kernel_ms": round(k_ms, 3), "prepass_ms": round(pre_ms, 3), "value": total / ((k_ms + pre_ms) * 1e-3), "unit": "anchors/s",
                                "pieces": pieces, "pieces_one_wave_each": one_wave, "pieces_several_waves_each": coop, "identical_to_oracle": ok}
            say(f"DP {route:9s}: kernel {k_ms:9.2f} ms + prepass {pre_ms:6.2f} ms = {total / ((k_ms + pre_ms) * 1e-3) / 1e9:6.3f} G anchors/s  "
                f"(all runs {[round(m[0], 2) for m in ms]})  identical to the oracle: {ok}  pieces {pieces}: {one_wave} x 1 wave, {coop} x 16 waves")
            plan.close()
        set_route("auto")
        del d_f, d_p
    if seed:
        # matches -> sorted anchors: matches derived from the same reads (one match per query position, hits = the anchors at that position)
        ms_, hs_, mo_, ao_ = [], [], [0], [0]
        for k in range(distinct):
            m_k, h_k = synth.matches_from_anchors(a1_h[off1_h[k]:off1_h[k + 1]], QLEN)
            m_k["cr_off"] += ao_[-1]
            ms_.append(m_k); hs_.append(h_k); mo_.append(mo_[-1] + m_k.size); ao_.append(ao_[-1] + h_k.size)
        m1_, h1_ = np.concatenate(ms_), np.concatenate(hs_)
        mt_ = np.tile(m1_, times); mt_["cr_off"] += np.repeat(np.arange(times, dtype=np.int64) * h1_.size, m1_.size)
        mo_t = np.concatenate([[0], np.tile(np.diff(mo_), times).cumsum()]).astype(np.int64)
        ao_t = np.concatenate([[0], np.tile(np.diff(ao_), times).cumsum()]).astype(np.int64)
        sp = mm2chain.SeedPlan(mo_t, ao_t)
        d_m = torch.from_numpy(mt_.view(np.uint8)).cuda(); d_h = torch.from_numpy(np.tile(h1_, times).view(np.int64)).cuda()
        d_q = torch.full((n_tasks,), QLEN, dtype=torch.int32, device="cuda")
        d_as = sp.run(d_m, d_h, d_q)
        sms = []
        for _ in range(reps):
            d_as = sp.run(d_m, d_h, d_q, d_as)
            torch.cuda.synchronize()
            sms.append(sp.last_ms())
        n_ties = sp.check()
        got = d_as[: int(ao_[n_chk])].cpu().numpy().view(np.uint64)
        ok_s = True
        for k in range(n_chk):
            mk = ms_[k].copy(); mk["cr_off"] -= ao_[k]
            ok_s = ok_s and np.array_equal(got[ao_[k]:ao_[k + 1]], ob.collect_seed_hits(mk, hs_[k], QLEN))
        s_ms = min(sms)
        out["seed_hits"] = {"ms": round(s_ms, 3), "value": int(ao_t[-1]) / (s_ms * 1e-3), "unit": "anchors/s", "reads_with_equal_x": int(n_ties), "identical_to_oracle": bool(ok_s)}
        say(f"seed hits -> sorted anchors: {s_ms:9.2f} ms = {int(ao_t[-1]) / (s_ms * 1e-3) / 1e9:6.3f} G anchors/s  (all runs {[round(m, 2) for m in sms]})  "
            f"reads with equal x: {n_ties} of {n_tasks}; identical to the oracle: {bool(ok_s)}")
        sp.close(); del d_m, d_h, d_as, d_q
    del a, a1
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="2048x100000,1024x300000,256x1000000")
    ap.add_argument("--routes", default="auto,one-wave,coop16,coop8")
    ap.add_argument("--profile", default="mixed")
    ap.add_argument("--distinct", type=int, default=0, help="distinct reads generated per size (0: about 3.2e6 anchors' worth, at least 2)")
    ap.add_argument("--no-seed", action="store_true")
    ap.add_argument("--no-dp", action="store_true")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--check", type=int, default=2, help="reads per size compared with the oracle")
    args = ap.parse_args()
    mm2chain.init()
    P = params.ava_ont()
    print(f"# long reads: ava-ont scalars (max_dist {P.max_dist_x}, bw {P.bw}, max_iter {P.max_iter}, max_skip {P.max_skip}), profile {args.profile}, locus = 20 x anchors per read")
    for spec in args.sizes.split(","):
        reads, per = (int(v) for v in spec.split("x"))
        measure(reads, per, routes=args.routes.split(","), dp=not args.no_dp, seed=not args.no_seed, reps=args.reps, check=args.check, profile=args.profile,
                distinct=args.distinct, say=lambda *a: print(*a, flush=True))
    mm2chain.shutdown()


if __name__ == "__main__":
    main()
