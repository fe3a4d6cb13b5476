#!/usr/bin/env python3
"""throughput of the seed-hit path (matches -> sorted anchors) and of the whole device pipeline matches -> anchors -> f/p -> chains.
The matches are derived from the bench's synthetic anchor stream (one match per query position, its hits = the anchors at that position),
so the anchors the GPU produces are that stream again and the DP that follows is the headline workload.
usage: python tools/seed_probe.py [n_reads] [anchors_per_read] [profile]
       python tools/seed_probe.py [n_reads] --real-like [genome_mb]   (matches of simulated reads on a synthetic genome, computed on the spot by
       oracle/_ref/seed_dump = the reference's own sketch/index objects; needs the prebuilt oracle/_ref)
       --heap: the order of collect_seed_hits_heap (MM_F_HEAP_SORT) among equal x instead of radix_sort_128x's"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "minimap2-fpga_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import mm2chain
from mm2chain import params, synth
import oracle_binding as ob

args_pos = [a for a in sys.argv[1:] if not a.startswith("--")]
n_reads = int(args_pos[0]) if len(args_pos) > 0 else 16384
per = int(args_pos[1]) if len(args_pos) > 1 else 5000
profile = args_pos[2] if len(args_pos) > 2 else "mixed"
distinct = min(n_reads, 512)
QLEN = 1 << 20
real_like = "--real-like" in sys.argv
heap = "--heap" in sys.argv
if not real_like:
    off1, a1 = synth.make_stream(profile, distinct, (per, per), seed=5)
    off1 = off1.numpy(); a1 = a1.numpy().view(np.uint64)


def matches_of(a):
    return synth.matches_from_anchors(a, QLEN)


def real_like_reads(genome_mb, n):
    import struct, subprocess, tempfile
    tmp = tempfile.mkdtemp()
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_synth_genome.py"), os.path.join(tmp, "syn"), "--genome-mb", str(genome_mb),
                           "--reads", str(n), "--seed", "7"], stdout=subprocess.DEVNULL)
    subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "seed_dump"), os.path.join(tmp, "syn.ref.fa"), os.path.join(tmp, "syn.reads.fa"),
                           os.path.join(tmp, "s.bin")], stderr=subprocess.DEVNULL)
    raw = open(os.path.join(tmp, "s.bin"), "rb").read()
    pos, out = 0, []
    while pos < len(raw):
        qlen, n_m = struct.unpack_from("<ii", raw, pos); pos += 8
        rec = np.frombuffer(raw, dtype=np.uint32, count=4 * n_m, offset=pos).reshape(n_m, 4); pos += 16 * n_m
        tot = int(rec[:, 0].sum())
        hits = np.frombuffer(raw, dtype=np.uint64, count=tot, offset=pos).copy(); pos += 8 * tot
        m = np.zeros(n_m, ob.MATCH_DTYPE)
        m["n"], m["q_pos"], m["q_span"], m["seg_tandem"] = rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3]
        m["cr_off"] = np.concatenate([[0], np.cumsum(rec[:, 0].astype(np.int64))[:-1]])
        out.append((qlen, m, hits))
    return out


ms, hs, mo, ao, qlens = [], [], [0], [0], []
if real_like:
    gmb = int(args_pos[1]) if len(args_pos) > 1 else 50
    for qlen, m, h in real_like_reads(gmb, distinct):
        m = m.copy(); m["cr_off"] += ao[-1]
        ms.append(m); hs.append(h); mo.append(mo[-1] + m.size); ao.append(ao[-1] + h.size); qlens.append(qlen)
    distinct = len(ms); profile = f"real-like ({gmb} Mb synthetic genome)"
else:
    for k in range(distinct):
        m, h = matches_of(a1[off1[k]:off1[k + 1]])
        m["cr_off"] += ao[-1]
        ms.append(m); hs.append(h); mo.append(mo[-1] + m.size); ao.append(ao[-1] + h.size); qlens.append(QLEN)
m1, h1 = np.concatenate(ms), np.concatenate(hs)
times = max(1, n_reads // distinct)
n_reads = distinct * times
mt = np.tile(m1, times)
mt["cr_off"] += np.repeat(np.arange(times, dtype=np.int64) * h1.size, m1.size)
ht = np.tile(h1, times)
mo = np.concatenate([[0], (np.tile(np.diff(mo), times)).cumsum()]).astype(np.int64)
ao = np.concatenate([[0], (np.tile(np.diff(ao), times)).cumsum()]).astype(np.int64)
total = int(ao[-1])
print(f"{profile}: {n_reads} reads, {mt.size} matches ({mt.size / n_reads:.0f} per read), {total} anchors")

mm2chain.init()
P = params.map_ont()
d_m = torch.from_numpy(mt.view(np.uint8)).cuda(); d_h = torch.from_numpy(ht.view(np.int64)).cuda()
d_q = torch.from_numpy(np.tile(np.array(qlens, np.int32), times)).cuda()
sp = mm2chain.SeedPlan(mo, ao); cp = mm2chain.ChainPlan(P, ao)
if heap:
    sp.set_heap_sort(True)
d_a = torch.empty((total, 2), dtype=torch.int64, device="cuda")
d_f = torch.empty(total, dtype=torch.int32, device="cuda"); d_p = torch.empty_like(d_f)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sp.run(d_m, d_h, d_q, d_a)
    torch.cuda.synchronize(); t_seed = time.perf_counter() - t0
    cp.run(d_a, d_f, d_p)
    u_off, u, b_off, b = cp.chains(d_a, d_f, d_p, 3, 40)
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
n_ties = sp.check()
# the first reads against the oracle
ok = True
got = d_a[: int(ao[4])].cpu().numpy().view(np.uint64)
for k in range(4):
    mk = ms[k].copy(); mk["cr_off"] -= ao[k]
    ok = ok and np.array_equal(got[ao[k]:ao[k + 1]], ob.collect_seed_hits(mk, hs[k], qlens[k], heap=heap))
# CPU baseline beside it: the oracle's collect_seed_hits (expansion + radix_sort_128x restated) on one host core, distinct reads only
t0 = time.perf_counter(); n_cpu = 0
for k in range(min(distinct, 256)):
    mk = ms[k].copy(); mk["cr_off"] -= ao[k]
    n_cpu += ob.collect_seed_hits(mk, hs[k], qlens[k]).shape[0]
t_cpu = time.perf_counter() - t0
print(f"CPU oracle, 1 thread (incl. the ctypes call per read): {n_cpu / t_cpu / 1e6:.1f} M anchors/s")
print(f"seed hits -> anchors: {sp.last_ms():.2f} ms (wall {t_seed*1e3:.2f}) = {total / (sp.last_ms()*1e-3) / 1e9:.2f} G anchors/s; reads with equal x: {n_ties} of {n_reads}; first reads equal the oracle: {ok}")
print(f"matches -> anchors -> f/p -> chains on the device: {t_all*1e3:.2f} ms = {total / t_all / 1e9:.3f} G anchors/s (DP {cp.last_kernel_ms():.2f} + prepass {cp.last_prepass_ms():.2f} + epilogue {cp.last_epilogue_ms():.2f} ms); "
      f"input {mt.nbytes / total:.1f} B of matches + {ht.nbytes / total:.1f} B of hits per anchor")
mm2chain.shutdown()
