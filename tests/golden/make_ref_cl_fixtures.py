"""f[] / p[] PRODUCED BY THE REFERENCE'S OWN DEVICE KERNEL.

oracle/_ref/libref_cl_chain.so is /root/reference/device/minimap2_opencl.cl compiled, unmodified and where it lies, for the
host CPU by the image's clang OpenCL front end (recipe and the reason for each flag: oracle/ref_host/Makefile).  This script
calls its kernel entry `chain0` exactly as run_chaining_on_hw does (chain_hardware.cpp:118-146: total_subparts, max_dist_x,
max_dist_y, bw, q_span, avg_qspan_scaled, a, f, p, num_subparts) and stores inputs and outputs as a fixture.  The kernel's
host-side input num_subparts[] is what chain.c:62-78 computes; it is an INPUT here (computed by the repo's restatement
mm2o_predict and stored in the fixture), the outputs f/p are the reference's.

Only runs where /root/reference exists.  Output: tests/golden/ref_cl_kernel_fp.npz (data only).
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "minimap2-fpga_amd"))
import oracle_binding as ob  # noqa: E402
from mm2chain import synth  # noqa: E402

subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle", "ref_host"), "../_ref/libref_cl_chain.so"])
lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_cl_chain.so"))
lib.chain0.restype = None
lib.chain0.argtypes = [C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float] + [C.c_void_p] * 4
PAD = 128   # the kernel prefetches a[] / num_subparts[] 64 entries ahead of the anchor it works on (.cl:40-43,61-63)


def ref_kernel(a, max_dist_x, max_dist_y, bw, q_span, avg):
    n = a.shape[0]
    ns, total_sub, _ = ob.predict(a, max_dist_x)
    a_pad = np.zeros((n + PAD, 2), np.uint64); a_pad[:n] = a
    ns_pad = np.zeros(n + PAD, np.uint8); ns_pad[:n] = ns
    f = np.full(n + PAD, -77, np.int32); p = np.full(n + PAD, -77, np.int32)
    lib.chain0(total_sub, max_dist_x, max_dist_y, bw, q_span, C.c_float(avg), a_pad.ctypes.data, f.ctypes.data, p.ctypes.data,
               ns_pad.ctypes.data)
    assert np.all(f[n:] == -77) and np.all(p[n:] == -77), "the kernel wrote beyond n"
    return ns, f[:n].copy(), p[:n].copy()


cases = []
# (1) the anchor lists the reference's map.o made of its own test FASTA pairs
real = np.load(os.path.join(HERE, "ref_testdata_anchors.npz"))
for k in range(int(real["n_calls"])):
    par = real[f"c{k}_par"]
    cases.append((f"real:{real[f'c{k}_src']}", real[f"c{k}_anchors"], int(par[0]), int(par[1]), int(par[2])))
# (2) synthetic ONT-shaped tasks; `locus` squeezes 4000 anchors into 12 kb so that windows hold > 1024 candidates (all 8 sub-parts)
for prof, n, locus, seed in [("mixed", 3000, None, 11), ("mixed", 2500, None, 12), ("dense", 4000, None, 13), ("dense", 4000, 12000, 14),
                             ("colinear", 3000, None, 15), ("sparse", 1000, None, 16), ("dense", 1500, 3000, 17)]:
    off, a = synth.make_stream(prof, 1, n, seed=seed, locus=locus)
    cases.append((f"synth:{prof}:n={n}:locus={locus}:seed={seed}", a.numpy().view(np.uint64), 5000, 5000, 500))
# (3) other scalar sets: ava-ont (bw 2000, max_dist 10000, options.c:83-86), a tight band, max_dist_y < max_dist_x
off, a = synth.make_stream("mixed", 1, 3000, seed=21)
cases.append(("synth:mixed:ava-ont scalars", a.numpy().view(np.uint64), 10000, 10000, 2000))
cases.append(("synth:mixed:bw=37", a.numpy().view(np.uint64), 5000, 5000, 37))
cases.append(("synth:mixed:max_dist_y=800", a.numpy().view(np.uint64), 5000, 800, 500))
# (4) small coordinates: many equal x (dr == 0), equal scores (ties), dq <= 0, dd == 0 -- the hazards of SURVEY.md App. B
rng = np.random.default_rng(5)
for n, xr, qr in [(600, 300, 300), (900, 2000, 1500), (64, 10, 10), (1, 5, 5), (130, 100000, 100000)]:
    x = np.sort(rng.integers(0, xr, n).astype(np.uint64) + np.uint64(1 << 20))
    y = (np.uint64(15) << np.uint64(32)) | rng.integers(15, 15 + qr, n).astype(np.uint64)
    cases.append((f"random:n={n}:x<{xr}:q<{qr}", np.stack([x, y], 1), 5000, 5000, 500))

# (5) longer tasks (the HIP tile kernel leaves its LDS rings: look-back of 1024 > 448 anchors), k = 19 spans (asm20, options.c:113-122), and a task
#     that crosses reference sequences and strands (the high word of x changes inside the task: the window test is a 64-bit compare)
for prof, n, locus, seed, span in [("dense", 12000, 30000, 31, 15), ("mixed", 15000, None, 32, 15), ("mixed", 6000, None, 33, 19), ("colinear", 5000, None, 34, 19)]:
    off, a = synth.make_stream(prof, 1, n, seed=seed, locus=locus, q_span=span)
    cases.append((f"synth:{prof}:n={n}:locus={locus}:seed={seed}:span={span}", a.numpy().view(np.uint64), 5000, 5000, 500))
parts = []
for rid, strand, n, seed in [(0, 0, 700, 41), (0, 1, 500, 42), (1, 0, 900, 43), (3, 1, 300, 44)]:
    off, a = synth.make_stream("dense", 1, n, seed=seed, locus=4000)
    a = a.numpy().view(np.uint64).copy()
    a[:, 0] = (a[:, 0] & np.uint64(0xffffffff)) | (np.uint64(rid) << np.uint64(33)) | (np.uint64(strand) << np.uint64(32))   # rid << 33 | strand << 32 | pos (map.c:232-241)
    parts.append(a)
multi = np.concatenate(parts)
multi = multi[np.argsort(multi[:, 0], kind="stable")]
cases.append(("synth:dense:4 reference/strand groups in one task", multi, 5000, 5000, 500))

# (6) the shapes bench.py times (BASELINE configs 2, 4, 5): the FIRST reads of its own streams (same generator, same seed 20240, so these
#     are tasks of the benchmarked batches) under the scalars of each preset -- map-ont (options.c:24-31), asm20 (span 19, 7 500 anchors per
#     read, options.c:113-122), ava-ont (bw 2000, max_gap 10000, 20 000 anchors per read in a 400 kb locus, options.c:83-86) -- and the ragged
#     variant of SURVEY 8d
BENCH_SEED = 20240
for prof, n_reads in [("mixed", 2), ("dense", 1), ("colinear", 1), ("sparse", 1)]:
    off, a = synth.make_stream(prof, n_reads, 5000, seed=BENCH_SEED)
    for r in range(n_reads):
        cases.append((f"bench:map-ont:{prof}:read {r}", a.numpy().view(np.uint64)[int(off[r]):int(off[r + 1])], 5000, 5000, 500))
for prof in ("mixed", "dense"):
    off, a = synth.make_stream(prof, 1, 7500, seed=BENCH_SEED, q_span=19)
    cases.append((f"bench:asm20:{prof}:read 0", a.numpy().view(np.uint64), 5000, 5000, 500))
for prof in ("mixed", "colinear"):
    off, a = synth.make_stream(prof, 1, 20000, seed=BENCH_SEED, locus=400000)
    cases.append((f"bench:ava-ont:{prof}:read 0", a.numpy().view(np.uint64), 10000, 10000, 2000))
off, a = synth.make_stream("mixed", 3, (1000, 9000), seed=BENCH_SEED)
for r in range(3):
    cases.append((f"bench:map-ont:mixed ragged:read {r}", a.numpy().view(np.uint64)[int(off[r]):int(off[r + 1])], 5000, 5000, 500))

out = {"n_cases": np.array(len(cases))}
tot = 0
for k, (name, a, mdx, mdy, bw) in enumerate(cases):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 2)
    q_span = int(a[0, 1] >> np.uint64(32) & np.uint64(0xff))           # chain.c:93: the span of a[0] stands for all
    avg = ob.avg_qspan(a)                                               # chain.c:48-49
    ns, f, p = ref_kernel(a, mdx, mdy, bw, q_span, avg)
    out[f"c{k}_name"] = np.array(name); out[f"c{k}_anchors"] = a; out[f"c{k}_num_subparts"] = ns
    out[f"c{k}_scalars"] = np.array([mdx, mdy, bw, q_span], np.int32); out[f"c{k}_avg"] = np.array(avg, np.float32)
    out[f"c{k}_f"] = f; out[f"c{k}_p"] = p
    tot += a.shape[0]
    print(f"case {k}: {name}: n = {a.shape[0]}, sub-parts = {int(ns.sum())}, anchors with a predecessor = {int((p >= 0).sum())}, max f = {int(f.max())}")
np.savez_compressed(os.path.join(HERE, "ref_cl_kernel_fp.npz"), **out)
print(f"{len(cases)} cases, {tot} anchors")
