#!/usr/bin/env python3
"""Synthetic genome + simulated ONT reads (BASELINE config 3 stand-in; hg38 is not available offline).
  make_synth_genome.py <out_prefix> [--genome-mb 20] [--reads 2000] [--read-len 10000] [--err 0.10] [--seed 7]
writes <out_prefix>.ref.fa and <out_prefix>.reads.fa.  Random sequence with planted repeat families (so that minimizers hit
many loci, as in a real genome) cut into 4 'chromosomes'; reads from random positions and strands with substitutions,
insertions and deletions.  Deterministic (numpy PCG64 with the given seed)."""
import argparse

import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("prefix")
ap.add_argument("--genome-mb", type=float, default=20.0)
ap.add_argument("--reads", type=int, default=2000)
ap.add_argument("--read-len", type=int, default=10000)
ap.add_argument("--err", type=float, default=0.10)
ap.add_argument("--repeat-frac", type=float, default=0.25)
ap.add_argument("--seed", type=int, default=7)
args = ap.parse_args()
rng = np.random.default_rng(args.seed)
G = int(args.genome_mb * 1e6)
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
g = rng.integers(0, 4, G, dtype=np.uint8)
# planted repeats: families of 300..6000 bp, copies diverged by 2..15 %
filled = 0
while filled < args.repeat_frac * G:
    L = int(rng.integers(300, 6000))
    fam = rng.integers(0, 4, L, dtype=np.uint8)
    copies = int(rng.integers(20, 400))
    div = float(rng.uniform(0.02, 0.15))
    for _ in range(copies):
        c = fam.copy()
        m = rng.random(L) < div
        c[m] = (c[m] + rng.integers(1, 4, int(m.sum()), dtype=np.uint8)) % 4
        pos = int(rng.integers(0, G - L))
        g[pos:pos + L] = c
        filled += L
comp = np.array([3, 2, 1, 0], dtype=np.uint8)
with open(args.prefix + ".ref.fa", "w") as fp:
    n_chr = 4
    for c in range(n_chr):
        s = ACGT[g[c * G // n_chr:(c + 1) * G // n_chr]].tobytes().decode()
        fp.write(f">chr{c + 1}\n")
        for k in range(0, len(s), 80):
            fp.write(s[k:k + 80] + "\n")
with open(args.prefix + ".reads.fa", "w") as fp:
    e = args.err
    for r in range(args.reads):
        L = int(rng.normal(args.read_len, args.read_len * 0.2))
        L = max(1000, min(L, 3 * args.read_len))
        pos = int(rng.integers(0, G - L))
        seq = g[pos:pos + L].copy()
        if rng.random() < 0.5:
            seq = comp[seq[::-1]]
        u = rng.random(L)
        sub = u < e * 0.4
        seq[sub] = (seq[sub] + rng.integers(1, 4, int(sub.sum()), dtype=np.uint8)) % 4
        keep = ~((u >= e * 0.4) & (u < e * 0.7))                       # deletions
        ins = (u >= e * 0.7) & (u < e)                                  # insertions after the base
        out = np.repeat(seq, 1 + ins.astype(np.int64) * keep)            # duplicated slot = inserted base
        out = out[np.repeat(keep, 1 + ins.astype(np.int64) * keep)] if False else out
        # simpler exact construction: walk once
        pieces = seq[keep]
        insert_at = np.nonzero(ins[keep])[0]
        pieces = np.insert(pieces, insert_at + 1, rng.integers(0, 4, insert_at.size, dtype=np.uint8))
        fp.write(f">read{r}_pos{pos}_len{L}\n{ACGT[pieces].tobytes().decode()}\n")
print(f"wrote {args.prefix}.ref.fa ({G} bp) and {args.prefix}.reads.fa ({args.reads} reads)")
