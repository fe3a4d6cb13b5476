#!/usr/bin/env python3
"""How often an anchor of a synthetic stream takes each path of the hand-written loop of chain_dp_tile (own-tile chunk, older tiles from the
ring, deep f / p, beyond the ring; fold A / B1 / B2; the `break` of chain.c:231), counted by the NumPy model of the kernel's control flow
(tests/tile_model.py, checked against the CPU oracle in tests/test_cpu_oracle.py).  Together with tools/isa_budget.py this decomposes the
per-anchor instruction counts of the PMC profiles.   python tools/chunk_stats.py [profile ...] [--reads N] [--anchors M]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimap2-fpga_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mm2chain import params, synth  # noqa: E402
from tile_model import chain_tile_model  # noqa: E402


def avg_qspan(t):
    return float(np.float32(.01 * float(np.float32(int(((t[:, 1] >> np.uint64(32)) & np.uint64(0xff)).sum()))) / t.shape[0]))


if __name__ == "__main__":
    args = sys.argv[1:]
    reads = int(args[args.index("--reads") + 1]) if "--reads" in args else 6
    m = int(args[args.index("--anchors") + 1]) if "--anchors" in args else 5000
    profiles = [a for a in args if not a.startswith("--") and not a.isdigit()] or ["mixed", "dense", "colinear"]
    P = params.map_ont()
    NX = 16 if "--nx16" in args else 8
    span = 19 if "--asm20" in args else 15
    keys = ("no_window", "own_chunks", "own_pass", "ring_chunks", "ring_pass", "deep_fp", "far_chunks", "far_pass", "fold_a", "fold_b0", "fold_b1",
            "fold_b2_closed", "fold_b2_scan", "breaks", "eq_run_anchors")
    if "--tile-skip" in args:
        # round 6: what a per-tile bitmap of diagonal buckets would reject (skip_rejects of skip_tested ring-tile visits; skip_missed: visits without a surviving lane that
        # the bitmap lets through; skip_wrong must be 0)
        keys = ("ring_chunks", "ring_pass", "skip_tested", "skip_rejects", "skip_missed", "skip_wrong")
    print(f"Per anchor, map-ont parameters, {reads} reads x {m} anchors of each bench.py stream (seed 1, span {span}), NX {NX} / NF 2:\n")
    print("| stream | " + " | ".join(keys) + " |")
    print("|---|" + "---|" * len(keys))
    for prof in profiles:
        off, a = synth.make_stream(prof, reads, m, seed=1, q_span=span)
        off = off.numpy(); a = a.numpy().view(np.uint64)
        tot = dict.fromkeys(("anchors",) + keys, 0)
        for k in range(reads):
            t = a[off[k]:off[k + 1]]
            st = {}
            chain_tile_model(P, t, avg_qspan(t), stats=st, NX=NX)
            for kk in tot:
                tot[kk] += st[kk]
        print(f"| {prof} | " + " | ".join(f"{tot[k] / tot['anchors']:.3f}" for k in keys) + " |")
