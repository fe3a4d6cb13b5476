"""Regenerates tests/golden/*.npz.

PROVENANCE: these vectors are produced by the repo's own CPU oracle (oracle/chain_oracle.c), NOT by the reference
binary: the reference's chain.c cannot be compiled in this image (it needs the Xilinx XRT header
CL/cl_ext_xilinx.h through chain_hardware.h -> xcl2.hpp:34) and its tree holds no f[]/p[] vectors.  They are
regression fixtures: they freeze the oracle's behaviour at the commit where it was validated against the hand-derived
known answers, the independent Python restatement and the literal FPGA-kernel emulation (tests/test_cpu_oracle.py).
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "minimap2-fpga_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_binding as ob  # noqa: E402
from mm2chain import synth, params  # noqa: E402

INT32_MAX = 2**31 - 1
CASES = {
    "map_ont_mixed": (params.map_ont(), "mixed", 6, (200, 1200), 1, {}),
    "map_ont_dense": (params.map_ont(), "dense", 4, 1500, 2, {}),
    "ava_ont_colinear": (params.ava_ont(), "colinear", 4, (300, 900), 3, {}),
    "asm20_span19": (params.asm20(), "mixed", 4, 800, 4, {"q_span": 19}),
    "v2_noskip_1024": (params.make_params(max_skip=INT32_MAX, max_iter=1024), "dense", 3, 1800, 5, {"locus": 9000}),
    "corner_skip3_iter100_gs08": (params.make_params(max_skip=3, max_iter=100, gap_scale=0.8), "dense", 4, 900, 6, {}),
}

for name, (par, profile, n_reads, n_per, seed, kw) in CASES.items():
    off, a = synth.make_stream(profile, n_reads, n_per, seed=seed, **kw)
    off = off.numpy(); a = a.numpy().view(np.uint64)
    f, p, _ = ob.chain_batch(par, off, a, 1)
    d = {"offsets": off, "anchors": a, "f": f, "p": p}
    for k, _ in par._fields_:
        if k in ("q_span_override", "flags"):
            continue
        d["par_" + k] = np.array(getattr(par, k))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(name, a.shape[0], "anchors")
