"""Dumps the anchor lists that reach mm_chain_dp when the reference's own host objects (oracle/_ref/mm2_refhost, built by
oracle/ref_host/Makefile from /root/reference) map the reference's test FASTA pairs with -x map-ont, and stores them
with the oracle's f/p and chains.  The PAF lines the same runs print are checked against SURVEY.md section 4 (recorded
from the real reference) in tests/test_cpu_ref_host.py, which is what pins these vectors to the reference.
Only runs where /root/reference exists.  Output: tests/golden/ref_testdata_anchors.npz (data only: anchors, params, f, p)."""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding as ob  # noqa: E402

REF = "/root/reference/test"
HOST = os.path.join(ROOT, "oracle", "_ref", "mm2_refhost")
PAIRS = [("MT-human.fa", "MT-orang.fa"), ("t-inv.fa", "q-inv.fa"), ("t2.fa", "q2.fa")]

subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle", "ref_host")])
out = {}
paf = []
k = 0
for ref, qry in PAIRS:
    with tempfile.NamedTemporaryFile(delete=False) as tf:
        dump = tf.name
    os.unlink(dump)
    env = dict(os.environ, MM2O_DUMP=dump)
    txt = subprocess.check_output([HOST, os.path.join(REF, ref), os.path.join(REF, qry)], env=env, text=True)
    paf.append(f"# {ref} {qry}\n" + txt)
    if not os.path.exists(dump):
        continue
    raw = open(dump, "rb").read()
    os.unlink(dump)
    pos = 0
    while pos < len(raw):
        n, = struct.unpack_from("<q", raw, pos); pos += 8
        h = struct.unpack_from("<9i", raw, pos); pos += 36
        gs, = struct.unpack_from("<f", raw, pos); pos += 4
        a = np.frombuffer(raw, dtype=np.uint64, count=2 * n, offset=pos).reshape(n, 2).copy(); pos += 16 * n
        par = ob.OParams(h[0], h[1], h[2], h[3], h[4], gs, h[7], h[8])
        f, p, v = ob.chain_fpv(par, a)
        u, b = ob.mm_chain_dp(par, h[5], h[6], a)
        out[f"c{k}_anchors"] = a; out[f"c{k}_f"] = f; out[f"c{k}_p"] = p; out[f"c{k}_u"] = u; out[f"c{k}_b"] = b
        out[f"c{k}_par"] = np.array(list(h) + [gs], dtype=np.float64)   # max_dist_x, max_dist_y, bw, max_skip, max_iter, min_cnt, min_sc, is_cdna, n_segs, gap_scale
        out[f"c{k}_src"] = np.array(f"{ref} vs {qry}")
        print(f"call {k}: {ref} vs {qry}: n = {n}, chains = {u.size}, chained anchors = {b.shape[0]}")
        k += 1
out["n_calls"] = np.array(k)
np.savez_compressed(os.path.join(HERE, "ref_testdata_anchors.npz"), **out)
open(os.path.join(HERE, "ref_host_paf_observed.txt"), "w").write("".join(paf))
