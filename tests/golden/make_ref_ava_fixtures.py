"""All-vs-all seed-hit fixtures (BASELINE config 5, `-x ava-ont`): reads of a small synthetic genome mapped against THEMSELVES by the
reference's own host objects with the ava-ont options (options.c:82-86: NO_DIAG | NO_DUAL, k = 15, w = 5) -- skip_seed (map.c:122-147) drops
the diagonal, maps every pair once and sets MM_SEED_SELF.  Per read: the matches (reference's sketch.o / index.o via oracle/_ref/seed_dump
-x ava-ont, with the rank form of the name comparison) and the anchor list the reference's map.o handed to mm_chain_dp (MM2O_DUMP of
oracle/_ref/mm2_refhost -x ava-ont).  Only runs where /root/reference exists.  Output: tests/golden/ref_seed_hits_ava.npz (data only)."""
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

HOST = os.path.join(ROOT, "oracle", "_ref", "mm2_refhost")
SEED = os.path.join(ROOT, "oracle", "_ref", "seed_dump")
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle", "ref_host")])
tmp = tempfile.mkdtemp()
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_synth_genome.py"), os.path.join(tmp, "syn"), "--genome-mb", "0.06",
                       "--reads", "36", "--read-len", "4000", "--seed", "5"], stdout=subprocess.DEVNULL)
reads = os.path.join(tmp, "syn.reads.fa")
# two reads with the same name and length as an earlier one would be "self" for each other; one renamed copy exercises cmp == 0 with another length
txt = open(reads).read().split(">")[1:]
txt.append(txt[3].split("\n")[0] + "\n" + txt[3].split("\n")[1][:2500] + "\n")            # same name, shorter: cmp == 0 but length differs
open(reads, "w").write("".join(">" + t for t in txt))
a_dump, s_dump = os.path.join(tmp, "a.bin"), os.path.join(tmp, "s.bin")
subprocess.check_output([HOST, "-x", "ava-ont", reads, reads], env=dict(os.environ, MM2O_DUMP=a_dump, MM2O_DUMP_ALL="1"), stderr=subprocess.DEVNULL)
subprocess.check_call([SEED, "-x", "ava-ont", reads, reads, s_dump], stderr=subprocess.DEVNULL)

raw = open(a_dump, "rb").read(); pos = 0; calls = []; pars = None
while pos < len(raw):
    n, = struct.unpack_from("<q", raw, pos); pos += 8
    h = struct.unpack_from("<9i", raw, pos); pos += 36 + 4
    pars = h
    calls.append(np.frombuffer(raw, dtype=np.uint64, count=2 * n, offset=pos).reshape(n, 2).copy()); pos += 16 * n
raw = open(s_dump, "rb").read(); pos = 0
n_ref, = struct.unpack_from("<i", raw, pos); pos += 4
rr = np.frombuffer(raw, dtype=np.int32, count=2 * n_ref, offset=pos).reshape(n_ref, 2).copy(); pos += 8 * n_ref
out = {"ref_rank": rr[:, 0].copy(), "ref_len": rr[:, 1].copy(), "flag": np.array(ob.F_NO_DIAG | ob.F_NO_DUAL), "chain_scalars": np.array(pars, np.int32)}
k = 0
while pos < len(raw):
    qlen, n_m, q_lo, q_eq = struct.unpack_from("<iiii", raw, pos); pos += 16
    rec = np.frombuffer(raw, dtype=np.uint32, count=4 * n_m, offset=pos).reshape(n_m, 4).copy(); pos += 16 * n_m
    tot = int(rec[:, 0].sum())
    hits = np.frombuffer(raw, dtype=np.uint64, count=tot, offset=pos).copy(); pos += 8 * tot
    m = np.zeros(n_m, ob.MATCH_DTYPE)
    m["n"], m["q_pos"], m["q_span"], m["seg_tandem"] = rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3]
    m["cr_off"] = np.concatenate([[0], np.cumsum(rec[:, 0].astype(np.int64))[:-1]]) if n_m else np.zeros(0, np.int64)
    a_ref = calls[k]
    n_self = int(((a_ref[:, 1] >> np.uint64(43)) & np.uint64(1)).sum())
    out[f"r{k}_qlen"] = np.array(qlen); out[f"r{k}_qlo"] = np.array(q_lo); out[f"r{k}_qeq"] = np.array(q_eq)
    out[f"r{k}_matches"] = m; out[f"r{k}_hits"] = hits; out[f"r{k}_anchors"] = a_ref
    print(f"read {k}: qlen {qlen}, q_lo {q_lo}, q_eq {q_eq}, {n_m} matches, {tot} hits -> {a_ref.shape[0]} anchors kept ({n_self} with MM_SEED_SELF)")
    k += 1
assert k == len(calls), (k, len(calls))
out["n_reads"] = np.array(k)
np.savez_compressed(os.path.join(HERE, "ref_seed_hits_ava.npz"), **out)
print("wrote", os.path.getsize(os.path.join(HERE, "ref_seed_hits_ava.npz")), "bytes")
