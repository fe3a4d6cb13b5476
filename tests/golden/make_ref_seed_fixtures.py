"""Seed-hit fixtures: for the reference's test FASTA pairs and for a small synthetic genome with planted
repeats, the matches every read brings to collect_seed_hits (oracle/_ref/seed_dump: the reference's own sketch.o / index.o, with
collect_matches restated) and the anchor list the reference's map.o hands to mm_chain_dp for the same read (MM2O_DUMP of
oracle/_ref/mm2_refhost).  Only runs where /root/reference exists.  Output: tests/golden/ref_seed_hits.npz (data only)."""
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
SEED = os.path.join(ROOT, "oracle", "_ref", "seed_dump")
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle", "ref_host")])

tmp = tempfile.mkdtemp()
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_synth_genome.py"), os.path.join(tmp, "syn"), "--genome-mb", "2",
                       "--reads", "24", "--seed", "11"], stdout=subprocess.DEVNULL)
PAIRS = [(os.path.join(REF, "MT-human.fa"), os.path.join(REF, "MT-orang.fa"), "MT-human vs MT-orang", 10**9),
         (os.path.join(REF, "t-inv.fa"), os.path.join(REF, "q-inv.fa"), "t-inv vs q-inv", 10**9),
         (os.path.join(tmp, "syn.ref.fa"), os.path.join(tmp, "syn.reads.fa"), "synthetic 2 Mb genome with repeats, 10 kb reads", 24)]


def read_anchor_dump(path):
    raw = open(path, "rb").read()
    pos, calls = 0, []
    while pos < len(raw):
        n, = struct.unpack_from("<q", raw, pos); pos += 8 + 36 + 4
        calls.append(np.frombuffer(raw, dtype=np.uint64, count=2 * n, offset=pos).reshape(n, 2).copy()); pos += 16 * n
    return calls


def read_seed_dump(path):
    raw = open(path, "rb").read()
    pos, reads = 0, []
    while pos < len(raw):
        qlen, n_m = struct.unpack_from("<ii", raw, pos); pos += 8
        rec = np.frombuffer(raw, dtype=np.uint32, count=4 * n_m, offset=pos).reshape(n_m, 4).copy(); pos += 16 * n_m
        tot = int(rec[:, 0].sum())
        hits = np.frombuffer(raw, dtype=np.uint64, count=tot, offset=pos).copy(); pos += 8 * tot
        m = np.zeros(n_m, ob.MATCH_DTYPE)
        m["n"], m["q_pos"], m["q_span"], m["seg_tandem"] = rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3]
        m["cr_off"] = np.concatenate([[0], np.cumsum(rec[:, 0].astype(np.int64))[:-1]])
        reads.append((qlen, m, hits))
    return reads


out, k = {}, 0
for ref, qry, what, limit in PAIRS:
    a_dump, s_dump = os.path.join(tmp, "a.bin"), os.path.join(tmp, "s.bin")
    for f in (a_dump, s_dump):
        if os.path.exists(f):
            os.unlink(f)
    subprocess.check_output([HOST, ref, qry], env=dict(os.environ, MM2O_DUMP=a_dump), stderr=subprocess.DEVNULL)
    subprocess.check_call([SEED, ref, qry, s_dump], stderr=subprocess.DEVNULL)
    calls, reads = read_anchor_dump(a_dump), read_seed_dump(s_dump)
    assert len(calls) == len(reads), (what, len(calls), len(reads))
    for (qlen, m, hits), a_ref in list(zip(reads, calls))[:limit]:
        assert int(m["n"].sum()) == a_ref.shape[0], (what, k)
        x = a_ref[:, 0]
        ties = int((x[1:] == x[:-1]).sum())
        out[f"r{k}_qlen"] = np.array(qlen); out[f"r{k}_matches"] = m; out[f"r{k}_hits"] = hits; out[f"r{k}_anchors"] = a_ref
        out[f"r{k}_src"] = np.array(what)
        print(f"read {k}: {what}: qlen {qlen}, {m.size} matches, {a_ref.shape[0]} anchors, {ties} equal-x neighbours")
        k += 1
out["n_reads"] = np.array(k)
np.savez_compressed(os.path.join(HERE, "ref_seed_hits.npz"), **out)
print("wrote", os.path.join(HERE, "ref_seed_hits.npz"), os.path.getsize(os.path.join(HERE, "ref_seed_hits.npz")), "bytes")
