"""Heap-variant seed-hit fixture (collect_seed_hits_heap, map.c:149-213; MM_F_HEAP_SORT = --heap-sort, main.c:245): for the reads of
tests/golden/ref_seed_hits.npz (same files, same order: the generator checks that the matches are the same) the anchor list the reference's map.o
hands to mm_chain_dp when that flag is set (MM2O_DUMP of oracle/_ref/mm2_refhost with MM2_HEAP_SORT=1).  Same anchors as the radix-sorted
lists, another order among equal x.  Only runs where /root/reference exists.  Output: tests/golden/ref_seed_hits_heap.npz (data only)."""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
import oracle_binding as ob  # noqa: E402

REF = "/root/reference/test"
HOST = os.path.join(ROOT, "oracle", "_ref", "mm2_refhost")
SEED = os.path.join(ROOT, "oracle", "_ref", "seed_dump")
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle", "ref_host")])
from make_ref_seed_fixtures import read_anchor_dump, read_seed_dump  # noqa: E402  (importing it regenerates ref_seed_hits.npz: same content)

base = np.load(os.path.join(HERE, "ref_seed_hits.npz"))
tmp = tempfile.mkdtemp()
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_synth_genome.py"), os.path.join(tmp, "syn"), "--genome-mb", "2",
                       "--reads", "24", "--seed", "11"], stdout=subprocess.DEVNULL)
PAIRS = [(os.path.join(REF, "MT-human.fa"), os.path.join(REF, "MT-orang.fa"), 10**9), (os.path.join(REF, "t-inv.fa"), os.path.join(REF, "q-inv.fa"), 10**9),
         (os.path.join(tmp, "syn.ref.fa"), os.path.join(tmp, "syn.reads.fa"), 24)]
out, k, n_diff = {}, 0, 0
for ref, qry, limit in PAIRS:
    a_dump, s_dump = os.path.join(tmp, "a.bin"), os.path.join(tmp, "s.bin")
    for f in (a_dump, s_dump):
        if os.path.exists(f):
            os.unlink(f)
    subprocess.check_output([HOST, ref, qry], env=dict(os.environ, MM2O_DUMP=a_dump, MM2_HEAP_SORT="1"), stderr=subprocess.DEVNULL)
    subprocess.check_call([SEED, ref, qry, s_dump], stderr=subprocess.DEVNULL)
    calls, reads = read_anchor_dump(a_dump), read_seed_dump(s_dump)
    assert len(calls) == len(reads)
    for (qlen, m, hits), a_heap in list(zip(reads, calls))[:limit]:
        assert np.array_equal(m, base[f"r{k}_matches"]) and np.array_equal(hits, base[f"r{k}_hits"]) and qlen == int(base[f"r{k}_qlen"]), k
        a_radix = base[f"r{k}_anchors"]
        assert a_heap.shape == a_radix.shape and np.array_equal(np.sort(a_heap.view([("x", "<u8"), ("y", "<u8")]).ravel(), order=("x", "y")),
                                                                 np.sort(a_radix.view([("x", "<u8"), ("y", "<u8")]).ravel(), order=("x", "y"))), k
        d = int((a_heap != a_radix).any(axis=1).sum())
        n_diff += d
        print(f"read {k}: {a_heap.shape[0]} anchors, {d} at another place than in the radix-sorted list")
        out[f"r{k}_anchors_heap"] = a_heap
        k += 1
assert k == int(base["n_reads"]) and n_diff > 0
out["n_reads"] = np.array(k)
np.savez_compressed(os.path.join(HERE, "ref_seed_hits_heap.npz"), **out)
print("wrote", os.path.join(HERE, "ref_seed_hits_heap.npz"), os.path.getsize(os.path.join(HERE, "ref_seed_hits_heap.npz")), "bytes")
