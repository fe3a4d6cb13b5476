"""BASELINE config 3 stand-in (hg38 is not available offline): map-ont end to end on a synthetic genome.

Runs, in the build container, the reference's own host objects with CPU chaining (oracle/_ref/mm2_refhost; built in place from
/root/reference by oracle/ref_host/Makefile) on the genome and reads that tools/make_synth_genome.py generates deterministically, and
records what the GPU test (tests/test_gpu_config3.py) must reproduce on the MI355X box: md5 of the inputs, of the PAF, of the anchor
lists that reached mm_chain_dp (MM2O_DUMP, one thread = read order), and of the chains the oracle made of them.
Output: tests/golden/config3_expected.json (numbers and digests only)."""
import hashlib
import json
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

ARGS = {"genome_mb": 16, "reads": 8000, "seed": 7}


def md5_file(path):
    h = hashlib.md5()
    with open(path, "rb") as fp:
        for blk in iter(lambda: fp.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()


def read_dump(path):
    """MM2O_DUMP records (oracle/ref_host/chain_shim.c) -> list of (header ints[9], gap_scale, anchors uint64 [n,2])"""
    raw = np.fromfile(path, dtype=np.uint8)
    calls, pos = [], 0
    while pos < raw.size:
        n, = struct.unpack_from("<q", raw, pos); pos += 8
        h = struct.unpack_from("<9i", raw, pos); pos += 36
        gs, = struct.unpack_from("<f", raw, pos); pos += 4
        a = raw[pos:pos + 16 * n].view(np.uint64).reshape(n, 2); pos += 16 * n
        calls.append((h, gs, a))
    return calls


def chains_digest(results):
    """md5 over u then b of every call, in call order"""
    h = hashlib.md5()
    for u, b in results:
        h.update(np.ascontiguousarray(u, dtype=np.uint64).tobytes()); h.update(np.ascontiguousarray(b, dtype=np.uint64).tobytes())
    return h.hexdigest()


if __name__ == "__main__":
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle", "ref_host")])
    with tempfile.TemporaryDirectory() as w:
        pre = os.path.join(w, "syn")
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_synth_genome.py"), pre, "--genome-mb", str(ARGS["genome_mb"]),
                               "--reads", str(ARGS["reads"]), "--seed", str(ARGS["seed"])])
        dump = os.path.join(w, "dump.bin")
        paf = subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", "mm2_refhost"), "-t", "1", pre + ".ref.fa", pre + ".reads.fa"],
                                      env=dict(os.environ, MM2O_DUMP=dump))
        calls = read_dump(dump)
        res = []
        for h, gs, a in calls:
            res.append(ob.mm_chain_dp(ob.OParams(h[0], h[1], h[2], h[3], h[4], gs, h[7], h[8]), h[5], h[6], a))
        exp = dict(ARGS, ref_md5=md5_file(pre + ".ref.fa"), reads_md5=md5_file(pre + ".reads.fa"), paf_md5=hashlib.md5(paf).hexdigest(),
                   paf_lines=paf.count(b"\n"), n_calls=len(calls), total_anchors=int(sum(c[2].shape[0] for c in calls)),
                   max_anchors=int(max(c[2].shape[0] for c in calls)), dump_md5=md5_file(dump), chains_md5=chains_digest(res),
                   n_chains=int(sum(r[0].size for r in res)), scalars=list(calls[0][0]))
    json.dump(exp, open(os.path.join(HERE, "config3_expected.json"), "w"), indent=1)
    print(json.dumps(exp, indent=1))
