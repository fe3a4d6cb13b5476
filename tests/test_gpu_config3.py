"""BASELINE config 3 stand-in under pytest: `-x map-ont` END TO END on 8 000 simulated ONT reads against a 16 Mb synthetic genome
(hg38 is not available offline; tools/make_synth_genome.py regenerates the same bytes on the GPU box, digests checked).

Expected values (tests/golden/config3_expected.json) were recorded in the build container from the reference's own host objects with
CPU chaining (tests/golden/make_config3_fixture.py).  Here, on the MI355X:
  * the same host objects with the PRODUCT's mm_chain_dp (one GPU call per read, INTEGRATION.md path B) and the batched host
    (seed all -> one GPU call, matches in, chains out -> post all; path C) must print that PAF byte for byte (map.c:272-392);
  * the anchor lists that reach mm_chain_dp (10.8 M anchors in 8 000 calls) go through the device-resident plan (f/p against the
    oracle) and through mm2c_mm_chain_dp_batch_host (DP + epilogue on the GPU, chain.c:348-422): chains equal to the recorded digest."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_binding as ob
from helpers import oracle_batch, gpu_batch, assert_same

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_config3_fixture import md5_file, read_dump, chains_digest  # noqa: E402

EXP = json.load(open(os.path.join(ROOT, "tests", "golden", "config3_expected.json")))


@pytest.fixture(scope="module")
def data(tmp_path_factory):
    for exe in ("mm2_refhost", "mm2_gpuhost", "mm2_batchhost"):
        if not os.path.exists(os.path.join(REF_DIR, exe)):
            pytest.skip(f"oracle/_ref/{exe} not built (needs /root/reference at build time: __graft_entry__.build())")
    w = tmp_path_factory.mktemp("config3")
    pre = str(w / "syn")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_synth_genome.py"), pre, "--genome-mb", str(EXP["genome_mb"]),
                           "--reads", str(EXP["reads"]), "--seed", str(EXP["seed"])], stdout=subprocess.DEVNULL)
    assert md5_file(pre + ".ref.fa") == EXP["ref_md5"] and md5_file(pre + ".reads.fa") == EXP["reads_md5"], "generator is not reproducible here"
    return pre, w


def _map(exe, pre, threads, env=None):
    r = subprocess.run([os.path.join(REF_DIR, exe), "-t", str(threads), pre + ".ref.fa", pre + ".reads.fa"], capture_output=True,
                       timeout=900, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return r.stdout, r.stderr.decode()


def test_drop_in_host_prints_the_recorded_paf(data):
    pre, _ = data
    paf, err = _map("mm2_gpuhost", pre, 8)
    assert paf.count(b"\n") == EXP["paf_lines"]
    assert hashlib.md5(paf).hexdigest() == EXP["paf_md5"]
    assert f"GPU chaining: {EXP['n_calls']} tasks, {EXP['total_anchors']} anchors" in err, err[-500:]


def test_batched_host_prints_the_recorded_paf(data):
    pre, _ = data
    paf, err = _map("mm2_batchhost", pre, 8)
    assert hashlib.md5(paf).hexdigest() == EXP["paf_md5"], err[-500:]


def test_anchor_stream_of_the_run_through_the_plan_and_the_batch_api(data):
    import torch
    import mm2chain
    from mm2chain import params
    pre, w = data
    dump = str(w / "dump.bin")
    paf, _ = _map("mm2_refhost", pre, 1, {"MM2O_DUMP": dump})           # CPU chaining (the checker), one thread = read order
    assert hashlib.md5(paf).hexdigest() == EXP["paf_md5"] and md5_file(dump) == EXP["dump_md5"]
    calls = read_dump(dump)
    assert len(calls) == EXP["n_calls"] and all(list(c[0]) == EXP["scalars"] and c[1] == 1.0 for c in calls)
    h = EXP["scalars"]
    P = params.make_params(h[0], h[1], h[2], h[3], h[4], 1.0, h[7], h[8])
    off = np.concatenate([[0], np.cumsum([c[2].shape[0] for c in calls])]).astype(np.int64)
    a = np.concatenate([c[2] for c in calls])
    assert int(off[-1]) == EXP["total_anchors"]
    assert torch.cuda.is_available()
    mm2chain.init()
    try:
        f_ref, p_ref = oracle_batch(P, off, a)
        f, p = gpu_batch(P, off, a)
        assert_same(f, p, f_ref, p_ref, off, "config 3 anchor stream, device-resident plan")
        for threads in (0, 4):                                           # epilogue on the GPU / on host threads
            res = mm2chain.mm_chain_dp_batch(P, h[5], h[6], off, a, epilogue_threads=threads)
            assert sum(r[0].size for r in res) == EXP["n_chains"]
            assert chains_digest(res) == EXP["chains_md5"], f"chains differ (epilogue_threads={threads})"
    finally:
        mm2chain.shutdown()
