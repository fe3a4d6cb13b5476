"""End-to-end drop-in: the reference's own host objects (index, sketch, seeding, hit filtering, PAF writer; built in place
from /root/reference into oracle/_ref/ by oracle/ref_host/Makefile, the objects travel to the GPU box) linked with the
PRODUCT library's mm_chain_dp (DP on the GPU, INTEGRATION.md path B) must print the PAF the real reference prints for
its own test data.  FASTA inputs are the reference's test files kept as fixtures."""
import hashlib
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "mm2_gpuhost")
DATA = os.path.join(ROOT, "tests", "golden", "ref_testdata")
MT_MD5 = "f49a6331f92e6f24acc73485827a2eba"


def _run(ref, qry):
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/mm2_gpuhost not built (needs /root/reference at build time: __graft_entry__.build())")
    r = subprocess.run([EXE, os.path.join(DATA, ref), os.path.join(DATA, qry)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "GPU chaining:" in r.stderr and " 0 tasks" not in r.stderr or qry == "q2.fa", r.stderr
    return r.stdout


def test_mt_human_vs_orang_with_gpu_chaining():
    out = _run("MT-human.fa", "MT-orang.fa")
    assert hashlib.md5(out.encode()).hexdigest() == MT_MD5, out
    assert "cm:i:342\ts1:i:3189" in out


def test_inversion_pair_with_gpu_chaining():
    lines = _run("t-inv.fa", "q-inv.fa").splitlines()
    assert len(lines) == 2
    assert lines[0].startswith("read1\t") and "cm:i:211\ts1:i:1816" in lines[0]
    assert lines[1].startswith("read2\t") and "cm:i:700\ts1:i:4415" in lines[1]
    want = open(os.path.join(ROOT, "tests", "golden", "ref_host_paf_observed.txt")).read().split("# t-inv.fa q-inv.fa\n")[1].split("#")[0]
    assert "\n".join(lines) + "\n" == want        # byte-identical to the run with the oracle's mm_chain_dp


def test_short_pair_with_gpu_chaining():
    assert _run("t2.fa", "q2.fa") == ""


# ---- the batched host: worker_for restructured as seed all -> ONE GPU call (matches in, chains out) -> post all;
# examples/batch_host/batch_driver.c over the reference's own objects and mm2c_seed_chain_batch_host
BATCH_EXE = os.path.join(ROOT, "oracle", "_ref", "mm2_batchhost")


def _run_batch(ref, qry, threads=2):
    if not os.path.exists(BATCH_EXE):
        pytest.skip("oracle/_ref/mm2_batchhost not built (needs /root/reference at build time: __graft_entry__.build())")
    r = subprocess.run([BATCH_EXE, "-t", str(threads), os.path.join(DATA, ref), os.path.join(DATA, qry)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return r.stdout


def test_batched_host_prints_the_reference_paf():
    out = _run_batch("MT-human.fa", "MT-orang.fa")
    assert hashlib.md5(out.encode()).hexdigest() == MT_MD5, out
    lines = _run_batch("t-inv.fa", "q-inv.fa")
    want = open(os.path.join(ROOT, "tests", "golden", "ref_host_paf_observed.txt")).read().split("# t-inv.fa q-inv.fa\n")[1].split("#")[0]
    assert lines == want
    assert _run_batch("t2.fa", "q2.fa") == ""
