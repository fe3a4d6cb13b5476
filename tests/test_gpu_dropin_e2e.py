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


# ---- the caller's side of path A restated (oracle/ref_host/chain_shim_split.c): chain.c's prediction pass, HW/SW decision and busy protocol (chain.c:53-164) over the product's
# mm2c_chain_task_host_pred for the device branch and the oracle's loop for the software branch; both compute the V1 recurrence, so the PAF is the reference's either way
SPLIT_EXE = os.path.join(ROOT, "oracle", "_ref", "mm2_splithost")


@pytest.mark.parametrize("all_hw", [False, True])
def test_host_that_keeps_the_references_hw_sw_split(all_hw):
    if not os.path.exists(SPLIT_EXE):
        pytest.skip("oracle/_ref/mm2_splithost not built (needs /root/reference at build time: __graft_entry__.build())")
    env = dict(os.environ)
    env.pop("MM2_SPLIT_ALL_HW", None)
    if all_hw:
        env["MM2_SPLIT_ALL_HW"] = "1"                                  # C_HW = -1e30: the model sends every read to the device
    outs = {}
    for ref, qry in (("MT-human.fa", "MT-orang.fa"), ("t-inv.fa", "q-inv.fa"), ("t2.fa", "q2.fa")):
        r = subprocess.run([SPLIT_EXE, "-t", "3", os.path.join(DATA, ref), os.path.join(DATA, qry)], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr
        outs[qry] = r.stdout
        if qry == "MT-orang.fa":
            m = [l for l in r.stderr.splitlines() if l.startswith("[mm2_splithost] split model")]
            assert m, r.stderr
            on_device = int(m[0].split(": ")[1].split(" reads on the device")[0])
            assert (on_device > 0) == all_hw, m[0]                     # a 346-anchor read: the MI355X model keeps it on the CPU thread; with C_HW = -1e30 it goes to the device
    assert hashlib.md5(outs["MT-orang.fa"].encode()).hexdigest() == MT_MD5
    want = open(os.path.join(ROOT, "tests", "golden", "ref_host_paf_observed.txt")).read().split("# t-inv.fa q-inv.fa\n")[1].split("#")[0]
    assert outs["q-inv.fa"] == want and outs["q2.fa"] == ""
