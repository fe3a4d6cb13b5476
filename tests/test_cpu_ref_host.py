"""Pins the oracle to the reference on the reference's own test data, end to end.

oracle/ref_host builds the reference's NON-PATH host sources in place (index, sketch, seeding, hit filtering, PAF writer)
and links them with an mm_chain_dp computed by the repo's oracle.  The PAF it prints for the reference's test FASTA
pairs must equal what the real reference printed (recorded in SURVEY.md section 4 from a run of the unmodified
reference host): same chains, same chain scores, same mapping coordinates.  Needs /root/reference (absent on the GPU
box -> skipped there); the anchor lists it dumps are committed (tests/golden/ref_testdata_anchors.npz) and are what the
GPU tests use."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

import oracle_binding as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
HOST = os.path.join(ROOT, "oracle", "_ref", "mm2_refhost")
GOLD = os.path.join(ROOT, "tests", "golden", "ref_testdata_anchors.npz")

# SURVEY.md section 4 ("[probe] Golden values obtained by building the unmodified reference host")
MT_LINE = ("MT_orang\t16499\t61\t16018\t+\tMT_human\t16569\t637\t16562\t3196\t15967\t60\ttp:A:P\tcm:i:342\ts1:i:3189\ts2:i:0\t"
           "dv:f:0.1349\trl:i:0\n")
MT_MD5 = "f49a6331f92e6f24acc73485827a2eba"

needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")


@pytest.fixture(scope="module")
def host():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle", "ref_host")])
    return HOST


def _run(host, ref, qry):
    return subprocess.check_output([host, f"{REF}/test/{ref}", f"{REF}/test/{qry}"], text=True)


@needs_ref
def test_mt_human_vs_orang_paf_is_the_reference_line(host):
    out = _run(host, "MT-human.fa", "MT-orang.fa")
    assert out == MT_LINE
    assert hashlib.md5(out.encode()).hexdigest() == MT_MD5


@needs_ref
def test_inversion_pair_chain_counts_and_scores(host):
    lines = _run(host, "t-inv.fa", "q-inv.fa").splitlines()
    assert len(lines) == 2
    r1, r2 = (ln.split("\t") for ln in lines)
    assert r1[0] == "read1" and "cm:i:211" in r1 and "s1:i:1816" in r1
    assert r2[0] == "read2" and "cm:i:700" in r2 and "s1:i:4415" in r2


@needs_ref
def test_too_short_pair_prints_nothing(host):
    assert _run(host, "t2.fa", "q2.fa") == ""


def test_committed_anchor_fixture_matches_the_oracle():
    """the dumped real anchor lists (n = 346 / 223 / 732) with the f, p, chains stored beside them"""
    z = np.load(GOLD)
    assert int(z["n_calls"]) == 3
    assert z["c0_anchors"].shape[0] == 346 and z["c0_b"].shape[0] == 342      # 346 SD lines -> 342 CN lines
    for k in range(3):
        h = z[f"c{k}_par"]
        par = ob.OParams(int(h[0]), int(h[1]), int(h[2]), int(h[3]), int(h[4]), float(h[9]), int(h[7]), int(h[8]))
        f, p, _ = ob.chain_fpv(par, z[f"c{k}_anchors"])
        assert np.array_equal(f, z[f"c{k}_f"]) and np.array_equal(p, z[f"c{k}_p"])
        u, b = ob.mm_chain_dp(par, int(h[5]), int(h[6]), z[f"c{k}_anchors"])
        assert np.array_equal(u, z[f"c{k}_u"]) and np.array_equal(b, z[f"c{k}_b"])
    # the best chain of the MT pair: score 3189 (PAF s1), 342 anchors (PAF cm)
    assert int(z["c0_u"][0] >> np.uint64(32)) == 3189 and int(z["c0_u"][0] & np.uint64(0xFFFFFFFF)) == 342
