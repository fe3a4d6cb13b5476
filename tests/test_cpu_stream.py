"""Anchor-stream files: C and numpy reader/writer agree, and the --print-seeds importer rebuilds the
anchors that the reference host really handed to mm_chain_dp (tests/golden/ref_print_seeds_SD.txt was printed by the
reference's own map.c:298-303 through oracle/_ref/mm2_refhost, MM2_PRINT_SEEDS=1)."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_stream_round_trip_c_and_numpy(tmp_path):
    from mm2chain import stream, params, synth
    off, a = synth.make_stream("mixed", 5, (10, 400), seed=9)
    off = off.numpy(); a = a.numpy().view(np.uint64)
    par = params.make_params(max_skip=7, max_iter=123, gap_scale=0.8, bw=321, n_segs=2, is_cdna=1)
    p1, p2 = tmp_path / "a.mm2a", tmp_path / "b.mm2a"
    stream.write(p1, par, off, a, min_cnt=4, min_sc=55)
    stream.write_c(p2, par, off, a, min_cnt=4, min_sc=55)
    assert open(p1, "rb").read() == open(p2, "rb").read()
    for rd in (stream.read, stream.read_c):
        q, min_cnt, min_sc, off2, a2 = rd(p1)
        assert params.as_dict(q) == params.as_dict(par) and (min_cnt, min_sc) == (4, 55)
        assert np.array_equal(off2, off) and np.array_equal(a2, a)
    # a sub-range of a bigger batch is rebased to offset 0
    stream.write_c(p2, par, off[2:], a)
    _, _, _, off3, a3 = stream.read(p2)
    assert off3[0] == 0 and np.array_equal(a3, a[off[2]:off[-1]])


def test_seed_dump_import_rebuilds_the_real_anchor_lists():
    from mm2chain import stream, params
    par, _, _, off, a = stream.from_seed_dump(os.path.join(GOLD, "ref_print_seeds_SD.txt"), params.map_ont())
    z = np.load(os.path.join(GOLD, "ref_testdata_anchors.npz"))
    assert off.tolist() == [0, 346, 346 + 223, 346 + 223 + 732]
    rid_mask = np.uint64(0x7FFFFFFF00000000)
    for k in range(3):
        got = a[off[k]:off[k + 1]]
        ref = z[f"c{k}_anchors"]
        assert np.array_equal(got[:, 0] & ~rid_mask, ref[:, 0] & ~rid_mask)        # strand + position (reference ids are renumbered)
        assert np.array_equal(got[:, 1] & np.uint64(0xFFFFFFFFFF), ref[:, 1] & np.uint64(0xFFFFFFFFFF))   # span + query position
