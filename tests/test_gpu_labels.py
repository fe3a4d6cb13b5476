"""Every label of the hand-written anchor loop (csrc/chain_dp_tile.h, MM2C_SCAN_TILE_ASM: the instruction sequence the benchmark times) is reached by the
parity inputs, in every one of its eight instantiations (lean / far x computed gap cost / table x 32-bit / compact ring) -- counted on the REAL assembly:
minimap2-fpga_amd/variants/labelcount.so is the same kernel source compiled with -DMM2C_LABEL_COUNT, which adds a counter at each label and on the
fall-through side of each branch that picks a fold (chain.c:226-233: new maximum, skip event, the `break`).  The parity tests run against it in a child
process (so they are checked against the oracle in that build too) and the table of hits comes back; a label that goes cold fails this test."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANT = os.path.join(ROOT, "minimap2-fpga_amd", "variants", "labelcount.so")

# bit -> (label of the assembly or the branch side it stands for, what it means in chain.c's terms)
LABELS = [
    ("Lk", "an anchor enters the loop (chain.c:187)"),
    ("own-tile chunk scored", "a predecessor in the anchor's own tile passes the filters (chain.c:202-205)"),
    ("Lloop", "one more whole older tile of the window"),
    ("Lold", "... with a surviving lane"),
    ("Lhf", "stamps, score and fold of a chunk (chain.c:207-233)"),
    ("far stamp store", "a stamp for a target before the LDS ring goes to the global scratch t[]"),
    ("break in fold A", "no lane beats the best, the skip counter passes max_skip (chain.c:231)"),
    ("Lfg", "f / p of a tile deeper than the f / p ring, from L2"),
    ("Lpart", "the partly covered last tile of the window"),
    ("partly covered tile scored", "... with a surviving lane"),
    ("Limp", "some lane beats the running best (chain.c:226)"),
    ("fold B0", "the first surviving lane is the only new maximum: closed form"),
    ("Lslow2", "the first lane beats the best but another lane beats it"),
    ("Lslow", "general folds"),
    ("fold B1 single", "one candidate, no marks, no skips so far"),
    ("Lb1m", "several candidates, no marks, no skips so far: prefix max only"),
    ("Lb2", "prefix max + skip counter"),
    ("B2 without skip events", "only new maxima: the counter just goes down"),
    ("Lli", "new maxima and skip events in one chunk"),
    ("closed form, no break", "every new maximum precedes every skip event, counter stays within max_skip"),
    ("Lcfb", "... and the counter passes max_skip: the break lane in closed form"),
    ("Lgen", "interleaved new maxima and skip events: max-plus scan of the counter"),
    ("Lbk", "... with the break inside the chunk"),
    ("Ltk", "take the prefix maximum up to the break lane"),
    ("Laf", "after the take: go on or stop"),
    ("Lend", "the ring part of the window is exhausted"),
    ("Lfloop", "whole tiles beyond the LDS ring, from memory"),
    ("Lfpart", "the partly covered tile beyond the ring"),
    ("Lfold", "a chunk from memory with a surviving lane"),
    ("Ldone", "commit f[i], p[i] (chain.c:236)"),
    ("Lspec", "anchors without a window, or handed to the C++ path (equal-x run reaching into the tile before)"),
]
FAR_ONLY = {5, 26, 27, 28}
ROWS = [f"{'compact' if r & 4 else '32-bit'} ring, {'table' if r & 2 else 'computed'} gap cost, {'far' if r & 1 else 'lean'}" for r in range(8)]
MIN_HITS = 5
# the parity cases whose inputs make up the union (the reference-kernel vectors through every route, the per-path unit cases, the compact-ring corners,
# windows around the ring boundaries, the BASELINE shapes, random scalars on adversarial tasks, and the cases written to drive the rare folds)
SELECT = ("hand_written_loop or tile_kernel_paths or compact_ring or window_lengths or far_lookback or ring_size_classes or configs_4_and_5 or randomised "
          "or parameter_corners or drive_every_fold or references_own_device_kernel or profiles_map_ont")


def test_every_label_of_the_hand_written_loop_is_reached_by_the_parity_inputs(tmp_path):
    assert os.path.exists(VARIANT), "minimap2-fpga_amd/variants/labelcount.so is built by `make -C minimap2-fpga_amd` (__graft_entry__.build())"
    table = str(tmp_path / "labels.json")
    env = dict(os.environ, MM2C_LIB_PATH=VARIANT, MM2C_LABEL_TABLE=table)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "-k", SELECT],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, "the parity tests fail in the label-counting build:\n" + r.stdout[-3000:] + r.stderr[-2000:]
    rec = json.load(open(table))
    assert rec["lib"] == VARIANT and rec["exitstatus"] == 0
    hits = rec["hits"]
    lines = ["# Label hits of the hand-written anchor loop under the parity inputs (`tests/test_gpu_labels.py`, counted by `variants/labelcount.so` on the MI355X)", "",
             "Rows: labels of `MM2C_SCAN_TILE_ASM` (and the fall-through side of the branches that choose a fold); columns: its eight instantiations.  `-`: the label does not exist in "
             "the lean instantiations.  The test fails when an entry falls below %d." % MIN_HITS, "",
             "| label | meaning | " + " | ".join(ROWS) + " |", "|---|---|" + "---|" * 8]
    cold = []
    for b, (name, what) in enumerate(LABELS):
        cells = []
        for row in range(8):
            if b in FAR_ONLY and not (row & 1):
                cells.append("-")
                continue
            h = hits[row * 32 + b]
            cells.append(str(h))
            if h < MIN_HITS:
                cold.append(f"{name} in [{ROWS[row]}]: {h} hits")
        lines.append(f"| `{name}` | {what} | " + " | ".join(cells) + " |")
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        open(os.path.join(out_dir, "label_hits.md"), "w").write("\n".join(lines) + "\n")
    assert not cold, "labels of the hand-written loop that the parity inputs do not reach:\n" + "\n".join(cold)
