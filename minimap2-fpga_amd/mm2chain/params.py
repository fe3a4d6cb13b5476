"""Chaining parameter presets, mirroring options.c of the reference (file:line cited per preset)."""
from ._native import Params, MM2C_F_IGNORE_SEG, MM2C_F_FORCE_GENERAL

INT32_MAX = 2**31 - 1


def make_params(max_dist_x=5000, max_dist_y=5000, bw=500, max_skip=25, max_iter=5000, gap_scale=1.0,
                is_cdna=0, n_segs=1, q_span_override=-1, flags=0):
    return Params(max_dist_x, max_dist_y, bw, max_skip, max_iter, gap_scale, is_cdna, n_segs, q_span_override, flags)


def map_ont():
    """-x map-ont: options.c:24-31 defaults kept by :93-99; max_gap feeds both max_dist (map.c:305-316)."""
    return make_params()


def asm20():
    """-x asm20 (options.c:113-122): same chaining scalars as the defaults; k=19 changes only the anchor span."""
    return make_params()


def ava_ont():
    """-x ava-ont (options.c:83-86): bw=2000, max_gap=10000."""
    return make_params(max_dist_x=10000, max_dist_y=10000, bw=2000)


def fpga_v2(max_dist_x=5000, max_dist_y=5000, bw=500, q_span=15):
    """What the reference's FPGA kernel computes for one run_chaining_on_hw call (device/minimap2_opencl.cl)."""
    return make_params(max_dist_x, max_dist_y, bw, INT32_MAX, 1024, 1.0, 0, 1, q_span, MM2C_F_IGNORE_SEG)


def as_dict(p):
    return {k: getattr(p, k) for k, _ in p._fields_}
