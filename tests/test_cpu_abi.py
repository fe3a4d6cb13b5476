"""CPU tests of the drop-in boundary: libmm2chain_hip.so loads, exports every symbol include/mm2chain.h declares plus
the reference's three C++ symbols, and fails loudly (no CPU fallback) when there is no GPU."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_c_functions():
    src = open(os.path.join(ROOT, "include", "mm2chain.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(mm2c_[a-z0-9_]+|mm_chain_dp)\s*\(", src))
    return names - {"mm2c_anchor_t", "mm2c_params_t", "mm2c_plan_t", "mm2c_stats_t"}


def test_library_exports_every_declared_symbol():
    from mm2chain import _native as N
    lib = N.load()
    declared = _declared_c_functions()
    assert declared, "header parse failed"
    assert declared == set(N.C_SYMBOLS), f"binding table and header differ: {declared ^ set(N.C_SYMBOLS)}"
    for name in list(declared) + list(N.CXX_SYMBOLS):
        assert getattr(lib, name) is not None
    out = subprocess.check_output(["nm", "-D", "--defined-only", N.LIB_PATH], text=True)
    for name in list(declared) + list(N.CXX_SYMBOLS):
        assert re.search(rf"\bT {re.escape(name)}\b", out), f"{name} not exported"


def test_presets_match_reference_options():
    """options.c:24-31 (defaults), :83-86 (ava-ont); chain_hardware.h:58-60 (V2 look-back 128*8)"""
    import ctypes as C
    from mm2chain import _native as N, params
    lib = N.load()
    p = N.Params()
    lib.mm2c_params_map_ont(C.byref(p))
    assert (p.max_dist_x, p.max_dist_y, p.bw, p.max_skip, p.max_iter, p.gap_scale, p.is_cdna, p.n_segs) == (5000, 5000, 500, 25, 5000, 1.0, 0, 1)
    assert params.as_dict(params.map_ont()) == params.as_dict(p)
    lib.mm2c_params_fpga_v2(C.byref(p), 5000, 4000, 500, 19)
    assert (p.max_skip, p.max_iter, p.q_span_override, p.flags & N.MM2C_F_IGNORE_SEG, p.max_dist_y) == (2**31 - 1, 1024, 19, 1, 4000)
    a = params.ava_ont()
    assert (a.max_dist_x, a.bw) == (10000, 2000)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_gpu_means_loud_failure_not_cpu_fallback():
    import mm2chain
    from mm2chain import params
    with pytest.raises(mm2chain.Mm2cError):
        mm2chain.init()
    a = np.zeros((4, 2), np.uint64)
    with pytest.raises(mm2chain.Mm2cError):
        mm2chain.chain_task(params.map_ont(), a, 0.15)
    with pytest.raises(mm2chain.Mm2cError):
        mm2chain.chain_batch_host(params.map_ont(), [0, 4], a)
    with pytest.raises(mm2chain.Mm2cError):
        mm2chain.ChainPlan(params.map_ont(), [0, 4])
    assert mm2chain.hardware_init() is False          # main.c:367-369 then returns -1
    # the whole-function and seed-hit entries: no device, no result (there is no host path behind them)
    with pytest.raises(mm2chain.Mm2cError):
        mm2chain.mm_chain_dp_batch(params.map_ont(), 3, 40, [0, 4], a, epilogue_threads=0)
    with pytest.raises(mm2chain.Mm2cError):
        mm2chain.mm_chain_dp_batch(params.map_ont(), 3, 40, [0, 4], a, epilogue_threads=2)      # the DP still needs the GPU
    m = np.zeros(1, mm2chain.MATCH_DTYPE); m["n"] = 2
    with pytest.raises(mm2chain.Mm2cError):
        mm2chain.seed_hits_batch([0, 1], m, np.zeros(2, np.uint64), [100])
    with pytest.raises(mm2chain.Mm2cError):
        mm2chain.seed_chain_batch(params.map_ont(), 3, 40, [0, 1], m, np.zeros(2, np.uint64), [100])
    with pytest.raises(mm2chain.Mm2cError):
        mm2chain.SeedPlan([0, 1], [0, 2])
    # the asynchronous initialisation (runtime start-up beside the host's index loading) cannot report at once: the failure surfaces at the wait and at the
    # first call that needs the device, with the reason; the busy-protocol entry does not turn it into "declined"
    from mm2chain import _native as N
    lib = N.load()
    assert lib.mm2c_init_async(-1) == 0
    assert lib.mm2c_init_wait() == -1 and b"asynchronous initialisation" in lib.mm2c_last_error()
    with pytest.raises(mm2chain.Mm2cError, match="asynchronous initialisation"):
        mm2chain.chain_task(params.map_ont(), a, 0.15)
    with pytest.raises(mm2chain.Mm2cError):
        mm2chain.chain_task_pred(params.map_ont(), a, 0.15, 0, 0.1, 5.0)
    with pytest.raises(mm2chain.Mm2cError):
        mm2chain.chain_task_pred(params.map_ont(), a, 0.15, 0, 10.0, 5.0)      # predictions that WOULD decline on a busy device: no device is an error, not "declined"
    assert lib.mm2c_device_count() == 0


def test_argument_validation_happens_before_any_device_work():
    import mm2chain
    from mm2chain import params
    bad = params.make_params(max_dist_x=-1)
    with pytest.raises(mm2chain.Mm2cError):
        mm2chain.ChainPlan(bad, [0, 4])
    with pytest.raises(mm2chain.Mm2cError):
        mm2chain.chain_batch_host(params.map_ont(), [0, 4, 2], np.zeros((4, 2), np.uint64))
    m = np.zeros(1, mm2chain.MATCH_DTYPE); m["n"] = 5                       # a match that reaches beyond the hit pool
    with pytest.raises(mm2chain.Mm2cError):
        mm2chain.seed_hits_batch([0, 1], m, np.zeros(2, np.uint64), [100])
    with pytest.raises(mm2chain.Mm2cError):
        mm2chain.seed_chain_batch(bad, 3, 40, [0, 1], m, np.zeros(8, np.uint64), [100])


def test_header_is_self_contained_c99_and_cxx11(tmp_path):
    """include/mm2chain.h compiles alone as C99 (pedantic) and as C++11, and the record layouts are the documented ones"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "h.c"
    src.write_text('#include "mm2chain.h"\n'
                   'typedef char anchor_is_16_bytes[sizeof(mm2c_anchor_t) == 16 ? 1 : -1];\n'
                   'typedef char match_is_24_bytes[sizeof(mm2c_match_t) == 24 ? 1 : -1];\n'
                   'int main(void) { mm2c_params_t p; mm2c_params_map_ont(&p); return p.bw != 500; }\n')
    inc = os.path.join(root, "include")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", inc, "-c", str(src), "-o", str(tmp_path / "c.o")])
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-I", inc, "-x", "c++", "-c", str(src), "-o", str(tmp_path / "cxx.o")])


def test_hand_written_loop_owns_the_kernels_only_lds_object():
    """the assembly of chain_dp_tile addresses LDS from 0: every instantiation's group segment in the shipped code objects must be exactly
    its Lds<> object (tools/check_lds_layout.py, also run by the Makefile after linking)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_lds_layout
    seen, bad = check_lds_layout.check(os.path.join(ROOT, "minimap2-fpga_amd", "libmm2chain_hip.so"))
    assert seen >= 20 and not bad, bad
    assert check_lds_layout.lds_bytes(8, 2, 0, 0, 0) == 5632   # 5.5 KB per wave: 28 waves per CU
    assert check_lds_layout.lds_bytes(16, 2, 0, 0, 1) == 6144  # the compact ring: 16 tiles in 6 KB, 26 waves per CU


def test_split_model_getter_returns_the_header_constants():
    """f4: mm2c_split_model hands out the constants of include/mm2chain_split.h (the form of chain_hardware.h:19-30) -- non-negative slopes,
    a positive per-call cost; no GPU needed"""
    import re
    import mm2chain
    txt = open(os.path.join(ROOT, "include", "mm2chain_split.h")).read()
    hdr = {m.group(1): float(m.group(2)) for m in re.finditer(r"#define MI355X_(\w+) ([-+0-9.eE]+)", txt)}
    assert len(hdr) == 10
    for preset, pre in (("map-ont", "ONT"), ("asm20", "PBCCS")):
        c = mm2chain.split_model(preset)
        for k, v in c.items():
            assert abs(v - hdr[f"{pre}_{k}"]) <= 1e-6 * max(1.0, abs(v)), (preset, k)
        assert c["K1_HW"] >= 0 and c["K2_HW"] >= 0 and c["K_SW"] > 0 and c["C_HW"] > 0
    with pytest.raises(mm2chain.Mm2cError):
        mm2chain.split_model("no-such-preset")


def test_bench_reports_profiled_traffic_only_for_the_profiled_kernel_sources():
    """bench.py's roofline.traffic comes from profiles/traffic.json; it must be a number exactly when the hash recorded there is the hash of
    the DP kernel's current sources, and None (never a stale figure) otherwise"""
    import hashlib
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    rec = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["mixed"]
    h = hashlib.sha256()
    for fn in ("chain_dp_tile.h", "chain_wave.h", "chain_kernel.hip", "chain_kernel.h"):
        h.update(open(os.path.join(ROOT, "minimap2-fpga_amd", "csrc", fn), "rb").read())
    got = bench.measured_traffic("mixed", rec["anchors_per_launch"])
    if rec["kernel_source_sha"] == h.hexdigest()[:16]:
        assert got == rec["hbm_bytes_per_launch"]
    else:
        assert got is None
    assert bench.measured_traffic("no such profile", 1) is None
