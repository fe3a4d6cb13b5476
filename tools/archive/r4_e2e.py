#!/usr/bin/env python3
"""the end-to-end leg of bench.py on its own (three hosts over a synthetic genome), printed as JSON; environment (MM2C_COOP_WAVES, ...) reaches the hosts"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else bench.host_cores()
r = bench.e2e_map_ont(threads, reads, 50.0, 100_000_000)
print(json.dumps({k: (v if k != "hosts" else {h: {kk: vv for kk, vv in d.items() if kk in ("wall_s", "rc", "paf_md5", "per_read_calls", "library_stage_stats")} for h, d in v.items()}) for k, v in r.items()}, indent=1))
